#!/usr/bin/env python3
"""SQ counters of the three propagator kernels (one rocprofv3 --pmc pass over scripts/bench_asm.py): how busy the vector ALU and
the LDS are and where waves wait.  Usage (GPU box): python scripts/asm_pmc.py [B] [S] -> JSON on stdout."""
import csv
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = sys.argv[1] if len(sys.argv) > 1 else "64"
s = sys.argv[2] if len(sys.argv) > 2 else "256"
GROUPS = ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE",
          "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"]
out = {}
for gi, grp in enumerate(GROUPS):
    d = os.path.join(REPO, "gpurun_out", f"asm_pmc_{gi}")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + grp.split() + ["-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
           os.path.join(REPO, "scripts", "bench_asm.py"), b, s, "10", "--eager"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    if r.returncode:
        out[f"group{gi}_error"] = r.stderr[-500:]
        continue
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        if not name.startswith("asm_"):
            continue
        k = out.setdefault(name, {"n": {}, "c": {}, "ns": 0})
        cn = row["Counter_Name"]
        k["c"][cn] = k["c"].get(cn, 0.0) + float(row["Counter_Value"])
        k["n"][cn] = k["n"].get(cn, 0) + 1
        if cn == grp.split()[0]:
            k["ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
res = {}
for name, k in out.items():
    if not isinstance(k, dict) or "c" not in k:
        res[name] = k
        continue
    c = k["c"]
    n = max(k["n"].values())
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / n
    res[name] = {"launches": n, "avg_us_profiled": round(k["ns"] / max(k["n"].get("SQ_WAVE_CYCLES", 1), 1) / 1e3, 2),
                 "gpu_cycles_per_launch": round(cyc),
                 "valu_active_frac_of_wave_cycles": round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4),
                 "lds_active_frac_of_wave_cycles": round(c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 4),
                 "vmem_active_frac_of_wave_cycles": round(c.get("SQ_ACTIVE_INST_VMEM", 0) / wc, 4),
                 "wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 4), "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0) / wc, 4),
                 "insts_per_launch": {q: round(c.get(q, 0) / n) for q in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU")},
                 "lds_bank_conflict_over_idx_active": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / (c.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0), 4),
                 "raw": {q: v / n for q, v in c.items()}}
print(json.dumps(res, indent=1))
