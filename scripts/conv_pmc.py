#!/usr/bin/env python3
"""SQ counters of the bf16 3x3 convolution kernels on one layer shape, per tile variant (separate rocprofv3 --pmc passes over
scripts/bf16_layer_bench.py).  Usage (GPU box): python scripts/conv_pmc.py --tiles 5,8 --shape 128:256:256:0 -> JSON on stdout."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--tiles", default="5,8")
ap.add_argument("--shape", default="128:256:256:0")
ap.add_argument("--batch", default="64")
a = ap.parse_args()
GROUPS = ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE",
          "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES",
          "SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"]
res = {}
for tile in a.tiles.split(","):
    out = {}
    for gi, grp in enumerate(GROUPS):
        d = os.path.join(REPO, "gpurun_out", f"conv_pmc_t{tile}_{gi}")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + grp.split() + ["-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
               os.path.join(REPO, "scripts", "bf16_layer_bench.py"), "--tiles", tile, "--shapes", a.shape, "--batch", a.batch, "--rounds", "2"]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
        if r.returncode:
            out[f"group{gi}_error"] = r.stderr[-400:]
            continue
        f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "conv" not in name or "pack" in name:
                continue
            k = out.setdefault("k", {"n": {}, "c": {}, "name": name[:90]})
            cn = row["Counter_Name"]
            k["c"][cn] = k["c"].get(cn, 0.0) + float(row["Counter_Value"])
            k["n"][cn] = k["n"].get(cn, 0) + 1
    k = out.get("k")
    if not k:
        res[tile] = out
        continue
    c = {q: v / k["n"][q] for q, v in k["c"].items()}
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    res[tile] = {"kernel": k["name"], "launches": max(k["n"].values()),
                 "gpu_cycles_per_launch": round(c.get("GRBM_GUI_ACTIVE", 0) / 8.0),
                 "mfma_busy_frac_of_busy_cycles": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c.get("SQ_BUSY_CYCLES", 0) or 1) / 4.0, 4),
                 "wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 4), "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0) / wc, 4),
                 "active_inst_any_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4),
                 "wait_inst_lds_frac": round(c.get("SQ_WAIT_INST_LDS", 0) / wc, 4),
                 "lds_bank_conflict_over_idx_active": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / (c.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0), 4),
                 "raw": c, "errors": {q: v for q, v in out.items() if q != "k"}}
print(json.dumps(res, indent=1))
