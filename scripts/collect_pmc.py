#!/usr/bin/env python3
"""HBM traffic of the hot kernels from rocprofv3 PMC counters, as MI355X_MICROARCH.md §HBM prescribes:
FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one), --kernel-trace only (no other trace
domain), FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), units KiB -> bytes.  Each pass runs

    rocprofv3 --kernel-trace --pmc <COUNTER> -d <dir> -o pmc --output-format csv -- python3 bench.py \
        --steps 2 --warmup 1 --cpu-sample 0 --no-events [bench args]

and only the LAST step's dispatches are summed (the first ones include weight packing and cold caches).
Writes profiles/<name>.json with the hash of the library the numbers belong to; bench.py reports
`roofline.traffic` from that file only while the hash matches the loaded library.

    python scripts/collect_pmc.py --out profiles/r04_pmc_traffic.json [-- bench args]
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# kernel-name prefixes -> keys of the output ("wino": the Winograd 3x3 kernel, the dominant one; "conv": the implicit-GEMM
# kernel, which still runs the transposed convolutions and the small 3x3 layers)
KEYS = (("conv_wino_f32_kernel", "wino"), ("conv_mfma_kernel", "conv"), ("conv3x3_b16_wres_kernel", "b16_wres"),
        ("conv3x3_b16_kernel", "b16_staged"), ("convt2x2_b16_c64_kernel", "b16_convt"), ("conv_first_b16_kernel", "b16_first"),
        ("conv_last_b16_kernel", "b16_last"), ("asm_rows_fwd", "asm_rows_fwd"), ("asm_cols", "asm_cols"), ("asm_rows_inv", "asm_rows_inv"))
SQ_KEYS = ("wino", "conv", "b16_wres", "b16_staged", "b16_convt", "b16_first", "b16_last")
ASM = ("asm_rows_fwd", "asm_cols", "asm_rows_inv")


def one_pass(counter, outdir, bench_args):
    """one rocprofv3 pass; `counter` may be a space-separated list (one --pmc group).  Returns rows
    (dispatch id, kernel name, counter name, value, start ns, end ns) sorted by dispatch."""
    os.makedirs(outdir, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counter.split() + ["-d", outdir, "-o", "pmc", "--output-format", "csv", "--",
           sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--no-events"] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise SystemExit(f"rocprofv3 pass {counter} failed ({r.returncode}):\n{r.stderr[-2000:]}")
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {outdir}")
    rows = []
    with open(sorted(files)[-1]) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] in counter.split():
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], row["Counter_Name"], float(row["Counter_Value"]),
                             int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    rows.sort()
    return rows


def last_step(rows):
    """dispatches of the last bench step: from the last conv_first launch on (each step starts with it)"""
    firsts = sorted({r[0] for r in rows if "conv_first" in r[1]})
    if len(firsts) > 2:
        lo = firsts[2]
        hi = firsts[3] if len(firsts) > 3 else 1 << 60
        return [r for r in rows if lo <= r[0] < hi]
    starts = [i for i, r in enumerate(rows) if "conv_first" in r[1]]
    # bench: 1 warm-up + 2 timed steps + (unwrap probe step, f32s steps) afterwards; take the 3rd start (2nd timed step)
    i0 = starts[2] if len(starts) > 2 else starts[-1]
    i1 = starts[3] if len(starts) > 3 else len(rows)
    return rows[i0:i1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r04_pmc_traffic.json"))
    ap.add_argument("--scratch", default=os.path.join(REPO, "gpurun_out", "pmc_r04"))
    ap.add_argument("bench_args", nargs="*")
    args = ap.parse_args()
    from style_transfer_based_holographic_imaging_amd import build
    build.build(verbose=False)
    res = {}
    for counter, scale in (("FETCH_SIZE", 2.0 * 1024.0), ("WRITE_SIZE", 1024.0)):
        rows = last_step(one_pass(counter, os.path.join(args.scratch, counter.lower()), args.bench_args))
        for _, name, _cn, val, _t0, _t1 in rows:
            short = name.split("(")[0]
            key = next((k for pre, k in KEYS if pre in short), None)
            if key is None:
                continue
            r = res.setdefault(key, {"launches": {}, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
            r[counter] += val * scale
            r["launches"][counter] = r["launches"].get(counter, 0) + 1
    out = {"source": __doc__.split("\n\n")[0].replace("\n", " "), "bench_args": args.bench_args,
           "source_hash": build.source_hash(), "kernels": {}}
    for key, r in res.items():
        n = max(r["launches"].values())
        out["kernels"][key] = {"launches_per_step": n, "hbm_read_bytes_per_step": r["FETCH_SIZE"],
                               "hbm_write_bytes_per_step": r["WRITE_SIZE"]}
    for key, field in (("wino", "conv_wino_hbm_bytes_per_launch_avg"), ("conv", "conv_mfma_hbm_bytes_per_launch_avg")):
        if key in out["kernels"]:
            c = out["kernels"][key]
            out[field] = (c["hbm_read_bytes_per_step"] + c["hbm_write_bytes_per_step"]) / c["launches_per_step"]
    # SQ / GRBM view of the dominant kernels (one pass: 8 SQ slots + GRBM): matrix-pipe utilisation, clock, wait fractions.
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's 1024 SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count
    # quad-cycles per wave; GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, cycle constants / DVFS notes).
    mops = "SQ_INSTS_VALU_MFMA_MOPS_BF16" if "bf16" in args.bench_args else "SQ_INSTS_VALU_MFMA_MOPS_F32"
    sq = f"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES {mops} SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
    try:
        rows = last_step(one_pass(sq, os.path.join(args.scratch, "sq"), args.bench_args))
        agg = {}
        for did, name, cn, val, t0, t1 in rows:
            short = name.split("(")[0]
            key = next((k for pre, k in KEYS if pre in short), None)
            if key not in SQ_KEYS:
                continue
            a = agg.setdefault(key, {"ns": {}, "c": {}})
            a["ns"][did] = t1 - t0
            a["c"][cn] = a["c"].get(cn, 0.0) + val
        for key, a in agg.items():
            c, ns = a["c"], sum(a["ns"].values())
            cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            out["kernels"].setdefault(key, {})["sq"] = {
                "launches": len(a["ns"]), "ns_profiled": ns, "clock_mhz": round(cyc / ns * 1e3, 1) if ns else None,
                "mfma_pipe_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0), 4) if cyc else None,
                "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
                "wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
                "active_inst_any_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
                "raw": c}
    except SystemExit as e:   # the traffic numbers stand on their own
        out["sq_error"] = str(e)[:300]
    asm = [v for k, v in out["kernels"].items() if k in ASM]
    if asm:
        out["asm_hbm_bytes_per_call"] = sum(v["hbm_read_bytes_per_step"] + v["hbm_write_bytes_per_step"] for v in asm)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
