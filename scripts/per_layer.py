#!/usr/bin/env python3
"""Per-launch timing of one bench step (HIP events): name, algorithmic GFLOP, ms, TFLOP/s.
  python scripts/per_layer.py [--batch 64] [--size 256] [--dtype f32] [--conv-algo auto]"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from style_transfer_based_holographic_imaging_amd import ops, synth  # noqa: E402
from utils.functions import field_retrieval  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--conv-algo", default="auto")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tile", type=int, default=0, help="holo_debug_set_conv_tile variant (A/B of the implicit-GEMM kernel's tiles)")
a = ap.parse_args()
dev = torch.device("cuda:0")
ops.set_conv_tile(a.tile)
network, _ = bench.build_network(dev)
network.set_conv_algo(a.conv_algo)
if a.dtype != "f32":
    network.set_compute_dtype({"bf16": torch.bfloat16, "f32s": "f32s"}[a.dtype])
style = torch.load(os.path.join(REPO, "style_representation", "MNIST", "style_vector.pt"), map_location="cpu").float().to(dev)
content = (0.3 + 0.6 * synth.synthetic_phase_objects(a.batch, a.size, seed=1)).to(dev)
with torch.no_grad():
    for _ in range(2):
        field_retrieval(network, content, style, 1.0, True)
    torch.cuda.synchronize()
    t = ops.KernelTimer()
    with t:
        for _ in range(a.reps):
            field_retrieval(network, content, style, 1.0, True)
    torch.cuda.synchronize()
n = len(t.records) // a.reps
tot = 0.0
for i in range(n):
    recs = [t.records[i + k * n] for k in range(a.reps)]
    name, fl, by = recs[0][0], recs[0][1], recs[0][2]
    ms = sorted(e0.elapsed_time(e1) for *_, e0, e1 in recs)[len(recs) // 2]
    tot += ms
    print(f"{i:2d} {name:28s} {fl / 1e9:9.2f} GFLOP {by / 1e6:9.1f} MB {ms:8.4f} ms {fl / ms / 1e9 if ms else 0:8.1f} TFLOP/s {by / ms / 1e6 if ms else 0:8.1f} GB/s")
print(f"total {tot:.3f} ms per step -> {a.batch / tot * 1e3:.1f} holograms/s (network only)")
