#!/usr/bin/env python3
"""Per-layer sweep of the 3x3 algorithms at a small batch: direct kernel, Winograd, Winograd split-K 2/4/8 (us per call, back-to-back
launches between two HIP events).  `python scripts/wino_ksplit_sweep.py [--batch 5] [--size 128] [--reps 50]`"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from style_transfer_based_holographic_imaging_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=5)
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
s = a.size
# (name, H, Cin, Cout, pool) of the 3x3 layers with Cin >= 64 (net.py: vgg[:31] and decoder)
layers = [("E1", s, 64, 64, 1), ("E2", s // 2, 64, 128, 0), ("E3", s // 2, 128, 128, 1), ("E4", s // 4, 128, 256, 0),
          ("E5-7", s // 4, 256, 256, 0), ("E8", s // 8, 256, 512, 0), ("D0", s // 8, 512, 512, 0), ("D1", s // 8, 512, 256, 0),
          ("D3-5", s // 4, 256, 256, 0), ("D6", s // 4, 256, 128, 0), ("D8", s // 2, 128, 128, 0), ("D9", s // 2, 128, 64, 0),
          ("D11-12", s, 64, 64, 0)]
g = torch.Generator().manual_seed(0)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.reps


print(f"batch {a.batch}, {s}x{s} input; us per call")
print(f"{'layer':8s} {'HxW':>5s} {'Cin':>4s} {'Cout':>4s} {'direct':>8s} {'wino':>8s} " + " ".join(f"{'k=' + str(k):>8s}" for k in (2, 4, 8)))
with torch.no_grad():
    for name, h, cin, cout, pool in layers:
        x = torch.randn(a.batch, h, h, cin, generator=g).to(dev)
        w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        wd, ww = ops.pack_conv3x3(w, torch.float32), ops.pack_conv3x3_wino(w)
        row = [timeit(lambda: ops.conv3x3(x, wd, b, cout, True, bool(pool))), timeit(lambda: ops.conv3x3_wino(x, ww, b, cout, True, bool(pool)))]
        for k in (2, 4, 8):
            chunks = cin // 16
            if chunks % k == 0 and chunks // k >= 2:
                row.append(timeit(lambda: ops.conv3x3_wino(x, ww, b, cout, True, bool(pool), ksplit=k)))
            else:
                row.append(float("nan"))
        print(f"{name:8s} {h:5d} {cin:4d} {cout:4d} " + " ".join(f"{v:8.1f}" for v in row), flush=True)
