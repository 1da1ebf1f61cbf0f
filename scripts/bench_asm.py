#!/usr/bin/env python3
"""Propagator-only micro-benchmark: `python scripts/bench_asm.py [B] [S] [iters]` (run under rocprofv3
--kernel-trace --stats for the per-kernel split).  Prints achieved algorithmic GB/s (80*S^2 B per field)."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import HoloParams  # noqa: E402  (the five optics attributes Holo_Generator reads)
from utils.Forward_model import Holo_Generator  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
hg = Holo_Generator(HoloParams())
g = torch.Generator().manual_seed(0)
amp = (torch.rand(b, 1, s, s, generator=g) + 0.2).to(dev)
ph = (torch.rand(b, 1, s, s, generator=g) * 6 - 3).to(dev)
d = torch.full((b, 1, 1, 1), -0.2, device=dev)
for _ in range(5):
    hg(amp, ph, d, return_field=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    hg(amp, ph, d, return_field=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"B={b} S={s}: {dt * 1e6:.1f} us/call, {80.0 * s * s * b / dt / 1e9:.0f} GB/s algorithmic ({100 * 80.0 * s * s * b / dt / 8e12:.1f}% of 8 TB/s)")
