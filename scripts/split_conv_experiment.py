#!/usr/bin/env python3
"""Per-layer accuracy (against float64) and speed of the opt-in "f32s" conv (fp32 values as split fp16 halves on the fp16
matrix cores) next to the fp32-MFMA conv: `python scripts/split_conv_experiment.py`."""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_based_holographic_imaging_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def run(b, h, wd, cin, cout, reps=20):
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(b, cin, h, wd, generator=g).abs() * 0.7            # post-ReLU-like activations
    w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    ref = F.relu(F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double(), bias.cpu().double()))
    xp = ops.to_nhwc(x.to(dev))
    res = {}
    for name, dt in (("fp32 MFMA", torch.float32), ("f32s", ops.F32S)):
        xa, wp = ops.cast(xp, dt), ops.pack_conv3x3(w.to(dev), dt)
        y = ops.conv3x3(xa, wp, bias, cout, True, False)
        err = float((ops.as_nchw_view(ops.cast(y, torch.float32)).cpu().double() - ref).abs().max() / ref.abs().max())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.conv3x3(xa, wp, bias, cout, True, False)
        torch.cuda.synchronize()
        res[name] = (err, (time.perf_counter() - t0) / reps)
    fl = 2.0 * 9 * cin * cout * b * h * wd
    print(f"B={b} {cin}->{cout} @{h}x{wd}: " + " | ".join(
        f"{k}: {e:.2e} of float64, {t * 1e3:.3f} ms = {fl / t / 1e12:.0f} TFLOP/s" for k, (e, t) in res.items()), flush=True)


for cfg in ((64, 64, 64, 256, 256), (64, 256, 256, 64, 64), (64, 32, 32, 512, 512), (64, 128, 128, 128, 128)):
    run(*cfg)
