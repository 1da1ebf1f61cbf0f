#!/usr/bin/env python3
"""Phase-unwrapping micro-benchmark: `python tests/diagnostics/bench_unwrap.py [B] [S] [iters] [kind]`, kind = smooth | noisy |
focus (the back-propagated phase of a synthetic network step).  Prints ms per batch and images/s; with the oracle
importable also times the sequential restatement on one image for scale."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
from utils.functions import unwrap  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
kind = sys.argv[4] if len(sys.argv) > 4 else "smooth"
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
y, x = np.mgrid[0:s, 0:s] / s
imgs = []
for i in range(min(b, 8)):
    cx, cy = 0.3 + 0.4 * rng.random(2)
    true = (10 + 2 * i) * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / 0.08) + 6 * x - 3 * y
    if kind == "noisy":
        true = true + 0.9 * rng.standard_normal((s, s))
    imgs.append(np.angle(np.exp(1j * true)).astype(np.float32))
ph = torch.as_tensor(np.stack(imgs)).to(dev)
ph = ph[torch.arange(b) % ph.shape[0]].unsqueeze(1).contiguous()
for _ in range(3):
    unwrap(ph)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    out = unwrap(ph)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"unwrap B={b} S={s} {kind}: {dt * 1e3:.3f} ms/batch, {b / dt:.0f} images/s, {dt / b * 1e6:.1f} us/image")
try:
    import unwrap_oracle as uo
    t0 = time.perf_counter()
    ref = uo.apply_offsets(imgs[0], uo.mode_convention(uo.unwrap_offsets(imgs[0])))
    print(f"  sequential restatement (1 core, python): {(time.perf_counter() - t0) * 1e3:.0f} ms/image; "
          f"bit-exact: {np.array_equal(ref, out[0, 0].cpu().numpy())}")
except ImportError:
    pass
