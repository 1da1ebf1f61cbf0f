#!/usr/bin/env python3
"""Task timeline of the dataflow propagator from a -DHOLO_ASM_TRACE=1 side build (HOLO_HIP_LIB=...): per phase, how long the
scheduler waited and how long the task ran; per XCD, the busy fraction of its block slots.  Diagnostics."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import HoloParams  # noqa: E402
from style_transfer_based_holographic_imaging_amd import ops  # noqa: E402
from utils.Forward_model import Holo_Generator  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
hg = Holo_Generator(HoloParams())
g = torch.Generator().manual_seed(0)
amp = (torch.rand(b, 1, s, s, generator=g) + 0.2).to(dev)
ph = (torch.rand(b, 1, s, s, generator=g) * 6 - 3).to(dev)
d = torch.full((b, 1, 1, 1), -0.2, device=dev)
for _ in range(5):
    hg(amp, ph, d, return_field=True)
torch.cuda.synchronize()
ops._ASM_KEEP_WORKSPACE[:] = [None]
hg(amp, ph, d, return_field=True)
torch.cuda.synchronize()
ws = ops._ASM_KEEP_WORKSPACE[0]
NREC = 1 << 16
tr = ws[ws.numel() - (8 + 8 * NREC) * 4:].view(torch.int32).cpu().numpy().astype("int64") & 0xFFFFFFFF
n = int(tr[0])
rec = tr[8:8 + 8 * min(n, NREC)].reshape(-1, 8)
print(f"{n} task records")
t0 = rec[:, 3].min()
us = lambda v: (v - t0) / 100.0   # 100 MHz
import numpy as np
xcc = rec[:, 0] & 255
phase = rec[:, 2] & 255
pop, ready, done, sig = us(rec[:, 3]), us(rec[:, 4]), us(rec[:, 5]), us(rec[:, 6])
print(f"span {sig.max():.1f} us (first pop -> last signal)")
for p_, name in enumerate("ABC"):
    m = phase == p_
    print(f"phase {name}: {m.sum()} tasks  sched+wait {np.mean(ready[m] - pop[m]):.2f} us (p90 {np.percentile(ready[m] - pop[m], 90):.2f})  "
          f"run {np.mean(done[m] - ready[m]):.2f} us (p90 {np.percentile(done[m] - ready[m], 90):.2f})  signal {np.mean(sig[m] - done[m]):.2f} us")
for x in range(8):
    m = xcc == x
    if not m.any():
        continue
    blocks = len(set(rec[m, 0] >> 8))
    busy = float(np.sum(done[m] - ready[m]))
    print(f"XCD {x}: {m.sum()} tasks on {blocks} blocks, first pop {pop[m].min():.1f} last signal {sig[m].max():.1f} us, "
          f"run time / (blocks * span) = {busy / (blocks * (sig[m].max() - pop[m].min())):.2f}")
# one XCD's field timeline
x = int(xcc[0])
m = xcc == x
js = (rec[m, 2] >> 8) & 255
for j in sorted(set(js))[:10]:
    mm = m.copy()
    mm[m] = js == j
    line = []
    for p_, name in enumerate("ABC"):
        q = mm & (phase == p_)
        if q.any():
            line.append(f"{name}: pop {pop[q].min():6.1f} ready {ready[q].min():6.1f}..{ready[q].max():6.1f} done {done[q].max():6.1f}")
    print(f"XCD {x} local field {j}: " + " | ".join(line))
