#!/usr/bin/env python3
"""One training step of the reference's objective on the HIP kernels, timed: Net.forward (net.py:199-226, all seven results) ->
a scalar objective -> backward -> SGD on the decoder and the distance head.  Prints one JSON line (informational: the reference
ships no trainer).   python scripts/train_step.py [--batch 16] [--size 256] [--steps 5]"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from style_transfer_based_holographic_imaging_amd import ops, synth  # noqa: E402
from utils.Forward_model import Holo_Generator  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--lr", type=float, default=1e-5)
a = ap.parse_args()
dev = torch.device("cuda:0")
network, _ = bench.build_network(dev)
params = list(network.decoder.parameters()) + list(network.distance_g.parameters())
for p in params:
    p.requires_grad_(True)
hg = Holo_Generator(bench.HoloParams())
with torch.no_grad():
    ph = synth.synthetic_phase_objects(2 * a.batch, a.size, seed=5).to(dev)
    d = synth.synthetic_distances(2 * a.batch, seed=5).to(dev)
    holo = torch.sqrt(hg(torch.full_like(ph, 0.6), ph, d))
content, style = holo[:a.batch].contiguous(), holo[a.batch:].contiguous()
opt = torch.optim.SGD(params, lr=a.lr)


def step():
    opt.zero_grad(set_to_none=True)
    lc, ls, g_t, g_ph, s_re, d_c, d_s = network(content, style, 1.0, True, True)
    loss = lc + 10.0 * ls + (s_re - style).pow(2).mean() + (d_c - 0.5).pow(2).mean() + (d_s - 0.2).pow(2).mean()
    loss.backward()
    opt.step()
    return float(loss.detach())


losses = [step()]
torch.cuda.synchronize()
timer = ops.KernelTimer()
t0 = time.perf_counter()
with timer:
    for _ in range(a.steps):
        losses.append(step())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
summ = timer.summary()
top = sorted(((v["ms"] / a.steps, k, v["flops"] / max(v["ms"], 1e-9) / 1e9, v["calls"] // a.steps) for k, v in summ.items()), reverse=True)[:10]
if os.environ.get("HOLO_TRAIN_CALLS"):
    want = os.environ["HOLO_TRAIN_CALLS"]
    per = [(n, e0.elapsed_time(e1), fl) for n, fl, nb, e0, e1 in timer.records if want in n]
    per = per[-(len(per) // a.steps):]
    for n, ms, fl in per:
        print(f"  {n:34s} {ms:8.3f} ms  {fl / 1e9:9.2f} GF  {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s", file=sys.stderr)
print(json.dumps({"metric": "training steps/s (Net.forward + backward + SGD, informational)", "batch": a.batch, "size": a.size,
                  "steps_per_s": round(1.0 / dt, 3), "ms_per_step": round(dt * 1e3, 2), "holograms_per_s": round(a.batch / dt, 1),
                  "losses": [round(v, 6) for v in losses],
                  "top_kernels_[ms_per_step,TFLOPs,calls]": {k: [round(ms, 3), round(tf, 1), c] for ms, k, tf, c in top}}))
