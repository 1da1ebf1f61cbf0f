#!/usr/bin/env python3
"""What the reference's training objective costs on an MI355X through stock PyTorch-ROCm (MIOpen convolutions, aten autograd):
the same step as scripts/train_step.py (Net.forward, net.py:199-226, all seven results -> scalar objective -> backward -> SGD on
the decoder and the distance head) with the oracle's functional restatement on `cuda`.  Diagnostic only (test infrastructure:
it drives the oracle); the number is quoted in DESIGN.md next to the HIP path's.

  python tests/diagnostics/torch_rocm_train_baseline.py [--batch 16] [--size 256] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import holo_oracle as orc  # noqa: E402
from style_transfer_based_holographic_imaging_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--lr", type=float, default=1e-5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
vgg_sd, dec_sd, dist_sd = synth.synthetic_state_dicts(0, full_vgg=False)
vgg = {k: v.to(dev) for k, v in vgg_sd.items()}
dec = {k: v.to(dev).requires_grad_() for k, v in dec_sd.items()}
dist = {k: v.to(dev).requires_grad_() for k, v in dist_sd.items()}
ph = synth.synthetic_phase_objects(2 * a.batch, a.size, seed=5)
holo = (0.3 + 0.6 * ph).to(dev)                     # timing is data-independent
content, style = holo[:a.batch].contiguous(), holo[a.batch:].contiguous()
opt = torch.optim.SGD(list(dec.values()) + list(dist.values()), lr=a.lr)


def step():
    opt.zero_grad(set_to_none=True)
    lc, ls, g_t, g_ph, s_re, d_c, d_s = orc.net_forward(content, style, vgg, dec, dist, 1.0, True, True)
    loss = lc + 10.0 * ls + (s_re - style).pow(2).mean() + (d_c - 0.5).pow(2).mean() + (d_s - 0.2).pow(2).mean()
    loss.backward()
    opt.step()
    return float(loss.detach())


losses = [step(), step()]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    losses.append(step())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"metric": "training steps/s, stock PyTorch-ROCm (aten autograd, MIOpen), fp32 NCHW", "batch": a.batch, "size": a.size,
                  "steps_per_s": round(1.0 / dt, 3), "ms_per_step": round(dt * 1e3, 2), "holograms_per_s": round(a.batch / dt, 1),
                  "losses": [round(v, 6) for v in losses], "torch": torch.__version__}))
