#!/usr/bin/env python3
"""Spread of the parity measure (max|d|/max|ref| against the CPU fp32 oracle) over many 256x256 holograms, per convolution
algorithm: `python tests/diagnostics/parity_spread.py [n=32] [size=256]` (test infrastructure; run on the GPU box)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from hip_helpers import DEV, build_network, orc  # noqa: E402
from style_transfer_based_holographic_imaging_amd import synth  # noqa: E402
from utils.functions import field_retrieval  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
s = int(sys.argv[2]) if len(sys.argv) > 2 else 256


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


network, sds = build_network(0)
style = torch.load(os.path.join(REPO, "style_representation", "MNIST", "style_vector.pt"), map_location="cpu").float()
hp = orc.HoloParams()
torch.set_num_threads(16)
ph = synth.synthetic_phase_objects(n, s, seed=4321)
d_c = synth.synthetic_distances(n, seed=4321)
with torch.no_grad():
    content = torch.sqrt(orc.holo_generator(hp, torch.full_like(ph, 0.6), ph, d_c)).float()
    ref = orc.net_field_retrieval(content, style, *sds, 1.0, True)
    for algo in ("direct", "winograd", "auto"):      # "auto" at small n includes the split-K Winograd layers
        network.set_conv_algo(algo)
        got = field_retrieval(network, content.to(DEV), style.to(DEV), 1.0, True)
        for i, name in enumerate(("amplitude", "phase")):
            e = sorted(rel(got[i][k], ref[i][k]) for k in range(n))
            print(f"{algo:9s} {name:9s} per hologram: median {e[n // 2]:.2e}  90% {e[int(0.9 * n)]:.2e}  max {e[-1]:.2e}   whole batch {rel(got[i], ref[i]):.2e}")
