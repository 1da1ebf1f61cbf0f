"""Shared helpers of the GPU parity tests."""
import copy
import os
import sys

import torch
import torch.nn as nn

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "oracle"))
import holo_oracle as orc  # noqa: E402  (the checker)

DEV = torch.device("cuda:0")


def build_network(seed=0, device=DEV):
    import net
    from style_transfer_based_holographic_imaging_amd import synth
    vgg_sd, dec_sd, dist_sd = synth.synthetic_state_dicts(seed, full_vgg=True)
    vgg = copy.deepcopy(net.vgg)
    vgg.load_state_dict(vgg_sd)
    dec = copy.deepcopy(net.decoder)
    dec.load_state_dict(dec_sd)
    dg = net.Distance_G()
    dg.load_state_dict(dist_sd)
    network = net.Net(nn.Sequential(*list(vgg.children())[:31]), dec, None, dg).eval().to(device)
    return network, (vgg_sd, dec_sd, dist_sd)


def holo_generator():
    from utils.Forward_model import Holo_Generator
    return Holo_Generator(orc.HoloParams())


def cplx_rel(a, b):
    """relative error of two complex tensors on their re/im parts (phase compared through the field)."""
    a = torch.view_as_real(a.detach().cpu().to(torch.complex128))
    b = torch.view_as_real(b.detach().cpu().to(torch.complex128))
    return float((a - b).abs().max() / b.abs().max())
