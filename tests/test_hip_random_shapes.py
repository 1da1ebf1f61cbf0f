"""Whole-path differential check against the oracle over shapes the fixed tests do not visit: sizes that are not
multiples of 8 (ceil-mode pooling grows the output, e.g. 100 -> 104, which also sends the propagator down the
generic-size path), batches 1-3, alpha in {0, 0.5, 1}, and the three bundled style vectors."""
import os

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu
TOL = 1e-4   # BASELINE north_star: 1e-4 relative fp32 (max|d| / max|ref| per tensor); measured here <= 1e-5

CASES = [(200, 2, 0.5, "MNIST"), (136, 1, 1.0, "polystyrene_bead"), (40, 2, 1.0, "red_blood_cell"),
         (100, 1, 0.0, "polystyrene_bead"), (100, 2, 1.0, "MNIST"), (24, 2, 0.0, "MNIST"), (72, 3, 0.5, "red_blood_cell")]


@pytest.fixture(scope="module")
def env():
    from hip_helpers import build_network, holo_generator
    assert torch.cuda.is_available()
    network, sds = build_network(0)
    return network, sds, holo_generator()


@pytest.mark.parametrize("mode", ["f32", "f32s"])
@pytest.mark.parametrize("s,b,alpha,style_name", CASES)
def test_random_shapes_against_oracle(env, s, b, alpha, style_name, mode):
    from hip_helpers import DEV, cplx_rel, orc
    from utils.functions import field_retrieval
    network, sds, hg = env
    style = torch.load(os.path.join(REPO, "style_representation", style_name, "style_vector.pt"), map_location="cpu").float()
    g = torch.Generator().manual_seed(s * 10 + b)
    content = torch.rand(b, 1, s, s, generator=g) * 0.8 + 0.1
    d_style = torch.full((b, 1, 1, 1), 0.2)

    def rel(a, r):
        return float((a.double() - r.double()).abs().max() / r.double().abs().max())

    with torch.no_grad():
        a_r, p_r, d_r = orc.field_retrieval(content, style, *sds, alpha, True)
        network.set_compute_dtype(mode)          # "f32s": the opt-in split-fp16 conv stack, held to the same bar
        try:
            a, p, d = field_retrieval(network, content.to(DEV), style.to(DEV), alpha, True)
        finally:
            network.set_compute_dtype("f32")
        assert a.shape == a_r.shape and d.shape == d_r.shape == (b, 1, 1, 1)
        assert rel(a.cpu(), a_r) < TOL and rel(p.cpu(), p_r) < TOL and rel(d.cpu(), d_r) < TOL
        f_r = orc.holo_generator(orc.HoloParams(), a_r, p_r, -d_style, complex_number=True)
        f = hg(a, p, -d_style.to(DEV), complex_number=True)
        # the focused field inherits the network's phase error (|d exp(ip)| = |dp|): bound scaled as in test_hip_network
        assert cplx_rel(f, f_r) < TOL * (1.0 + float(p_r.abs().max()))
