"""GPU parity of holo_unwrap2d_f32 against oracle/unwrap_oracle.py (sequential restatement of the published
algorithm behind skimage.restoration.unwrap_phase) and against analytic known answers."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_test_data

sys.path.insert(0, os.path.join(REPO, "oracle"))
import unwrap_oracle as uo  # noqa: E402  (the checker)
import unwrap_cases as uc  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def unwrap():
    from hip_helpers import DEV
    from utils.functions import unwrap as f
    assert torch.cuda.is_available()
    return lambda a: f(torch.as_tensor(np.ascontiguousarray(a)).to(DEV)).cpu().numpy()


@pytest.mark.parametrize("h,w,seed", [(32, 32, 0), (48, 80, 1), (96, 64, 2), (128, 128, 3), (256, 256, 4), (200, 330, 5)])
def test_smooth_surface_known_answer(unwrap, h, w, seed):
    true = uc.smooth_surface(h, w, seed=seed)
    wrapped = uc.wrap(true)
    un = unwrap(wrapped[None, None])[0, 0]
    m = uc.global_shift(un, true)
    assert np.abs(un - true - 2 * np.pi * m).max() < 5e-6
    k = np.round((un.astype(np.float64) - wrapped) / (2 * np.pi)).astype(np.int64)
    vals, counts = np.unique(k, return_counts=True)
    assert vals[np.argmax(counts)] == 0                       # the library's global-cycle convention


def test_wrap_free_batch_is_unchanged(unwrap):
    rng = np.random.default_rng(7)
    x = (rng.random((6, 1, 40, 56)) * 2 - 1).astype(np.float32)
    assert np.array_equal(unwrap(x), x)


@pytest.mark.parametrize("h,w,sigma", [(40, 36, 0.9), (64, 64, 1.2), (96, 80, 0.7), (128, 128, 1.0)])
def test_bit_exact_with_oracle_on_fields_with_residues(unwrap, h, w, sigma):
    """Noise makes residues: the answer then depends on the spanning tree, which must be the oracle's."""
    x = np.stack([uc.noisy_wrapped(h, w, sigma, seed=s) for s in range(5)])[:, None]
    got = unwrap(x)
    want = uo.unwrap_batch(x, convention="mode")
    assert np.array_equal(got, want)
    assert len(np.unique(np.round((want - x) / (2 * np.pi)))) > 2


def test_vortex_and_constant_images(unwrap):
    x = np.stack([uc.vortex(64, 64), np.full((64, 64), 1.25, np.float32), np.zeros((64, 64), np.float32)])[:, None]
    assert np.array_equal(unwrap(x), uo.unwrap_batch(x, convention="mode"))


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (7, 1), (2, 2), (3, 3), (5, 4), (17, 3)])
def test_degenerate_shapes(unwrap, shape):
    x = uc.wrap(np.linspace(0, 9, shape[0] * shape[1]).reshape(shape))[None, None]
    assert np.array_equal(unwrap(x), uo.unwrap_batch(x, convention="mode"))


def test_strided_phase_plane_view_and_generator_flag(unwrap):
    """Holo_Generator(..., unwrap=True) (utils/Forward_model.py:30-31) on the bundled data, and a (B,2,S,S)[:,1:] view."""
    from hip_helpers import DEV, holo_generator
    hg = holo_generator()
    amp = load_test_data(0, "gt_amplitude").to(DEV)
    ph = load_test_data(0, "gt_phase").to(DEV) * 9.0                     # enough phase to wrap after propagation
    d = load_test_data(0, "distance_content").to(DEV)
    a0, p0 = hg(amp, ph, d, return_field=True)
    a1, p1 = hg(amp, ph, d, return_field=True, unwrap=True)
    assert torch.equal(a0, a1) and p1.is_cuda and p1.dtype == torch.float32 and p1.shape == p0.shape
    want = uo.unwrap_batch(p0.cpu().numpy(), convention="mode")
    assert np.array_equal(p1.cpu().numpy(), want)
    assert np.abs(want - p0.cpu().numpy()).max() > 6.0                   # something was unwrapped
    both = torch.stack([a0[:, 0], p0[:, 0]], dim=1)                      # (B,2,S,S); [:,1:] has batch stride 2 S^2
    from utils.functions import unwrap as f
    assert torch.equal(f(both[:, 1:]), p1)


def test_full_size_batch_properties(unwrap):
    """BASELINE config 2 size (64 x 256^2): congruence (output - input is a multiple of 2 pi), idempotence, and
    no neighbour step above pi on a residue-free field."""
    from hip_helpers import DEV
    from utils.functions import unwrap as f
    xs = np.stack([uc.wrap(uc.smooth_surface(256, 256, peak=30.0 + i, seed=i)) for i in range(64)])[:, None]
    x = torch.as_tensor(xs).to(DEV)
    un = f(x)
    k = (un.double() - x.double()) / (2 * np.pi)
    assert (k - k.round()).abs().max() < 1e-6
    assert torch.equal(f(un), un)
    assert max((un[..., 1:, :] - un[..., :-1, :]).abs().max(), (un[..., :, 1:] - un[..., :, :-1]).abs().max()) < np.pi
    assert k.round().abs().max() >= 3
