"""utils/Data_loader.py: the transform restatement against its definition (CPU) and the two loaders
end to end on the device."""
import numpy as np
import pytest
import torch


def test_shift_is_an_exact_zero_filled_translation():
    from utils.Data_loader import shift_zero_fill
    x = torch.arange(1.0, 21.0).view(1, 1, 4, 5)
    r = shift_zero_fill(x, 2, 0)                                  # content moves RIGHT by two columns
    assert torch.equal(r[0, 0, :, 2:], x[0, 0, :, :3]) and float(r[0, 0, :, :2].abs().sum()) == 0.0
    d = shift_zero_fill(x, 0, -1)                                 # content moves UP by one row
    assert torch.equal(d[0, 0, :3], x[0, 0, 1:]) and float(d[0, 0, 3].abs().sum()) == 0.0
    both = shift_zero_fill(x, -1, 2)
    assert torch.equal(both[0, 0, 2:, :4], x[0, 0, :2, 1:]) and float(both.sum()) == float(x[0, 0, :2, 1:].sum())
    assert torch.equal(shift_zero_fill(x, 0, 0), x) and float(shift_zero_fill(x, 5, 0).abs().sum()) == 0.0


def test_transform_consumes_the_generator_like_torchvision_and_acts_on_the_whole_batch():
    """Draw order: hflip (rand(1)), vflip (rand(1)), angle (uniform_(-0, 0)), tx, ty — five draws from torch's global
    generator; the SAME flip / shift for every image of the batch; |t| <= 10 % of the side, integer."""
    from utils.Data_loader import random_flip_translate, shift_zero_fill
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 1, 40, 40, generator=g)
    for seed in range(12):
        torch.manual_seed(seed)
        out = random_flip_translate(x)
        state_after = torch.get_rng_state()
        torch.manual_seed(seed)
        hf, vf = bool(torch.rand(1) < 0.5), bool(torch.rand(1) < 0.5)
        torch.empty(1).uniform_(-0.0, 0.0)
        tx = int(round(torch.empty(1).uniform_(-4.0, 4.0).item()))
        ty = int(round(torch.empty(1).uniform_(-4.0, 4.0).item()))
        assert torch.equal(torch.get_rng_state(), state_after)
        ref = x.flip(-1) if hf else x
        ref = ref.flip(-2) if vf else ref
        assert abs(tx) <= 4 and abs(ty) <= 4
        assert torch.equal(out, shift_zero_fill(ref, tx, ty))


class _Args:
    batch_size, distance_normalize, distance_normalize_constant = 6, 1.0, 0.0
    wavelength, pixel_size, phase_normalize = 532e-9, 1.5e-6, 1.0


@pytest.mark.gpu
def test_loaders_on_the_device():
    """mnist_loader_test / mnist_loader with a synthetic digit set: shapes, devices, the holograms are the propagator's
    intensity of (0.6, transformed phase, drawn distance) — recomputed here from the same seeds — and the interpolation
    mode returns one hologram per listed style distance."""
    from hip_helpers import DEV, orc
    from style_transfer_based_holographic_imaging_amd import synth
    from utils.Data_loader import mnist_loader, mnist_loader_test
    from utils.Forward_model import Holo_Generator
    ds = synth.SyntheticDigits(64)
    hg = Holo_Generator(_Args)
    d_list = [0.4, 0.5, 0.6, 0.7, 0.8]
    torch.manual_seed(11)
    np.random.seed(11)
    s_h, c_h, d_s, d_c = mnist_loader_test(_Args, ds, d_list, d_list, hg, DEV)
    assert s_h.shape == c_h.shape == (6, 1, 92, 92) and s_h.device == DEV and s_h.dtype == torch.float32
    assert d_s.shape == d_c.shape == (6, 1, 1, 1) and {round(float(v), 2) for v in d_c.view(-1).cpu()} <= set(d_list)
    # the same seeds through the pieces: first shuffled batch, two transform draws (content first), oracle propagation
    torch.manual_seed(11)
    np.random.seed(11)
    from utils.Data_loader import _draw_distances, _first_batch, random_flip_translate
    imgs = _first_batch(ds, 6, "cpu")
    ds_ref, dc_ref = _draw_distances(d_list, 6, "cpu"), _draw_distances(d_list, 6, "cpu")
    ph = torch.nn.functional.pad(imgs, (32, 32, 32, 32))
    ph_c = random_flip_translate(ph)
    ph_s = random_flip_translate(ph)
    assert torch.equal(d_s.cpu(), ds_ref) and torch.equal(d_c.cpu(), dc_ref)
    amp = torch.full_like(ph, 0.6)
    want_c = orc.holo_generator(orc.HoloParams(), amp, ph_c, dc_ref)
    want_s = orc.holo_generator(orc.HoloParams(), amp, ph_s, ds_ref)
    for got, want in ((c_h, want_c), (s_h, want_s)):
        assert float((got.cpu().double() - want.double()).abs().max() / want.double().abs().max()) < 2e-5
    # interpolation mode is the reference's batch-1 idiom: ONE amplitude plane broadcast against the repeated phase
    class One(_Args):
        batch_size = 1
    torch.manual_seed(12)
    np.random.seed(12)
    out = mnist_loader_test(One, ds, [0.2, 0.4, 0.9], d_list, hg, DEV, test_interpolation=True)
    assert len(out) == 6 and out[0].shape == out[1].shape == (3, 1, 92, 92) and out[2].shape == (3, 1, 1, 1)
    want = orc.holo_generator(orc.HoloParams(), out[4].cpu().expand(3, -1, -1, -1), out[5].cpu().repeat(3, 1, 1, 1), out[3].cpu().repeat(3, 1, 1, 1))
    assert float((out[1].cpu().double() - want.double()).abs().max() / want.double().abs().max()) < 2e-5
    four = mnist_loader(_Args, ds, d_list, d_list, hg, DEV)
    six = mnist_loader(_Args, ds, d_list, d_list, hg, DEV, return_gt=True)
    assert len(four) == 4 and len(six) == 6 and six[5].shape == (6, 1, 92, 92)
