"""GPU parity of the fused angular-spectrum propagator: the bundled known-answer data, the golden
vectors of the reference, the oracle at the BASELINE sizes, and size-independent properties."""
import pytest
import torch

from conftest import load_test_data, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hg():
    from hip_helpers import holo_generator
    assert torch.cuda.is_available()
    return holo_generator()


@pytest.fixture(scope="module")
def dev():
    from hip_helpers import DEV
    return DEV


@pytest.mark.parametrize("i", range(20))
def test_kat_bundled_holograms(hg, dev, i):
    """content_holo == |ASM(gt_amp*exp(i gt_phase), d mm)|^2 (reference reaches 5.3e-7)."""
    holo = load_test_data(i, "content_holo")
    out = hg(load_test_data(i, "gt_amplitude").to(dev), load_test_data(i, "gt_phase").to(dev),
             load_test_data(i, "distance_content").to(dev))
    assert out.dtype == torch.float32 and out.shape == holo.shape
    assert rel_err(out.cpu(), holo) < 5e-6


@pytest.mark.parametrize("s", [16, 64])
def test_golden_modes(hg, dev, golden, s):
    from hip_helpers import cplx_rel
    from utils.Angular_Spectrum_Method import ASM
    g = golden("golden_asm.npz")
    amp, ph, d = g[f"s{s}_amp_in"].to(dev), g[f"s{s}_ph_in"].to(dev), g[f"s{s}_d_mm"].to(dev)
    fld = hg(amp, ph, d, complex_number=True)
    assert fld.dtype == torch.complex128
    assert cplx_rel(fld, torch.complex(g[f"s{s}_field_re"], g[f"s{s}_field_im"])) < 5e-6
    a_o, p_o = hg(amp, ph, d, return_field=True)
    assert rel_err(a_o.cpu(), g[f"s{s}_amp"]) < 5e-6
    # phase through the field: a*exp(ip) against the golden complex field
    assert cplx_rel(torch.polar(a_o, p_o), torch.complex(g[f"s{s}_field_re"], g[f"s{s}_field_im"])) < 5e-6
    assert rel_err(hg(amp, ph, d).cpu(), g[f"s{s}_intensity"]) < 5e-6
    o_low = amp * torch.exp(1j * ph)
    nopad = ASM(o_low, 532e-9, d * 1e-3, 1.5e-6, zero_padding=False)
    assert cplx_rel(nopad, torch.complex(g[f"s{s}_nopad_re"], g[f"s{s}_nopad_im"])) < 5e-6
    scal = ASM(o_low, 532e-9, 3.0e-4, 1.5e-6, zero_padding=True)
    assert cplx_rel(scal, torch.complex(g[f"s{s}_scalar_re"], g[f"s{s}_scalar_im"])) < 5e-6


def test_golden_two_channel(dev, golden):
    from hip_helpers import cplx_rel
    from utils.Angular_Spectrum_Method import ASM
    g = golden("golden_asm.npz")
    o2 = torch.complex(g["c2_in_re"], g["c2_in_im"]).to(dev)
    r2 = ASM(o2, 532e-9, torch.tensor([1e-4, -3e-4]).view(-1, 1, 1, 1).to(dev), 1.5e-6, zero_padding=True)
    assert cplx_rel(r2, torch.complex(g["c2_re"], g["c2_im"])) < 5e-6


@pytest.mark.parametrize("s,b", [(32, 3), (128, 4), (256, 4), (512, 2), (1024, 1)])
def test_against_oracle_at_baseline_sizes(hg, dev, s, b):
    """P = 2S = 64 .. 2048: every FFT plan, against the complex128 oracle."""
    from hip_helpers import cplx_rel, orc
    from style_transfer_based_holographic_imaging_amd import synth
    ph = synth.synthetic_phase_objects(b, s, seed=s)
    amp = torch.full_like(ph, 0.6) + 0.05 * torch.rand(ph.shape, generator=torch.Generator().manual_seed(s))
    d = torch.tensor([0.4, -0.2, 0.8, 0.55][:b]).view(-1, 1, 1, 1)
    ref = orc.holo_generator(orc.HoloParams(), amp, ph * 3.0, d, complex_number=True)
    got = hg(amp.to(dev), (ph * 3.0).to(dev), d.to(dev), complex_number=True)
    assert cplx_rel(got, ref) < 1e-5


def test_strided_views_and_roundtrip(hg, dev):
    """amp/phase as views of one (B,2,S,S) buffer (net.py:250-251); +d then -d restores the field
    wherever the replicate padding does not bite (centre region)."""
    g = torch.Generator().manual_seed(9)
    gt = torch.rand(4, 2, 64, 64, generator=g).to(dev)
    amp, ph = gt[:, :1], gt[:, 1:]
    d = torch.tensor([0.05, 0.1, 0.05, 0.1]).view(-1, 1, 1, 1).to(dev)
    a1, p1 = hg(amp, ph, d, return_field=True)
    a_c, p_c = hg(amp.contiguous(), ph.contiguous(), d, return_field=True)
    assert torch.equal(a1, a_c) and torch.equal(p1, p_c)
    a2, p2 = hg(a1, p1, -d, return_field=True)
    back = torch.polar(a2, p2)[..., 16:48, 16:48]
    orig = torch.polar(amp, ph)[..., 16:48, 16:48]
    assert float((back - orig).abs().max()) < 0.15   # edge effects of the finite aperture only


def test_linearity_and_batch_independence(hg, dev):
    """ASM is linear in the field; each hologram is independent of its batch neighbours."""
    from utils.Angular_Spectrum_Method import ASM
    g = torch.Generator().manual_seed(4)
    o1 = torch.complex(torch.randn(3, 1, 128, 128, generator=g), torch.randn(3, 1, 128, 128, generator=g)).to(dev)
    o2 = torch.complex(torch.randn(3, 1, 128, 128, generator=g), torch.randn(3, 1, 128, 128, generator=g)).to(dev)
    d = torch.tensor([2e-4, 5e-4, -3e-4]).view(-1, 1, 1, 1).to(dev)
    r1, r2 = ASM(o1, 532e-9, d, 1.5e-6, zero_padding=True), ASM(o2, 532e-9, d, 1.5e-6, zero_padding=True)
    r12 = ASM(o1 + 2.0 * o2, 532e-9, d, 1.5e-6, zero_padding=True)
    assert float((r12 - (r1 + 2.0 * r2)).abs().max() / r12.abs().max()) < 5e-6
    single = ASM(o1[1:2], 532e-9, d[1:2], 1.5e-6, zero_padding=True)
    assert torch.equal(single, r1[1:2])


def test_sweep_matches_repeat_idiom(hg, dev):
    g = torch.Generator().manual_seed(6)
    amp = (torch.rand(2, 1, 64, 64, generator=g) + 0.3).to(dev)
    ph = torch.rand(2, 1, 64, 64, generator=g).to(dev)
    dist = torch.linspace(-0.1, -0.8, 8)
    a_s, p_s = hg.sweep(amp, ph, dist)
    assert a_s.shape == (2, 8, 64, 64)
    for k in range(8):
        a_k, p_k = hg(amp, ph, dist[k].view(1, 1, 1, 1).expand(2, 1, 1, 1).to(dev), return_field=True)
        assert torch.equal(a_s[:, k:k + 1], a_k) and torch.equal(p_s[:, k:k + 1], p_k)


@pytest.mark.parametrize("s,b", [(128, 3), (256, 2), (512, 2), (1024, 1)])
def test_sweep_at_baseline_sizes(hg, dev, s, b):
    """BASELINE configs[4] idiom at its own size (and the sizes below it): every field to 8 distances in one sweep.  Covers
    the n_dist > 1 indexing of every column / inverse-row kernel variant (P = 256, 512: one wave per column; P = 1024: the
    persistent column kernel; P = 2048: the row-major path).  Held (i) bit-exact against the reference idiom
    ``forward(a.repeat(n), p.repeat(n), distances)`` (utils/Data_loader.py:59-62) run field by field, and (ii) within 1e-5 of the
    complex128 oracle, through the complex field so that no angle discontinuity enters."""
    from hip_helpers import cplx_rel, orc
    from style_transfer_based_holographic_imaging_amd import synth
    ph = synth.synthetic_phase_objects(b, s, seed=1237) * 3.0
    amp = torch.full_like(ph, 0.6) + 0.05 * torch.rand(ph.shape, generator=torch.Generator().manual_seed(s))
    dist = torch.linspace(-0.1, -0.8, 8)
    a_s, p_s = hg.sweep(amp.to(dev), ph.to(dev), dist)
    i_s = hg.sweep(amp.to(dev), ph.to(dev), dist, return_field=False)
    assert a_s.shape == (b, 8, s, s) and p_s.shape == (b, 8, s, s) and i_s.shape == (b, 8, s, s)
    hp = orc.HoloParams()
    for f in range(b):
        a_k, p_k = hg(amp[f:f + 1].repeat(8, 1, 1, 1).to(dev), ph[f:f + 1].repeat(8, 1, 1, 1).to(dev),
                      dist.view(8, 1, 1, 1).to(dev), return_field=True)
        assert torch.equal(a_s[f], a_k[:, 0]) and torch.equal(p_s[f], p_k[:, 0]), f"field {f}"
        ref = orc.holo_generator(hp, amp[f:f + 1].repeat(8, 1, 1, 1), ph[f:f + 1].repeat(8, 1, 1, 1), dist.view(8, 1, 1, 1),
                                 complex_number=True)
        got = torch.polar(a_s[f], p_s[f]).unsqueeze(1)
        assert cplx_rel(got, ref) < 1e-5, f"field {f}"
        assert rel_err(i_s[f].cpu().unsqueeze(1), ref.abs() ** 2) < 2e-5, f"field {f}"


def test_config5_red_blood_cell_sweep(dev, hg):
    """BASELINE configs[4] end to end on one 1024x1024 frame: field retrieval with the red_blood_cell style vector, then
    the 8-distance sweep of the retrieved field; the sweep of the NETWORK's output against the oracle's propagator fed the
    same amplitude / phase (1e-5), and against the one-distance-at-a-time path (bit-exact)."""
    import os
    from conftest import REPO
    from hip_helpers import build_network, cplx_rel, orc
    from style_transfer_based_holographic_imaging_amd import synth
    from utils.functions import field_retrieval
    network, _ = build_network(0)
    style = torch.load(os.path.join(REPO, "style_representation", "red_blood_cell", "style_vector.pt"), map_location="cpu").float()
    ph = synth.synthetic_phase_objects(1, 1024, seed=1237)
    d_c = synth.synthetic_distances(1, seed=1237)
    dist = torch.linspace(-0.1, -0.8, 8)
    with torch.no_grad():
        content = torch.sqrt(hg(torch.full_like(ph, 0.6).to(dev), ph.to(dev), d_c.to(dev)))
        amp, p, d = field_retrieval(network, content, style.to(dev), 1.0, True)
        assert torch.isfinite(amp).all() and torch.isfinite(p).all() and 0.0 < float(d) < 1.0
        a_s, p_s = hg.sweep(amp, p * 1.0, dist)
        for k in (0, 3, 7):
            a_k, p_k = hg(amp, p * 1.0, dist[k].view(1, 1, 1, 1).to(dev), return_field=True)
            assert torch.equal(a_s[:, k:k + 1], a_k) and torch.equal(p_s[:, k:k + 1], p_k)
        ref = orc.holo_generator(orc.HoloParams(), amp.cpu().repeat(8, 1, 1, 1), p.cpu().repeat(8, 1, 1, 1),
                                 dist.view(8, 1, 1, 1), complex_number=True)
    assert cplx_rel(torch.polar(a_s[0], p_s[0]).unsqueeze(1), ref) < 1e-5


@pytest.mark.parametrize("s,b,padded", [(40, 2, True), (100, 2, True), (36, 3, False), (25, 2, False), (6, 1, True)])
def test_generic_sizes_against_oracle(hg, dev, s, b, padded):
    """Transform sides that are not powers of two (80, 200, 36, 25, 12) take the direct-DFT kernels."""
    from hip_helpers import cplx_rel, orc
    from utils.Angular_Spectrum_Method import ASM
    g = torch.Generator().manual_seed(s)
    amp = torch.rand(b, 1, s, s, generator=g) + 0.3
    ph = (torch.rand(b, 1, s, s, generator=g) - 0.5) * 5.0
    d = torch.tensor([0.3, -0.45, 0.7][:b]).view(-1, 1, 1, 1)
    if padded:
        ref = orc.holo_generator(orc.HoloParams(), amp, ph, d, complex_number=True)
        got = hg(amp.to(dev), ph.to(dev), d.to(dev), complex_number=True)
        a_o, p_o = hg(amp.to(dev), ph.to(dev), d.to(dev), return_field=True)
        assert cplx_rel(torch.polar(a_o, p_o), ref) < 1e-5
    else:
        o_low = amp * torch.exp(1j * ph)
        ref = orc.asm(o_low, 532e-9, d * 1e-3, 1.5e-6, zero_padding=False)
        got = ASM(o_low.to(dev), 532e-9, (d * 1e-3).to(dev), 1.5e-6, zero_padding=False)
    assert got.shape == ref.shape and cplx_rel(got, ref) < 1e-5


def test_rejects_unsupported_sizes(hg, dev):
    with pytest.raises(RuntimeError, match="even field side"):
        hg(torch.ones(1, 1, 41, 41, device=dev), torch.zeros(1, 1, 41, 41, device=dev), 0.3)
    with pytest.raises(RuntimeError, match="outside"):
        hg(torch.ones(1, 1, 2050, 2050, device=dev), torch.zeros(1, 1, 2050, 2050, device=dev), 0.3)
    with pytest.raises(RuntimeError, match="ROCm device"):
        hg(torch.ones(1, 1, 32, 32), torch.zeros(1, 1, 32, 32), 0.3)


def test_batch_chunking_and_odd_layouts(hg, dev, monkeypatch):
    """Large batches are processed in groups sized for the Infinity Cache; force tiny groups and compare.
    Also: non-contiguous amplitude/phase (transposed views) and a float64 distance tensor."""
    from style_transfer_based_holographic_imaging_amd import ops
    g = torch.Generator().manual_seed(12)
    amp = (torch.rand(7, 1, 64, 64, generator=g) + 0.2).to(dev)
    ph = (torch.rand(7, 1, 64, 64, generator=g) * 4 - 2).to(dev)
    d = torch.linspace(0.2, 0.8, 7).view(-1, 1, 1, 1).to(dev)
    a_ref, p_ref = hg(amp, ph, d, return_field=True)
    monkeypatch.setattr(ops, "_ASM_CACHE_BUDGET", 3 * 64 * 128 * 8 * 2)      # 3 fields per launch group
    a_c, p_c = hg(amp, ph, d, return_field=True)
    assert torch.equal(a_c, a_ref) and torch.equal(p_c, p_ref)
    sw_ref = hg.sweep(amp[:5], ph[:5], [-0.1, -0.3, -0.5])                    # 5 fields x 3 distances, 1 per group
    monkeypatch.setattr(ops, "_ASM_CACHE_BUDGET", 1)
    sw_c = hg.sweep(amp[:5], ph[:5], [-0.1, -0.3, -0.5])
    assert torch.equal(sw_c[0], sw_ref[0]) and torch.equal(sw_c[1], sw_ref[1])
    monkeypatch.undo()
    a_t, p_t = hg(amp.transpose(-1, -2), ph.transpose(-1, -2), d, return_field=True)   # strided planes -> copied
    a_tc, p_tc = hg(amp.transpose(-1, -2).contiguous(), ph.transpose(-1, -2).contiguous(), d, return_field=True)
    assert torch.equal(a_t, a_tc) and torch.equal(p_t, p_tc)
    a64, _ = hg(amp, ph, d.double(), return_field=True)                        # float64 distances: full-precision 2*pi*d
    assert rel_err(a64.cpu(), a_ref.cpu()) < 1e-4


def test_back_prop_module(dev):
    import types
    from hip_helpers import cplx_rel, orc
    from utils.Forward_model import Back_prop
    args = types.SimpleNamespace(amplitude_normalize=1.0, wavelength=532e-9, pixel_size=1.5e-6, distance_normalize=1.0,
                                 distance_normalize_constant=0.0, Holo_G_input="re_im")
    holo = torch.rand(2, 1, 64, 64, generator=torch.Generator().manual_seed(2)) + 0.1
    d = torch.tensor([-0.4, -0.6]).view(-1, 1, 1, 1)
    out = Back_prop(args)(holo.to(dev), d.to(dev))
    ref = orc.asm(torch.sqrt(holo).to(torch.complex64), 532e-9, d * 1e-3, 1.5e-6, zero_padding=False)
    assert out.shape == (2, 2, 64, 64)
    assert cplx_rel(torch.complex(out[:, :1].float(), out[:, 1:].float()), ref) < 1e-5


@pytest.mark.parametrize("pn,dn,dc", [(1.0, 1.0, 0.0), (2.0, 0.5, 0.1), (0.5, 2.0, -0.05)])
def test_holo_generator_normalisation_constants(dev, pn, dn, dc):
    """phase_normalize / distance_normalize / distance_normalize_constant (Forward_model.py:9-20), tensor and
    scalar distances, single- and two-channel fields, all three return modes, against the oracle."""
    import types
    from hip_helpers import cplx_rel, orc
    from utils.Forward_model import Holo_Generator
    hp = orc.HoloParams(phase_normalize=pn, distance_normalize=dn, distance_normalize_constant=dc)
    args = types.SimpleNamespace(wavelength=hp.wavelength, pixel_size=hp.pixel_size, distance_normalize=dn,
                                 distance_normalize_constant=dc, phase_normalize=pn)
    hgp = Holo_Generator(args).to(dev)
    g = torch.Generator().manual_seed(int(pn * 10 + dn * 100))
    amp = torch.rand(3, 1, 32, 32, generator=g) + 0.2
    ph = (torch.rand(3, 1, 32, 32, generator=g) - 0.5) * 3
    d = torch.tensor([0.3, 0.5, -0.2]).view(-1, 1, 1, 1)
    ref = orc.holo_generator(hp, amp, ph, d, complex_number=True)
    assert cplx_rel(hgp(amp.to(dev), ph.to(dev), d.to(dev), complex_number=True), ref) < 1e-5
    assert rel_err(hgp(amp.to(dev), ph.to(dev), d.to(dev)).cpu(), orc.holo_generator(hp, amp, ph, d)) < 1e-5
    ref_s = orc.holo_generator(hp, amp, ph, 0.35, complex_number=True)       # Python-float distance
    assert cplx_rel(hgp(amp.to(dev), ph.to(dev), 0.35, complex_number=True), ref_s) < 1e-5
    amp2, ph2 = amp.repeat(1, 2, 1, 1) * torch.tensor([1.0, 0.5]).view(1, 2, 1, 1), ph.repeat(1, 2, 1, 1)
    ref2 = orc.holo_generator(hp, amp2, ph2, d, complex_number=True)          # C = 2: d broadcast over channels
    assert cplx_rel(hgp(amp2.to(dev), ph2.to(dev), d.to(dev), complex_number=True), ref2) < 1e-5


def test_transfer_function_tables_dedupe_distances(hg, dev):
    """Output fields that share a distance share one transfer-function table (64 slots; filled by extra blocks of the
    forward row kernel); fields beyond the slots evaluate H in the column kernel.  Every field of a mixed batch — equal
    distances, distinct ones, more than 64 distinct ones — must come out bit-identical to the same field propagated alone."""
    g = torch.Generator().manual_seed(11)
    n, s = 90, 32
    amp = (torch.rand(n, 1, s, s, generator=g) + 0.2).to(dev)
    ph = (torch.rand(n, 1, s, s, generator=g) * 6 - 3).to(dev)
    d = torch.empty(n)
    d[:10] = 0.3                                   # shared slot 0
    d[10:80] = torch.linspace(-0.9, 0.9, 70)       # 70 distinct: the slots run out at field 64
    d[80:] = d[70:80]                              # repeats of distances that own no slot: inline evaluation
    d = d.view(n, 1, 1, 1).to(dev)
    a_all, p_all = hg(amp, ph, d, return_field=True)
    for i in (0, 9, 10, 63, 64, 79, 80, 89):
        a_i, p_i = hg(amp[i:i + 1], ph[i:i + 1], d[i:i + 1], return_field=True)
        assert torch.equal(a_all[i:i + 1], a_i) and torch.equal(p_all[i:i + 1], p_i), i
    from hip_helpers import orc
    ref = orc.holo_generator(orc.HoloParams(), amp.cpu(), ph.cpu(), d.cpu(), complex_number=True)
    from hip_helpers import cplx_rel
    assert cplx_rel(torch.polar(a_all, p_all), ref) < 1e-5


def test_phase_constants_kernel_reproduces_the_aten_rounding_chain(hg, dev):
    """holo_asm_phase_constants_f32 against the element-wise torch chain it replaces (Forward_model.py:18 then
    Angular_Spectrum_Method.py:29): bit-identical float64 constants for random distances, offsets and scales, per-sample,
    broadcast and sweep (period) forms — and the propagation through Holo_Generator is unchanged by it."""
    from style_transfer_based_holographic_imaging_amd import ops
    import types
    g = torch.Generator().manual_seed(17)
    for add, mul in ((0.0, 1.0), (0.137, 2.5), (-0.05, 0.731)):
        d = ((torch.rand(37, generator=g) - 0.5) * 3.0).to(dev)
        want = ops.phase_constant(((d + add) * mul) * 1e-3, 37, dev)
        got = ops.phase_constant_from_distance(d, add, mul, 37, 37)
        assert got.dtype == torch.float64 and torch.equal(got, want)
        sweep = ops.phase_constant_from_distance(d[:8].contiguous(), add, mul, 5 * 8, 8)
        assert torch.equal(sweep, want[:8].repeat(5))
        one = ops.phase_constant_from_distance(d[3:4], add, mul, 6, 6)
        assert torch.equal(one, want[3:4].expand(6))
    hp = types.SimpleNamespace(wavelength=532e-9, pixel_size=1.5e-6, distance_normalize=0.8, distance_normalize_constant=0.05, phase_normalize=1.0)
    from utils.Forward_model import Holo_Generator
    gen = Holo_Generator(hp)
    amp, ph = (torch.rand(3, 1, 64, 64, generator=g) + 0.3).to(dev), torch.rand(3, 1, 64, 64, generator=g).to(dev)
    d3 = torch.tensor([0.4, -0.2, 0.7]).view(-1, 1, 1, 1).to(dev)
    fast = gen(amp, ph, d3, return_field=True)
    pc = ops.phase_constant(((d3 + 0.05) * 0.8) * 1e-3, 3, dev)                      # the torch chain, fed to the low-level launcher
    from style_transfer_based_holographic_imaging_amd import _lib
    ref = ops.asm_propagate(amp, ph, pc, 64, True, 532e-9, 1.5e-6, 1.0, _lib.ASM_OUT_AMP_PHASE)
    assert torch.equal(fast[0], ref[0]) and torch.equal(fast[1], ref[1])
    slow = gen(amp, ph, d3.double(), return_field=True)        # a float64 tensor takes the torch path (float64 chain: other bits, same field)
    assert float((fast[0] - slow[0]).abs().max()) < 1e-4
