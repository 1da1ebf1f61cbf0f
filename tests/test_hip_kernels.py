"""GPU parity of each HIP kernel against the CPU oracle / a plain torch-CPU fp32 reference of the
same op, called through the C-ABI (ops -> ctypes -> libholo_hip.so).  Tolerances are
max|d|/max|ref| per tensor; fp32 kernels differ from the CPU only by summation order."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from style_transfer_based_holographic_imaging_amd import ops as o
    return o


@pytest.fixture(scope="module")
def dev():
    from hip_helpers import DEV
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return DEV


def _gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("b,h,w,cin,cout,relu,pool", [
    (2, 16, 16, 16, 64, True, False),
    (1, 8, 16, 64, 128, True, False),
    (3, 13, 21, 32, 64, True, True),       # odd sizes: partial tiles + ceil_mode pooling
    (2, 5, 5, 128, 256, True, True),
    (1, 2, 2, 16, 64, False, False),       # smallest legal reflection-padded image, no ReLU
    (2, 40, 24, 64, 64, True, True),
    (1, 33, 47, 256, 128, True, False),
    (2, 16, 16, 512, 512, True, False),
])
def test_conv3x3(ops, dev, b, h, w, cin, cout, relu, pool):
    g = _gen(h * 1000 + w + cin)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), wt, bias)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2, 0, ceil_mode=True)
    xp = ops.to_nhwc(x.to(dev))
    y = ops.conv3x3(xp, ops.pack_conv3x3(wt.to(dev)), bias.to(dev), cout, relu, pool)
    got = ops.as_nchw_view(y).cpu()
    assert got.shape == ref.shape
    # one fp32 fmaf chain over K = 9*Cin terms on the MFMA vs blocked sums on the CPU: grows ~sqrt(K)*eps
    assert rel_err(got, ref) < (5e-6 if cin >= 512 else 2e-6)


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 4, 4, 64, 64), (1, 7, 5, 128, 128), (2, 16, 16, 256, 256), (1, 1, 1, 16, 64)])
def test_convt2x2(ops, dev, b, h, w, cin, cout):
    g = _gen(h + w + cin)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 2, 2, generator=g) * math.sqrt(2.0 / cin)
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv_transpose2d(x, wt, bias, stride=2))
    y = ops.convt2x2(ops.to_nhwc(x.to(dev)), ops.pack_convt2x2(wt.to(dev)), bias.to(dev), cout, True)
    got = ops.as_nchw_view(y).cpu()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 2e-6


@pytest.mark.parametrize("b,h,w", [(2, 16, 16), (1, 13, 37), (3, 2, 2), (1, 128, 128)])
def test_conv_first(ops, dev, b, h, w):
    g = _gen(h * w)
    x = torch.rand(b, 1, h, w, generator=g)
    w0, b0 = torch.randn(3, 1, 1, 1, generator=g), torch.randn(3, generator=g) * 0.1
    w1, b1 = torch.randn(64, 3, 3, 3, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(F.pad(F.conv2d(x, w0, b0), (1, 1, 1, 1), mode="reflect"), w1, b1))
    y = ops.conv_first(x.to(dev), w0.to(dev), b0.to(dev), w1.to(dev), b1.to(dev))
    assert rel_err(ops.as_nchw_view(y).cpu(), ref) < 2e-6
    # intensity in, square root in the loader (test_field_retrieval_mnist.py:119 folded in): the bits of the correctly rounded root first
    # (torch.sqrt on the device gives the same; torch's vectorised HOST sqrt is an ulp off on ~16 % of values), for every
    # activation format (batches of 16 take the bf16 matrix-core form of the layer)
    inten = (x * x * 3.7).to(dev)
    for dt, nb in ((torch.float32, 1), (torch.bfloat16, 1), (torch.bfloat16, 16), ("f32s", 1)):
        xi = inten.repeat(nb, 1, 1, 1)
        args = (w0.to(dev), b0.to(dev), w1.to(dev), b1.to(dev))
        if dt == "f32s":
            dt = ops.compute_dtype("f32s")
        a = ops.conv_first(torch.sqrt(xi.double()).float(), *args, out_dtype=dt)      # the correctly rounded root (= torch.sqrt on the device)
        f = ops.conv_first(xi, *args, out_dtype=dt, sqrt_input=True)
        assert a.dtype == f.dtype and torch.equal(a.view(torch.int16 if a.element_size() == 2 else torch.int32),
                                                   f.view(torch.int16 if f.element_size() == 2 else torch.int32))


@pytest.mark.parametrize("b,h,w", [(2, 16, 16), (1, 19, 35), (2, 2, 2), (1, 64, 64)])
def test_conv_last(ops, dev, b, h, w):
    g = _gen(h * w + 1)
    x = torch.randn(b, 64, h, w, generator=g)
    wt, bias = torch.randn(2, 64, 3, 3, generator=g) * 0.06, torch.randn(2, generator=g) * 0.1
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), wt, bias)
    y = ops.conv_last(ops.to_nhwc(x.to(dev)), wt.to(dev), bias.to(dev))
    assert y.shape == ref.shape and y.is_contiguous()
    assert rel_err(y.cpu(), ref) < 2e-6


@pytest.mark.parametrize("b,c,h,w", [(2, 512, 4, 4), (3, 64, 7, 9), (1, 512, 32, 32), (2, 128, 1, 2)])
def test_mean_std_and_adain(ops, dev, b, c, h, w):
    import function
    from hip_helpers import orc
    g = _gen(c + h)
    x = torch.randn(b, c, h, w, generator=g) * 2 + 1.5
    style = torch.randn(b, c, h, w, generator=g).abs() + 0.1
    m, s = function.calc_mean_std(x.to(dev))
    rm, rs = orc.calc_mean_std(x)
    assert m.shape == rm.shape == (b, c, 1, 1)
    assert rel_err(m.cpu(), rm) < 1e-6 and rel_err(s.cpu(), rs) < 1e-6
    got = function.adaptive_instance_normalization(x.to(dev), style.to(dev))
    assert rel_err(got.cpu(), orc.adaptive_instance_normalization(x, style)) < 2e-6
    # stored-style form with alpha blend (net.py:235-246), single and multiple style
    sv = torch.cat([torch.randn(1, c, 1, 1, generator=g), torch.rand(1, c, 1, 1, generator=g) + 0.2], 0)
    for alpha in (1.0, 0.5, 0.0):
        ref = orc.adain_with_style_vector(x, sv, alpha)
        xm, xs = ops.channel_mean_std(ops.to_nhwc(x.to(dev)))
        y = ops.adain_apply(ops.to_nhwc(x.to(dev)), xm, xs, sv[:1].to(dev), sv[1:].to(dev), alpha)
        assert rel_err(ops.as_nchw_view(y).cpu(), ref) < 2e-6
        # the one-launch form (statistics + AdaIN): the same bits, for fp32 and bf16 activations
        for dt in (torch.float32, torch.bfloat16):
            xa = ops.to_nhwc(x.to(dev)).to(dt).contiguous()
            m2, s2 = ops.channel_mean_std(xa)
            y2 = ops.adain_apply(xa, m2, s2, sv[:1].to(dev), sv[1:].to(dev), alpha)
            yf, mf, sf = ops.adain_fused(xa, sv[:1].to(dev), sv[1:].to(dev), alpha)
            assert torch.equal(yf, y2) and torch.equal(mf, m2) and torch.equal(sf, s2)
    multi = torch.cat([torch.randn(b, c, 1, 1, generator=g), torch.rand(b, c, 1, 1, generator=g) + 0.2], 0)
    if b > 1:
        ref = orc.adain_with_style_vector(x, multi, 1.0)
        xm, xs = ops.channel_mean_std(ops.to_nhwc(x.to(dev)))
        y = ops.adain_apply(ops.to_nhwc(x.to(dev)), xm, xs, multi[:b].to(dev), multi[b:].to(dev), 1.0)
        assert rel_err(ops.as_nchw_view(y).cpu(), ref) < 2e-6
        yf, mf, sf = ops.adain_fused(ops.to_nhwc(x.to(dev)), multi[:b].to(dev), multi[b:].to(dev), 1.0)
        assert torch.equal(yf, y) and torch.equal(mf, xm) and torch.equal(sf, xs)


@pytest.mark.parametrize("b", [1, 3, 8, 19])
def test_distance_head(ops, dev, b, synth_weights):
    from hip_helpers import orc
    dist_sd = synth_weights[2]
    g = _gen(b)
    mu = torch.randn(b, 512, 1, 1, generator=g).abs()
    sd = torch.rand(b, 512, 1, 1, generator=g) + 0.1
    ref = orc.distance_head(mu, sd, dist_sd)
    dsd = {k: v.to(dev) for k, v in dist_sd.items()}
    got = ops.distance_head(mu.view(b, 512).to(dev), sd.view(b, 512).to(dev), dsd)
    assert got.shape == (b, 1)
    assert rel_err(got.cpu(), ref) < 2e-5


def test_distance_head_four_launch_form_against_the_layer_by_layer_one(ops, dev, synth_weights):
    """holo_distance_head_f32 (the normalisation inside the consuming Linear, the concatenation read in place) against the per-layer
    entry points it replaces: same values up to the re-ordered statistics sums; a width it does not take (C = 128) goes layer by layer."""
    dist_sd = synth_weights[2]
    dsd = {k: v.to(dev) for k, v in dist_sd.items()}
    g = _gen(77)
    for b in (1, 5, 8, 9, 64):
        mean, std = torch.randn(b, 512, generator=g).abs().to(dev), (torch.rand(b, 512, generator=g) + 0.1).to(dev)
        got = ops.distance_head(mean, std, dsd)
        x = torch.cat([mean, std], 1)
        for name in ("l1", "l2", "l3"):
            x = ops.rownorm_relu(ops.linear(x, dsd[f"{name}.weight"], dsd[f"{name}.bias"]))
        ref = ops.linear1_sigmoid(x, dsd["out.weight"], dsd["out.bias"])
        assert got.shape == ref.shape == (b, 1)
        assert rel_err(got.cpu(), ref.cpu()) < 2e-6
    c = 128
    sd = {"l1.weight": torch.randn(2 * c, 2 * c, generator=g) * 0.1, "l1.bias": torch.zeros(2 * c), "l2.weight": torch.randn(2 * c, 2 * c, generator=g) * 0.1,
          "l2.bias": torch.zeros(2 * c), "l3.weight": torch.randn(c, 2 * c, generator=g) * 0.1, "l3.bias": torch.zeros(c),
          "out.weight": torch.randn(1, c, generator=g) * 0.1, "out.bias": torch.zeros(1)}
    mean, std = torch.randn(3, c, generator=g).abs(), torch.rand(3, c, generator=g) + 0.1
    got = ops.distance_head(mean.to(dev), std.to(dev), {k: v.to(dev) for k, v in sd.items()})
    x = torch.cat([mean, std], 1)
    for name in ("l1", "l2", "l3"):
        x = torch.relu(torch.nn.functional.instance_norm((x @ sd[f"{name}.weight"].t() + sd[f"{name}.bias"]).unsqueeze(0)).squeeze(0))
    ref = torch.sigmoid(x @ sd["out.weight"].t() + sd["out.bias"])
    assert rel_err(got.cpu(), ref) < 2e-5


def test_validation_distance_head():
    import ctypes
    from style_transfer_based_holographic_imaging_amd._lib import lib
    fake = ctypes.c_void_p(16)
    args = lambda b, c, ws: [fake] * 11 + [b, c, 1e-5, fake, ws, None]      # noqa: E731
    assert lib.holo_distance_head_f32(*([None] + args(2, 512, 1 << 30)[1:])) == 1
    assert lib.holo_distance_head_f32(*args(0, 512, 1 << 30)) == 1
    assert lib.holo_distance_head_f32(*args(2, 128, 1 << 30)) == 2          # c not a multiple of 256
    assert lib.holo_distance_head_f32(*args(2, 768, 1 << 30)) == 2          # 2c > 1024
    assert lib.holo_distance_head_f32(*args(2, 512, 16)) == 3               # workspace
    assert lib.holo_distance_head_workspace_bytes(5, 512) == 2 * 5 * 1024 * 4


def test_layout_roundtrip(ops, dev):
    x = torch.randn(3, 70, 9, 11, generator=_gen(5))
    y = ops.to_nhwc(x.to(dev))
    assert y.shape == (3, 9, 11, 70) and y.is_contiguous()
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.to_nchw_contiguous(y).cpu(), x)


def test_errors_are_loud(ops, dev):
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.channel_mean_std(torch.zeros(1, 2, 2, 64))
    x = torch.zeros(1, 1, 2, 16, device=dev)       # H = 1 < 2
    with pytest.raises(RuntimeError, match="reflection padding"):
        ops.conv3x3(x, torch.zeros(9 * 16 * 64, device=dev), torch.zeros(64, device=dev), 64)
    with pytest.raises(RuntimeError, match="multiple of 16"):
        ops.conv3x3(torch.zeros(1, 4, 4, 24, device=dev), torch.zeros(9 * 24 * 64, device=dev), torch.zeros(64, device=dev), 64)


def test_distance_head_does_not_depend_on_the_position_in_the_batch():
    """A sample's distance is bit-identical wherever it sits in a batch (shards of a multi-GPU job start at arbitrary
    offsets; the linear kernel handles eight samples per wave and each slot must run the same arithmetic)."""
    from hip_helpers import DEV, build_network
    from style_transfer_based_holographic_imaging_amd import ops
    _, sds = build_network(0)
    sd = {k: v.to(DEV) for k, v in sds[2].items()}
    g = torch.Generator().manual_seed(0)
    mean, std = torch.randn(40, 512, generator=g).to(DEV), (torch.rand(40, 512, generator=g) + 0.1).to(DEV)
    whole = ops.distance_head(mean, std, sd)
    for lo, hi in ((0, 7), (4, 11), (8, 16), (20, 40), (33, 40), (39, 40)):
        part = ops.distance_head(mean[lo:hi].contiguous(), std[lo:hi].contiguous(), sd)
        assert torch.equal(part, whole[lo:hi]), (lo, hi)
