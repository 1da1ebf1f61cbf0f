"""GPU checks of the opt-in "f32s" compute mode: fp32 values carried as two fp16 halves (hi + lo' * 2^-11, planar per
16 channels) so that the conv stack runs on the fp16 matrix cores with fp32 accumulation.  It is NOT fp32 arithmetic
(22-23 significant bits per operand, dropped lo'*lo' term), so it is held to what matters: float64 truth, where it
must be at least as close as the fp32 kernels, and the same 1e-4 bar against the CPU fp32 oracle for the whole path."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from style_transfer_based_holographic_imaging_amd import ops as o
    assert torch.cuda.is_available()
    return o


@pytest.fixture(scope="module")
def dev():
    from hip_helpers import DEV
    return DEV


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def test_cast_round_trip_precision_and_saturation(ops, dev):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(3, 5, 7, 64, generator=g) * torch.logspace(-3, 2, 64)).to(dev)
    xs = ops.cast(x, ops.F32S)
    assert xs.dtype == ops.F32S and xs.shape == x.shape
    back = ops.cast(xs, torch.float32)
    # hi (11 bits) + lo' (11 bits): 2^-22 relative while both halves are normal fp16 numbers, and an absolute floor of
    # 2^-25 / 2048 = 1.5e-11 from the subnormal spacing of lo' for |a| below ~1e-4
    assert torch.all((back - x).abs() <= 2.0 ** -21 * x.abs() + 2e-11)
    big = torch.full((1, 1, 1, 16), 1e6, device=dev)
    assert torch.all(ops.cast(ops.cast(big, ops.F32S), torch.float32) == 65504.0)   # documented saturation


@pytest.mark.parametrize("b,h,w,cin,cout,relu,pool", [
    (2, 16, 16, 32, 64, True, False), (1, 13, 21, 64, 128, True, True), (2, 5, 5, 128, 256, True, True),
    (1, 2, 2, 32, 64, False, False), (2, 24, 40, 64, 64, True, True), (1, 16, 16, 512, 512, True, False),
])
def test_conv3x3_f32s_against_float64(ops, dev, b, h, w, cin, cout, relu, pool):
    g = torch.Generator().manual_seed(h * 100 + cin)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), wt.double(), bias.double())
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2, 0, ceil_mode=True)
    x_nhwc = ops.to_nhwc(x.to(dev))
    y32 = ops.conv3x3(x_nhwc, ops.pack_conv3x3(wt.to(dev)), bias.to(dev), cout, relu, pool)
    ys = ops.conv3x3(ops.cast(x_nhwc, ops.F32S), ops.pack_conv3x3(wt.to(dev), ops.F32S), bias.to(dev), cout, relu, pool)
    assert ys.dtype == ops.F32S
    got = ops.as_nchw_view(ops.cast(ys, torch.float32)).cpu()
    e_s, e_32 = rel(got, ref), rel(ops.as_nchw_view(y32).cpu(), ref)
    assert got.shape == ref.shape and e_s < 1.5e-6 and e_s < 2.0 * e_32 + 2e-7


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 4, 4, 64, 64), (1, 7, 5, 128, 128), (1, 16, 16, 256, 256)])
def test_convt2x2_f32s_against_float64(ops, dev, b, h, w, cin, cout):
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 2, 2, generator=g) * math.sqrt(2.0 / cin)
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv_transpose2d(x.double(), wt.double(), bias.double(), stride=2))
    xs = ops.cast(ops.to_nhwc(x.to(dev)), ops.F32S)
    y = ops.convt2x2(xs, ops.pack_convt2x2(wt.to(dev), ops.F32S), bias.to(dev), cout, True)
    assert rel(ops.as_nchw_view(ops.cast(y, torch.float32)).cpu(), ref) < 1.5e-6


def test_first_last_stats_adain_f32s(ops, dev):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 1, 19, 35, generator=g)
    w0, b0 = torch.randn(3, 1, 1, 1, generator=g), torch.randn(3, generator=g) * 0.1
    w1, b1 = torch.randn(64, 3, 3, 3, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1
    ref = F.relu(F.conv2d(F.pad(F.conv2d(x, w0, b0), (1, 1, 1, 1), mode="reflect"), w1, b1))
    y = ops.conv_first(x.to(dev), w0.to(dev), b0.to(dev), w1.to(dev), b1.to(dev), out_dtype=ops.F32S)
    assert y.dtype == ops.F32S and rel(ops.as_nchw_view(ops.cast(y, torch.float32)).cpu(), ref) < 2e-6
    xl = torch.randn(2, 64, 19, 35, generator=g)
    wl, bl = torch.randn(2, 64, 3, 3, generator=g) * 0.06, torch.randn(2, generator=g) * 0.1
    refl = F.conv2d(F.pad(xl, (1, 1, 1, 1), mode="reflect"), wl, bl)
    yl = ops.conv_last(ops.cast(ops.to_nhwc(xl.to(dev)), ops.F32S), wl.to(dev), bl.to(dev))
    assert yl.dtype == torch.float32 and rel(yl.cpu(), refl) < 2e-6
    f = torch.randn(2, 512, 6, 5, generator=g) * 2 + 0.5
    fs = ops.cast(ops.to_nhwc(f.to(dev)), ops.F32S)
    mean, std = ops.channel_mean_std(fs)
    m_ref, s_ref = f.mean(dim=(2, 3)), (f.var(dim=(2, 3)) + 1e-5).sqrt()
    assert rel(mean.cpu(), m_ref) < 2e-6 and rel(std.cpu(), s_ref) < 2e-6
    sm, ss = torch.randn(1, 512, generator=g), torch.rand(1, 512, generator=g) + 0.5
    t = ops.adain_apply(fs, mean, std, sm.to(dev), ss.to(dev), 0.5)
    t_ref = 0.5 * ((f - m_ref[..., None, None]) / s_ref[..., None, None] * ss[..., None, None] + sm[..., None, None]) + 0.5 * f
    assert t.dtype == ops.F32S and rel(ops.as_nchw_view(ops.cast(t, torch.float32)).cpu(), t_ref) < 2e-6


@pytest.mark.parametrize("s,b", [(128, 2), (72, 1), (100, 1)])
def test_whole_path_f32s(dev, s, b):
    """Same bar as the fp32 path against the CPU fp32 oracle (1e-4), and no further from float64 than the fp32 path."""
    from hip_helpers import build_network, orc
    from utils.functions import field_retrieval
    network, sds = build_network(0)
    style = torch.load("style_representation/MNIST/style_vector.pt", map_location="cpu").float()
    g = torch.Generator().manual_seed(s)
    content = torch.rand(b, 1, s, s, generator=g) * 0.8 + 0.1
    sds64 = [{k: v.double() for k, v in sd.items()} for sd in sds]
    with torch.no_grad():
        a_r, p_r, d_r = orc.field_retrieval(content, style, *sds, 1.0, True)
        a64, p64, _ = orc.field_retrieval(content.double(), style.double(), *sds64, 1.0, True)
        a32, p32, _ = field_retrieval(network, content.to(dev), style.to(dev), 1.0, True)
        network.set_compute_dtype("f32s")
        a, p, d = field_retrieval(network, content.to(dev), style.to(dev), 1.0, True)
        feats = network.encode(content.to(dev))
    assert a.dtype == torch.float32 and feats.dtype == torch.float32          # fp32 at the module boundary
    assert rel(a.cpu(), a_r) < 1e-4 and rel(p.cpu(), p_r) < 1e-4 and rel(d.cpu(), d_r) < 1e-4
    assert rel(a.cpu(), a64) < 1.5 * rel(a32.cpu(), a64) + 1e-6 and rel(p.cpu(), p64) < 1.5 * rel(p32.cpu(), p64) + 1e-6
    assert rel(feats.cpu(), orc.encode(content, sds[0])) < 1e-4


@pytest.mark.parametrize("scale", [1e-3, 1.0, 300.0])
def test_conv3x3_f32s_dynamic_range(ops, dev, scale):
    """Activations three orders of magnitude smaller / larger than unit scale: the split keeps 2^-22 relative precision
    while both halves are normal fp16 numbers (|a| >= ~1e-4 ... 65504), so the layer error relative to its own output
    scale does not move; only below ~1e-4 does the absolute floor of lo' (1.5e-11) show."""
    g = torch.Generator().manual_seed(11)
    b, h, w, cin, cout = 2, 12, 20, 128, 128
    x = torch.randn(b, cin, h, w, generator=g) * scale
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    bias = torch.zeros(cout)
    ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), wt.double())
    xs = ops.cast(ops.to_nhwc(x.to(dev)), ops.F32S)
    y = ops.conv3x3(xs, ops.pack_conv3x3(wt.to(dev), ops.F32S), bias.to(dev), cout, False, False)
    err = rel(ops.as_nchw_view(ops.cast(y, torch.float32)).cpu(), ref)
    assert err < (1e-6 if scale >= 1.0 else 4e-6)
