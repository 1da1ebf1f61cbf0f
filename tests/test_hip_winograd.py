"""GPU parity of the fp32 Winograd F(2x2,3x3) convolution (csrc/conv_winograd.hip) through the C-ABI: against a plain
torch-CPU fp32 reference and a float64 evaluation of the same op, next to the direct implicit-GEMM kernel; then the whole
network with every large 3x3 layer on the Winograd kernel against the oracle and against float64 truth."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import REPO, rel_err

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


def _case(seed, b, h, w, cin, cout, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, cin, h, w, generator=g) * scale
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    return x, wt, bias


def _ref(x, wt, bias, relu, pool):
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), wt, bias)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2, 0, ceil_mode=True)
    return ref


@pytest.mark.parametrize("b,h,w,cin,cout,relu,pool", [
    (2, 32, 16, 32, 32, True, False),      # exactly one block tile, two chunks (the minimum)
    (1, 8, 16, 64, 128, True, False),
    (3, 13, 21, 32, 64, True, True),       # odd sizes: partial tiles, odd edge tiles + ceil_mode pooling
    (2, 5, 5, 128, 256, True, True),
    (1, 2, 2, 32, 64, False, False),       # smallest legal reflection-padded image, no ReLU
    (2, 40, 24, 64, 64, True, True),
    (1, 33, 47, 256, 128, True, False),
    (2, 16, 16, 512, 512, True, False),
    (1, 70, 35, 64, 96, False, False),     # Cout a multiple of 32 only; three tile rows, partial last ones
    (1, 64, 64, 32, 32, True, True),       # 8 tiles on a 256-block grid: one tile per block
    (5, 96, 80, 32, 64, True, False),      # 150 tiles x 2 channel blocks: some persistent blocks run 2 tiles, some 1
])
def test_conv3x3_winograd(ops, dev, b, h, w, cin, cout, relu, pool):
    x, wt, bias = _case(h * 1000 + w + cin, b, h, w, cin, cout)
    ref = _ref(x, wt, bias, relu, pool)
    ref64 = _ref(x.double(), wt.double(), bias.double(), relu, pool)
    xp = ops.to_nhwc(x.to(dev))
    y = ops.conv3x3_wino(xp, ops.pack_conv3x3_wino(wt.to(dev)), bias.to(dev), cout, relu, pool)
    got = ops.as_nchw_view(y).cpu()
    assert got.shape == ref.shape
    # transforms add a few fp32 roundings to each product; the accumulation chain is Cin long instead of 9*Cin
    assert rel_err(got, ref) < (6e-6 if cin >= 512 else 3e-6)
    if cout % 64 == 0:
        direct = ops.as_nchw_view(ops.conv3x3(xp, ops.pack_conv3x3(wt.to(dev)), bias.to(dev), cout, relu, pool)).cpu()
        e_w, e_d = rel_err(got, ref64), rel_err(direct, ref64)
        assert e_w < 4 * e_d + 1e-7, f"Winograd {e_w:.2e} vs direct {e_d:.2e} against float64"


@pytest.mark.parametrize("scale", [1e-3, 1.0, 1e3])
def test_conv3x3_winograd_scales(ops, dev, scale):
    """relative accuracy does not depend on the activation scale (no fixed-point assumption anywhere)"""
    x, wt, bias = _case(5, 1, 24, 40, 64, 64, scale)
    ref64 = _ref(x.double(), wt.double(), bias.double() * scale, True, False)
    y = ops.conv3x3_wino(ops.to_nhwc(x.to(dev)), ops.pack_conv3x3_wino(wt.to(dev)), (bias * scale).to(dev), 64, True, False)
    assert rel_err(ops.as_nchw_view(y).cpu(), ref64) < 2e-6


def test_packed_winograd_weights_are_g_w_gt(ops, dev):
    g = torch.Generator().manual_seed(1)
    wt = torch.randn(64, 32, 3, 3, generator=g)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    U = (G @ wt.double() @ G.T).float()                                   # (co, ci, 4, 4)
    packed = ops.pack_conv3x3_wino(wt.to(dev)).cpu().view(2, 2, 16, 32, 16)   # [co/32][ci/16][pos][co%32][ci%16]
    want = U.view(2, 32, 2, 16, 16).permute(0, 2, 4, 1, 3)                # co_hi, ci_hi, pos, co_lo, ci_lo
    assert torch.equal(packed, want.contiguous())


def test_validation_winograd():
    import ctypes
    from style_transfer_based_holographic_imaging_amd._lib import lib
    fake = ctypes.c_void_p(16)
    assert lib.holo_conv3x3_wino_f32(None, fake, fake, fake, 1, 4, 4, 32, 32, 1, 0, None) == 1
    assert lib.holo_conv3x3_wino_f32(fake, fake, fake, fake, 1, 4, 4, 24, 32, 1, 0, None) == 2     # Cin % 16
    assert lib.holo_conv3x3_wino_f32(fake, fake, fake, fake, 1, 4, 4, 16, 32, 1, 0, None) == 2     # Cin < 32 (two chunks)
    assert lib.holo_conv3x3_wino_f32(fake, fake, fake, fake, 1, 4, 4, 32, 48, 1, 0, None) == 2     # Cout % 32
    assert lib.holo_conv3x3_wino_f32(fake, fake, fake, fake, 1, 4, 4, 32, 32, 0, 1, None) == 2     # pool without relu
    assert lib.holo_conv3x3_wino_f32(fake, fake, fake, fake, 1, 1, 4, 32, 32, 1, 0, None) == 1     # H < 2
    assert lib.holo_conv3x3_wino_packed_elems(128, 64) == 16 * 64 * 128


@pytest.fixture(scope="module")
def net_and_inputs(dev):
    from hip_helpers import build_network, orc
    from style_transfer_based_holographic_imaging_amd import synth
    network, sds = build_network(0)
    style = torch.load(os.path.join(REPO, "style_representation", "MNIST", "style_vector.pt"), map_location="cpu").float()
    hp = orc.HoloParams()
    ph = synth.synthetic_phase_objects(2, 96, seed=11)
    d_c = synth.synthetic_distances(2, seed=11)
    with torch.no_grad():
        content = torch.sqrt(orc.holo_generator(hp, torch.full_like(ph, 0.6), ph, d_c)).float()
    return network, sds, style, content


def test_whole_network_on_winograd(net_and_inputs, dev):
    """every 3x3 layer the MFMA kernels cover on the Winograd kernel: the graded outputs keep the 1e-4 bar against the CPU
    oracle with the same margin as the direct kernel, and sit as close to a float64 evaluation as the direct path"""
    from hip_helpers import orc
    from utils.functions import field_retrieval
    network, sds, style, content = net_and_inputs
    dbl = [{k: v.double() for k, v in sd.items()} for sd in sds]
    with torch.no_grad():
        want = orc.net_field_retrieval(content, style, *sds, 1.0, True)
        truth = orc.net_field_retrieval(content.double(), style.double(), *dbl, 1.0, True)
        outs = {}
        for algo in ("direct", "winograd"):
            network.set_conv_algo(algo)
            outs[algo] = [t.cpu() for t in field_retrieval(network, content.to(dev), style.to(dev), 1.0, True)]
        network.set_conv_algo("auto")
    for name, i in (("amplitude", 0), ("phase", 1), ("distance", 2)):
        w, d = outs["winograd"][i].view(-1), outs["direct"][i].view(-1)
        assert rel_err(w, want[i].reshape(-1)) < 1e-4 / 3, name
        e_w, e_d = rel_err(w, truth[i].reshape(-1)), rel_err(d, truth[i].reshape(-1))
        assert e_w < 3 * e_d + 1e-7, f"{name}: Winograd {e_w:.2e} vs direct {e_d:.2e} against float64"


def test_auto_picks_winograd_only_for_layers_that_fill_the_chip(dev):
    from style_transfer_based_holographic_imaging_amd.engine import Plan
    import net
    plan = Plan(list(net.decoder.children()))
    conv = next(s for s in plan.steps if s.kind == "conv")
    big = torch.empty(64, 32, 32, conv.conv.in_channels, device="meta")
    small = torch.empty(2, 16, 16, conv.conv.in_channels, device="meta")
    assert plan._use_winograd(conv, big)
    assert plan._wino_choice(conv, small) == (True, 8)                   # 512 -> 512: 32 items x 8 parts of 4 chunks = one full round
    assert plan._wino_choice(conv, torch.empty(5, 16, 16, conv.conv.in_channels, device="meta")) == (True, 3)   # 80 items x 3 parts
    c256 = next(s for s in plan.steps if s.kind == "conv" and s.conv.in_channels == 256 and s.conv.out_channels == 256)
    assert not plan._use_winograd(c256, torch.empty(1, 16, 16, 256, device="meta"))   # 8 items: no admissible split reaches half a wave
    plan.conv_algo = "direct"
    assert not plan._use_winograd(conv, big)
    plan.conv_algo = "winograd"
    assert plan._use_winograd(conv, small)
    plan.dtype = torch.bfloat16
    assert not plan._use_winograd(conv, big)


@pytest.mark.parametrize("b,h,w,cin,cout,relu,pool,ksplit", [
    (5, 32, 32, 256, 256, True, False, 2),     # the bundled shape's deep layers (5 x 128^2: 32 x 32 and 16 x 16 feature maps)
    (5, 16, 16, 512, 256, True, False, 4),
    (5, 16, 16, 256, 512, True, False, 2),
    (3, 13, 21, 64, 64, True, True, 2),        # partial tiles, ceil-mode pooling in the reduction
    (2, 5, 5, 128, 32, True, True, 4),
    (1, 2, 2, 64, 64, False, False, 2),        # smallest legal image, no activation
    (2, 40, 24, 256, 96, False, False, 8),     # 16 chunks in 8 parts of 2: the smallest part the entry point accepts
    (5, 32, 32, 256, 256, True, False, 3),     # uneven parts (what `auto` picks at 5 x 128^2): 16 chunks as 5 + 5 + 6
    (5, 16, 16, 512, 256, True, False, 6),     # 32 chunks as 5 5 6 5 5 6
    (5, 16, 16, 512, 512, True, False, 3),     # 10 + 11 + 11
    (3, 20, 18, 112, 64, True, True, 3),       # 7 chunks as 2 + 2 + 3, partial tiles, pooling
    (2, 9, 33, 80, 32, False, False, 2),       # 5 chunks as 2 + 3
])
def test_conv3x3_winograd_splitk(ops, dev, b, h, w, cin, cout, relu, pool, ksplit):
    """Split-K: the channel chunks of a tile dealt to `ksplit` work items + the fixed-order reduction (bias, ReLU, pooling in
    the reduction).  Same accuracy bar as the un-split kernel against torch-CPU fp32; bit-identical per hologram whatever the
    batch it arrives in."""
    x, wt, bias = _case(h * 1000 + w + cin + ksplit, b, h, w, cin, cout)
    ref = _ref(x, wt, bias, relu, pool)
    xp = ops.to_nhwc(x.to(dev))
    wp = ops.pack_conv3x3_wino(wt.to(dev))
    y = ops.conv3x3_wino(xp, wp, bias.to(dev), cout, relu, pool, ksplit=ksplit)
    got = ops.as_nchw_view(y).cpu()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < (6e-6 if cin >= 512 else 3e-6)
    plain = ops.as_nchw_view(ops.conv3x3_wino(xp, wp, bias.to(dev), cout, relu, pool)).cpu()
    assert rel_err(got, plain) < 2e-6                                   # re-associated sum over channels: rounding only
    if b > 1:
        one = ops.conv3x3_wino(xp[1:2].contiguous(), wp, bias.to(dev), cout, relu, pool, ksplit=ksplit)
        assert torch.equal(one, y[1:2])


def test_validation_winograd_splitk():
    import ctypes
    from style_transfer_based_holographic_imaging_amd._lib import lib
    fake = ctypes.c_void_p(16)
    big = 1 << 40
    args = lambda cin, cout, k, ws=big: (fake, fake, fake, fake, 1, 4, 4, cin, cout, 1, 0, k, fake, ws, None)  # noqa: E731
    assert lib.holo_conv3x3_wino_splitk_f32(None, fake, fake, fake, 1, 4, 4, 64, 32, 1, 0, 2, fake, big, None) == 1
    assert lib.holo_conv3x3_wino_splitk_f32(*args(64, 32, 1)) == 2          # ksplit < 2
    assert lib.holo_conv3x3_wino_splitk_f32(*args(64, 32, 3)) == 2          # 4 chunks in 3 parts: one part of a single chunk
    assert lib.holo_conv3x3_wino_splitk_f32(*args(64, 32, 4)) == 2          # parts of one chunk
    assert lib.holo_conv3x3_wino_splitk_f32(*args(512, 32, 9)) == 2         # ksplit > 8
    assert lib.holo_conv3x3_wino_splitk_f32(*args(96, 48, 2)) == 2          # Cout % 32
    assert lib.holo_conv3x3_wino_splitk_f32(*args(64, 32, 2, 16)) == 3      # workspace
    assert lib.holo_conv3x3_wino_splitk_workspace_bytes(5, 32, 32, 256, 2) == 2 * 5 * 32 * 32 * 256 * 4


def test_random_shapes_all_three_kernels(ops, dev):
    """Seeded sweep over 36 random layer shapes (sizes 2..75 incl. odd ones, every channel count the kernels accept, batch 1..6,
    every admissible split): direct kernel, Winograd and split-K Winograd against torch-CPU fp32, and each against the others."""
    import random
    rng = random.Random(20261004)
    for case in range(36):
        b = rng.randint(1, 6)
        h, w = rng.randint(2, 75), rng.randint(2, 75)
        cin = 16 * rng.choice([2, 3, 4, 5, 7, 8, 12, 16])
        cout = 64 * rng.choice([1, 1, 2, 3, 4])
        relu = rng.random() < 0.8
        pool = relu and rng.random() < 0.4
        x, wt, bias = _case(9000 + case, b, h, w, cin, cout, scale=rng.choice([0.1, 1.0, 30.0]))
        ref = _ref(x, wt, bias, relu, pool)
        xp, bd = ops.to_nhwc(x.to(dev)), bias.to(dev)
        ww, wd = ops.pack_conv3x3_wino(wt.to(dev)), ops.pack_conv3x3(wt.to(dev), torch.float32)
        tag = f"case {case}: b={b} {h}x{w} {cin}->{cout} relu={relu} pool={pool}"
        outs = {"direct": ops.conv3x3(xp, wd, bd, cout, relu, pool), "wino": ops.conv3x3_wino(xp, ww, bd, cout, relu, pool)}
        ks = [k for k in range(2, 9) if (cin // 16) // k >= 2]
        if ks:
            k = rng.choice(ks)
            outs[f"splitk{k}"] = ops.conv3x3_wino(xp, ww, bd, cout, relu, pool, ksplit=k)
        for name, y in outs.items():
            got = ops.as_nchw_view(y).cpu()
            assert got.shape == ref.shape, f"{tag} {name}"
            assert rel_err(got, ref) < 4e-6, f"{tag} {name}: {rel_err(got, ref):.2e}"
