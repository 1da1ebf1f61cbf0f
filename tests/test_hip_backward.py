"""GPU: the backward pass (csrc/backward.hip + autograd.py) against torch.autograd on the CPU — every step
Function on small shapes with odd sizes, pooling and partial tiles, then the whole training objective (Net.forward,
net.py:199-226) against the reference's own gradients (tests/golden/golden_backward.npz, made by importing the reference)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ag():
    from style_transfer_based_holographic_imaging_amd import autograd
    return autograd


def _nhwc(t, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dev)


@pytest.mark.parametrize("b,h,w,cin,cout,relu,pool", [
    (2, 12, 10, 64, 64, True, False), (1, 13, 9, 64, 128, True, True), (2, 8, 8, 128, 64, False, False),
    (1, 34, 18, 64, 64, True, True), (3, 5, 7, 256, 128, True, False), (1, 2, 2, 64, 64, True, True),
    # widths of whole 32-pixel strips: the weight gradient runs on the LDS-staged kernel (one, two and three strips; row groups)
    (2, 6, 32, 64, 64, True, False), (1, 9, 64, 128, 64, True, True), (1, 37, 96, 64, 128, False, False), (1, 2, 32, 64, 64, True, False),
])
def test_conv_step(ag, dev, b, h, w, cin, cout, relu, pool):
    g = torch.Generator().manual_seed(h * 31 + cin)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), bias.clone().requires_grad_()
    y = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr, br)
    if relu:
        y = F.relu(y)
    if pool:
        y = F.max_pool2d(y, 2, 2, 0, ceil_mode=True)
    up = torch.randn(y.shape, generator=g)
    (y * up).sum().backward()
    xd, wd, bd = _nhwc(x, dev).requires_grad_(), wt.to(dev).requires_grad_(), bias.to(dev).requires_grad_()
    yd = ag.ConvStep.apply(xd, wd, bd, relu, pool)
    assert rel_err(yd.detach().permute(0, 3, 1, 2).cpu(), y.detach()) < 1e-5
    (yd * _nhwc(up, dev)).sum().backward()
    assert rel_err(xd.grad.permute(0, 3, 1, 2).cpu(), xr.grad) < 2e-5
    assert rel_err(wd.grad.cpu(), wr.grad) < 2e-5
    assert rel_err(bd.grad.cpu(), br.grad) < 2e-5


def test_conv_step_accuracy_against_float64(ag, dev):
    """a mid-size layer (2 x 128 -> 128 channels at 48 x 64, ReLU + pool) against float64 autograd: how close the fp32 MFMA
    gradients are to the truth, not to another fp32 summation order (the LDS-staged weight gradient sums 6144 pixels per entry)"""
    g = torch.Generator().manual_seed(64)
    b, cin, cout, h, w = 2, 128, 128, 48, 64
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(), wt.double().requires_grad_(), bias.double().requires_grad_()
    y = F.max_pool2d(F.relu(F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr, br)), 2, 2, 0, ceil_mode=True)
    up = torch.randn(y.shape, generator=g)
    (y * up.double()).sum().backward()
    xd, wd, bd = _nhwc(x, dev).requires_grad_(), wt.to(dev).requires_grad_(), bias.to(dev).requires_grad_()
    yd = ag.ConvStep.apply(xd, wd, bd, True, True)
    (yd * _nhwc(up, dev)).sum().backward()
    errs = (rel_err(xd.grad.permute(0, 3, 1, 2).cpu(), xr.grad), rel_err(wd.grad.cpu(), wr.grad), rel_err(bd.grad.cpu(), br.grad))
    print("relative error against float64 (dx, dw, db): %.1e %.1e %.1e" % errs)
    assert max(errs) < 3e-6


@pytest.mark.parametrize("gx,b,h,w,cin,cout,pool", [(2, 4, 6, 32, 64, 64, False), (3, 5, 9, 10, 64, 128, True), (4, 4, 8, 32, 128, 64, False),
                                                    (2, 3, 33, 64, 64, 64, True)])
def test_conv_step_in_a_mosaic(ag, dev, monkeypatch, gx, b, h, w, cin, cout, pool):
    """the embedded gradients laid out gx images per mosaic row (shared zero rows AND columns; the last row ragged where gx does
    not divide the batch): same gradients as torch, through both weight-gradient kernels"""
    monkeypatch.setattr(ag, "_mosaic_per_row", lambda *a: gx)
    g = torch.Generator().manual_seed(gx * 100 + b)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), bias.clone().requires_grad_()
    y = F.relu(F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr, br))
    if pool:
        y = F.max_pool2d(y, 2, 2, 0, ceil_mode=True)
    up = torch.randn(y.shape, generator=g)
    (y * up).sum().backward()
    xd, wd, bd = _nhwc(x, dev).requires_grad_(), wt.to(dev).requires_grad_(), bias.to(dev).requires_grad_()
    yd = ag.ConvStep.apply(xd, wd, bd, True, pool)
    (yd * _nhwc(up, dev)).sum().backward()
    assert rel_err(xd.grad.permute(0, 3, 1, 2).cpu(), xr.grad) < 2e-5
    assert rel_err(wd.grad.cpu(), wr.grad) < 2e-5
    assert rel_err(bd.grad.cpu(), br.grad) < 2e-5


@pytest.mark.parametrize("b,h,w,cin,cout,relu", [(2, 5, 7, 64, 64, False), (1, 9, 20, 256, 128, True), (1, 3, 3, 1024, 256, False),
                                                 (3, 17, 16, 512, 192, True)])
def test_conv1x1(dev, b, h, w, cin, cout, relu):
    """the plain 1x1 convolution on the transposed convolution's one-tap GEMM (64- and 128-wide column tiles, partial pixel tiles)"""
    from style_transfer_based_holographic_imaging_amd import ops
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(b, h, w, cin, generator=g)
    wt = torch.randn(cin, cout, generator=g) / math.sqrt(cin)
    bias = torch.randn(cout, generator=g)
    want = torch.einsum("bhwk,kn->bhwn", x.double(), wt.double()) + bias.double()
    if relu:
        want = want.clamp_min(0)
    got = ops.conv1x1(x.to(dev), wt.to(dev), bias.to(dev), relu)
    assert got.shape == want.shape and rel_err(got.cpu(), want) < 2e-6


@pytest.mark.parametrize("b,h,w,mc,nc", [(2, 11, 32, 64, 64), (1, 40, 64, 128, 64), (3, 7, 32, 64, 192)])
def test_pix_gemm_kernels_agree(dev, monkeypatch, b, h, w, mc, nc):
    """the LDS-staged 9-tap kernel against the operands-from-global one (HOLO_PIX_GEMM_DIRECT=1) and a float64 einsum"""
    from style_transfer_based_holographic_imaging_amd import ops
    g = torch.Generator().manual_seed(w + mc + h)
    pgrad = torch.randn(b, h + 4, w + 4, mc, generator=g).to(dev)
    q = torch.randn(b, h, w, nc, generator=g).to(dev)
    staged = ops.pix_gemm(pgrad, 2, q, 9)
    monkeypatch.setenv("HOLO_PIX_GEMM_DIRECT", "1")
    direct = ops.pix_gemm(pgrad, 2, q, 9)
    monkeypatch.delenv("HOLO_PIX_GEMM_DIRECT")
    qp = F.pad(q.permute(0, 3, 1, 2).double().cpu(), (1, 1, 1, 1), mode="reflect")
    pi = pgrad[:, 2:-2, 2:-2, :].double().cpu()
    want = torch.stack([torch.einsum("bhwm,bnhw->mn", pi, qp[:, :, t // 3:t // 3 + h, t % 3:t % 3 + w]) for t in range(9)], -1)
    assert rel_err(staged.view(mc, nc, 9).cpu(), want) < 2e-6
    assert rel_err(direct.view(mc, nc, 9).cpu(), want) < 2e-6
    assert torch.equal(staged, ops.pix_gemm(pgrad, 2, q, 9))          # fixed summation order: run to run bit-identical


@pytest.mark.parametrize("b,h,w,cin,cout,relu", [(2, 6, 5, 64, 64, True), (1, 7, 9, 128, 128, True), (1, 4, 4, 256, 256, False)])
def test_convt_step(ag, dev, b, h, w, cin, cout, relu):
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 2, 2, generator=g) * math.sqrt(2.0 / cin)
    bias = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), bias.clone().requires_grad_()
    y = F.conv_transpose2d(xr, wr, br, stride=2)
    if relu:
        y = F.relu(y)
    up = torch.randn(y.shape, generator=g)
    (y * up).sum().backward()
    xd, wd, bd = _nhwc(x, dev).requires_grad_(), wt.to(dev).requires_grad_(), bias.to(dev).requires_grad_()
    yd = ag.ConvTStep.apply(xd, wd, bd, relu)
    (yd * _nhwc(up, dev)).sum().backward()
    assert rel_err(xd.grad.permute(0, 3, 1, 2).cpu(), xr.grad) < 2e-5
    assert rel_err(wd.grad.cpu(), wr.grad) < 2e-5
    assert rel_err(bd.grad.cpu(), br.grad) < 2e-5


@pytest.mark.parametrize("b,h,w", [(2, 12, 10), (1, 33, 17), (1, 2, 3)])
def test_first_and_last_step(ag, dev, b, h, w):
    g = torch.Generator().manual_seed(h)
    x = torch.rand(b, 1, h, w, generator=g)
    w0, b0 = torch.randn(3, 1, 1, 1, generator=g), torch.randn(3, generator=g) * 0.1
    w1, b1 = torch.randn(64, 3, 3, 3, generator=g) * math.sqrt(2.0 / 27), torch.randn(64, generator=g) * 0.1
    xr = x.clone().requires_grad_()
    y = F.relu(F.conv2d(F.pad(F.conv2d(xr, w0, b0), (1, 1, 1, 1), mode="reflect"), w1, b1))
    up = torch.randn(y.shape, generator=g)
    (y * up).sum().backward()
    xd = x.to(dev).requires_grad_()
    yd = ag.FirstStep.apply(xd, w0.to(dev), b0.to(dev), w1.to(dev), b1.to(dev))
    (yd * _nhwc(up, dev)).sum().backward()
    assert rel_err(xd.grad.cpu(), xr.grad) < 2e-5
    # last layer
    xl = torch.randn(b, 64, h, w, generator=g)
    wl, bl = torch.randn(2, 64, 3, 3, generator=g) * 0.06, torch.randn(2, generator=g) * 0.1
    xlr, wlr, blr = xl.clone().requires_grad_(), wl.clone().requires_grad_(), bl.clone().requires_grad_()
    yl = F.conv2d(F.pad(xlr, (1, 1, 1, 1), mode="reflect"), wlr, blr)
    upl = torch.randn(yl.shape, generator=g)
    (yl * upl).sum().backward()
    xld, wld, bld = _nhwc(xl, dev).requires_grad_(), wl.to(dev).requires_grad_(), bl.to(dev).requires_grad_()
    yld = ag.LastStep.apply(xld, wld, bld)
    (yld * upl.to(dev)).sum().backward()
    assert rel_err(xld.grad.permute(0, 3, 1, 2).cpu(), xlr.grad) < 2e-5
    assert rel_err(wld.grad.cpu(), wlr.grad) < 2e-5 and rel_err(bld.grad.cpu(), blr.grad) < 2e-5


def test_mean_std_of_a_large_map_splits_the_pixels(ag, dev):
    """above ops.WIDE_STATS_MIN_HW pixels MeanStd runs the pixel-split kernel: same values as the one-block-per-image kernel"""
    from style_transfer_based_holographic_imaging_amd import ops
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(2, 130, 131, 64, generator=g) * 1.5 + 0.7).to(dev)          # 17030 pixels: 17 parts, the last one partial
    assert x.shape[1] * x.shape[2] > ops.WIDE_STATS_MIN_HW
    m0, s0 = ops.channel_mean_std(x)
    m1, s1 = ops.channel_mean_std_wide(x)
    ref = x.double().cpu().view(2, -1, 64)
    assert rel_err(m1.cpu(), ref.mean(1)) < 1e-6 and rel_err(s1.cpu(), (ref.var(1) + 1e-5).sqrt()) < 1e-6
    assert (m0 - m1).abs().max() <= 1.2e-7 * m0.abs().max() and (s0 - s1).abs().max() <= 1.2e-7 * s0.abs().max()   # one ulp of fp32
    xd = x.clone().requires_grad_()
    md, sd_ = ag.MeanStd.apply(xd, 1e-5)
    assert torch.equal(md, m1) and torch.equal(sd_, s1)


def test_mean_std_and_distance_head(ag, dev):
    import sys
    from conftest import REPO
    import os
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import holo_oracle as orc
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 128, 6, 7, generator=g) * 2 + 1
    xr = x.clone().requires_grad_()
    m, s = orc.calc_mean_std(xr)
    um, us = torch.randn(m.shape, generator=g), torch.randn(s.shape, generator=g)
    ((m * um).sum() + (s * us).sum()).backward()
    xd = _nhwc(x, dev).requires_grad_()
    md, sd_ = ag.MeanStd.apply(xd, 1e-5)
    ((md * um.view(3, 128).to(dev)).sum() + (sd_ * us.view(3, 128).to(dev)).sum()).backward()
    assert rel_err(xd.grad.permute(0, 3, 1, 2).cpu(), xr.grad) < 2e-5
    # distance head: against the reference-shaped torch module in eval mode (Dropout = identity)
    import net
    torch.manual_seed(3)
    head = net.Distance_G().eval()
    for p in head.parameters():
        torch.nn.init.normal_(p, std=0.05)
    mean, std = torch.randn(5, 512, 1, 1, generator=g), torch.rand(5, 512, 1, 1, generator=g) + 0.3
    # CPU reference: the same arithmetic with torch ops (net.py:288-308)
    ps = {k: v.detach().clone().requires_grad_() for k, v in head.state_dict().items()}
    mr, sr = mean.clone().requires_grad_(), std.clone().requires_grad_()
    h = torch.cat([mr, sr], 1).view(5, 1024)
    for name in ("l1", "l2", "l3"):
        z = F.linear(h, ps[f"{name}.weight"], ps[f"{name}.bias"])
        z = (z - z.mean(1, keepdim=True)) / torch.sqrt(z.var(1, unbiased=False, keepdim=True) + 1e-5)
        h = F.relu(z)
    yr = torch.sigmoid(F.linear(h, ps["out.weight"], ps["out.bias"]))
    uy = torch.randn(5, 1, generator=g)
    (yr * uy).sum().backward()
    head = head.to(dev)
    md2, sd2 = mean.to(dev).requires_grad_(), std.to(dev).requires_grad_()
    yd = head((md2, sd2))
    assert rel_err(yd.detach().cpu(), yr.detach()) < 1e-5
    (yd * uy.to(dev)).sum().backward()
    assert rel_err(md2.grad.cpu(), mr.grad) < 5e-5 and rel_err(sd2.grad.cpu(), sr.grad) < 5e-5
    for k, v in head.named_parameters():
        assert rel_err(v.grad.cpu(), ps[k].grad) < 5e-5, k


def test_distance_head_training_mode_dropout(ag, dev):
    """Distance_G in training mode (net.py:298-304: Dropout on each Linear's output, before the row norm): given the drawn masks,
    values and all gradients against the same arithmetic in torch; and the module itself draws masks from torch's generator."""
    import net
    g = torch.Generator().manual_seed(21)
    torch.manual_seed(5)
    head = net.Distance_G()
    for p in head.parameters():
        torch.nn.init.normal_(p, std=0.05)
    b = 6
    mean, std = torch.randn(b, 512, generator=g), torch.rand(b, 512, generator=g) + 0.3
    keep = [(torch.rand(b, n, generator=g) < 0.5).float() for n in (1024, 1024, 512)]
    ps = {k: v.detach().clone().requires_grad_() for k, v in head.state_dict().items()}
    mr, sr = mean.clone().requires_grad_(), std.clone().requires_grad_()
    h = torch.cat([mr, sr], 1)
    for name, kp in zip(("l1", "l2", "l3"), keep):
        z = F.linear(h, ps[f"{name}.weight"], ps[f"{name}.bias"]) * kp * 2.0
        z = (z - z.mean(1, keepdim=True)) / torch.sqrt(z.var(1, unbiased=False, keepdim=True) + 1e-5)
        h = F.relu(z)
    yr = torch.sigmoid(F.linear(h, ps["out.weight"], ps["out.bias"]))
    uy = torch.randn(b, 1, generator=g)
    (yr * uy).sum().backward()
    head = head.to(dev)
    md, sd_ = mean.to(dev).requires_grad_(), std.to(dev).requires_grad_()
    yd = ag.DistanceHead.apply(md, sd_, head.l1.weight, head.l1.bias, head.l2.weight, head.l2.bias, head.l3.weight, head.l3.bias,
                               head.out.weight, head.out.bias, [k.to(dev) for k in keep], 2.0)
    assert rel_err(yd.detach().cpu(), yr.detach()) < 1e-5
    (yd * uy.to(dev)).sum().backward()
    assert rel_err(md.grad.cpu(), mr.grad) < 5e-5 and rel_err(sd_.grad.cpu(), sr.grad) < 5e-5
    for k, v in head.named_parameters():
        assert rel_err(v.grad.cpu(), ps[k].grad) < 5e-5, k
    # the module: eval mode is deterministic, training mode draws masks (seeded: reproducible; differs from eval)
    m4, s4 = mean.to(dev).view(b, 512, 1, 1), std.to(dev).view(b, 512, 1, 1)
    with torch.no_grad():
        e0 = head.eval()((m4, s4))
        head.train()
        torch.manual_seed(1)
        t0 = head((m4, s4))
        torch.manual_seed(1)
        t1 = head((m4, s4))
        head.eval()
    assert torch.equal(t0, t1) and not torch.equal(t0, e0) and t0.shape == e0.shape == (b, 1)


def test_net_forward_gradients_against_reference_golden(golden, dev):
    """The whole training objective on the HIP kernels: ``Net.forward`` (all seven results) -> the scalar objective of the golden
    recipe -> ``backward()``: gradients of every decoder and distance-head parameter against the REFERENCE's own
    (tests/golden/golden_backward.npz, torch.autograd over its aten graph), and a plain SGD step moves the objective down."""
    from hip_helpers import build_network
    from test_oracle_golden import check_gradients, objective_of
    g, gf = golden("golden_backward.npz"), golden("golden_forward.npz")
    network, _ = build_network(0)
    for p in network.decoder.parameters():
        p.requires_grad_(True)
    for p in network.distance_g.parameters():
        p.requires_grad_(True)
    content, style = gf["content"].to(dev), gf["style_img"].to(dev)
    out = network(content, style, 0.6, True, True)
    obj = objective_of(out)
    assert abs(float(obj) - float(g["objective"])) < 1e-4 * abs(float(g["objective"]))
    obj.backward()
    grads = {f"decoder.{k}": p.grad for k, p in network.decoder.named_parameters()}
    grads.update({f"distance_g.{k}": p.grad for k, p in network.distance_g.named_parameters()})
    assert all(v is not None for v in grads.values())
    assert all(p.grad is None for p in network.enc_1.parameters())          # frozen (net.py:167-170)
    worst = check_gradients(g, grads, 5e-4)
    print(f"worst relative gradient deviation from the reference: {worst:.2e}")
    # one SGD step on the decoder lowers the objective (the gradient points the right way end to end)
    with torch.no_grad():
        for p in list(network.decoder.parameters()) + list(network.distance_g.parameters()):
            p -= 1e-4 * p.grad
    network.decoder.invalidate_packed_weights()
    with torch.no_grad():
        obj2 = objective_of(network(content, style, 0.6, True, True))
    assert float(obj2) < float(obj)


def test_gradients_at_benchmark_size_against_aten_on_the_device(dev):
    """4 x 256^2 — the maps of the benchmarked size: weight gradients on the LDS-staged kernel at widths 256 / 128 / 64 / 32,
    pixel-split statistics, Winograd data gradients — against torch.autograd over the oracle's restatement run with aten on the
    same device (too large for the CPU suite; the small-size tests above pin the oracle's gradients to the reference's own)."""
    import os
    import sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import holo_oracle as orc
    from hip_helpers import build_network
    from style_transfer_based_holographic_imaging_amd import synth
    network, _ = build_network(0)
    params = dict(("decoder." + k, p) for k, p in network.decoder.named_parameters())
    params.update(("distance_g." + k, p) for k, p in network.distance_g.named_parameters())
    for p in params.values():
        p.requires_grad_(True)
    ph = synth.synthetic_phase_objects(8, 256, seed=11)
    holo = (0.3 + 0.6 * ph).to(dev)
    content, style = holo[:4].contiguous(), holo[4:].contiguous()

    def objective(out):
        lc, ls, g_t, g_ph, s_re, d_c, d_s = out
        return lc + 10.0 * ls + (s_re - style).pow(2).mean() + g_ph.pow(2).mean() + (d_c - 0.5).pow(2).mean() + (d_s - 0.2).pow(2).mean()

    obj = objective(network(content, style, 0.8, True, True))
    obj.backward()
    vgg_sd, dec_sd, dist_sd = synth.synthetic_state_dicts(0)
    vgg = {k: v.to(dev) for k, v in vgg_sd.items()}
    dec = {k: v.to(dev).requires_grad_() for k, v in dec_sd.items()}
    dist = {k: v.to(dev).requires_grad_() for k, v in dist_sd.items()}
    ref = objective(orc.net_forward(content, style, vgg, dec, dist, 0.8, True, True))
    ref.backward()
    assert abs(float(obj.detach()) - float(ref.detach())) < 1e-4 * abs(float(ref.detach()))
    want = {"decoder." + k: v.grad for k, v in dec.items()}
    want.update({"distance_g." + k: v.grad for k, v in dist.items()})
    assert set(want) == set(params)
    worst = max(rel_err(params[k].grad.cpu(), want[k].cpu()) for k in want)
    print(f"worst relative gradient deviation from aten at 4 x 256^2: {worst:.2e}")
    assert worst < 3e-4


@pytest.mark.parametrize("b,h,w", [(3, 40, 56), (1, 24, 72)])
def test_gradients_at_small_rectangular_sizes(dev, b, h, w):
    """rectangular holograms whose maps are no whole 32-pixel strips or tiles (56 -> 28 -> 14 -> 7 wide: the operands-from-global
    weight-gradient kernel, partial Winograd tiles), a batch in the tall layout and a single hologram: the whole objective against
    torch.autograd over the oracle on the CPU (fp32 against fp32, 2e-4).  Sides are multiples of 8, as the objective itself
    requires (net.py:192: the style loss asserts equal map sizes)."""
    import os
    import sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import holo_oracle as orc
    from hip_helpers import build_network
    from style_transfer_based_holographic_imaging_amd import synth
    network, _ = build_network(0)
    params = dict(("decoder." + k, p) for k, p in network.decoder.named_parameters())
    params.update(("distance_g." + k, p) for k, p in network.distance_g.named_parameters())
    for p in params.values():
        p.requires_grad_(True)
    g = torch.Generator().manual_seed(h * w)
    content = torch.rand(b, 1, h, w, generator=g) * 0.8 + 0.1
    style = torch.rand(b, 1, h, w, generator=g) * 0.8 + 0.1

    def objective(out):
        lc, ls, g_t, g_ph, s_re, d_c, d_s = out
        return lc + 10.0 * ls + s_re.pow(2).mean() + g_ph.pow(2).mean() + (g_t - 0.3).pow(2).mean() + (d_c - 0.5).pow(2).sum() + (d_s - 0.2).pow(2).sum()

    obj = objective(network(content.to(dev), style.to(dev), 0.7, True, True))
    obj.backward()
    vgg_sd, dec_sd, dist_sd = synth.synthetic_state_dicts(0)
    dec = {k: v.clone().requires_grad_() for k, v in dec_sd.items()}
    dist = {k: v.clone().requires_grad_() for k, v in dist_sd.items()}
    ref = objective(orc.net_forward(content, style, vgg_sd, dec, dist, 0.7, True, True))
    ref.backward()
    assert abs(float(obj.detach()) - float(ref.detach())) < 1e-4 * abs(float(ref.detach()))
    want = {"decoder." + k: v.grad for k, v in dec.items()}
    want.update({"distance_g." + k: v.grad for k, v in dist.items()})
    worst = max(rel_err(params[k].grad.cpu(), want[k]) for k in want)
    assert worst < 2e-4, worst
