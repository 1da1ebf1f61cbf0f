"""torch.autograd Functions over the HIP kernels: the backward pass of the training objective.

The reference trains its decoder and distance head by differentiating ``Net.forward`` (net.py:199-226, losses net.py:186-197)
with torch.autograd over aten ops.  Here every step of a layer plan (``engine.Step``) is one ``torch.autograd.Function`` whose
forward AND backward are C-ABI calls (include/holo_hip.h, ABI 5; csrc/backward.hip):

  conv step      ReflPad + Conv3x3 [+ ReLU] [+ MaxPool ceil]
                 dX = fold_reflect( forward_conv( pad2(route/mask(dY)), flip(W)^T ) )      the data gradient runs on the FORWARD kernels
                 dW = pix_gemm(pad2(...), X, 9 taps),  db summed by the mask / pad pass.  A batch's pad2(...) is ONE tall image
                 (neighbours share their zero rows), so the convolution rounds the batch, not each image, up to whole tiles
  convt step     ConvTranspose k2s2 [+ ReLU]:  G = space_to_depth(mask(dY));  dX = 1x1 conv of G (the transposed convolution's own
                 one-tap MFMA GEMM, stored plainly);  dW = pix_gemm(X, G, 1 tap)
  first / last   the skinny layers: VALU kernels (first: data gradient only, the encoder is frozen, net.py:167-170)
  mean_std       calc_mean_std (function.py:4-12)
  distance head  Linear [/ Dropout in training mode] / per-row norm + ReLU / sigmoid, layer by layer (net.py:288-308)

fp32 only.  View ops (permute, reshape, slicing), ``torch.cat`` of the two statistics, the weight flips / transposes feeding the
packers and the scalar MSE reductions (``nn.MSELoss``, as in the reference) stay torch's own: plumbing, no arithmetic of the path.
"""
from __future__ import annotations

import weakref

import torch

from . import ops

WINO_MIN_ITEMS = 128


# packed forms of a parameter, kept until the parameter is written again (tensor._version: every in-place optimizer update bumps
# it).  A step uses each decoder weight in two passes and each frozen encoder weight in three, forward and flipped: packing once
# per version takes 54 of a step's 72 packing launches (and the flip / transpose copies in front of them) away.
_PACKED = {}      # id(parameter) -> (weak reference to it, {kind: (version, packed tensor)}); dropped when the parameter dies


def _packed(weight, kind):
    """kind: 'wino' | 'direct' (forward convolution), 'wino_t' | 'direct_t' (the data gradient's flipped, transposed weights)"""
    key = id(weight)
    slot = _PACKED.get(key)
    if slot is None or slot[0]() is not weight:         # a new object (an id can be reused once its tensor is gone)
        slot = _PACKED[key] = (weakref.ref(weight, lambda _, k=key: _PACKED.pop(k, None)), {})
    hit = slot[1].get(kind)
    if hit is not None and hit[0] == weight._version:
        return hit[1]
    w = weight.detach()
    if kind.endswith("_t"):
        w = w.flip(2, 3).transpose(0, 1).contiguous()                                     # (Cin, Cout, 3, 3): full correlation
    packed = ops.pack_conv3x3_wino(w) if kind.startswith("wino") else ops.pack_conv3x3(w, torch.float32)
    slot[1][kind] = (weight._version, packed)
    return packed


def forget_packed_weights():
    """drop every cached packed weight (``HoloSequential.invalidate_packed_weights`` calls this: a write through ``.data`` changes
    a parameter without bumping its version)"""
    _PACKED.clear()


def _conv_fwd(x_nhwc, weight, bias, relu, transposed=False):
    """forward 3x3 convolution (reflection padding) of an NHWC fp32 tensor with the OIHW parameter ``weight`` (``transposed``: with
    its flipped, transposed form — the data gradient), on the Winograd kernel where the layer fills the chip, else the
    implicit-GEMM kernel"""
    cout, cin = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
    b, h, w, _ = x_nhwc.shape
    items = b * ((h + 31) // 32) * ((w + 15) // 16) * (cout // 32)
    sfx = "_t" if transposed else ""
    if cin >= 32 and cin % 16 == 0 and cout % 32 == 0 and items >= WINO_MIN_ITEMS:
        return ops.conv3x3_wino(x_nhwc, _packed(weight, "wino" + sfx), bias, cout, relu, False)
    return ops.conv3x3(x_nhwc, _packed(weight, "direct" + sfx), bias, cout, relu, False)


_ZEROS = {}


def _zero_bias(n, device):
    key = (n, str(device))
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return z


def _mosaic_per_row(b, h, w, c):
    """images per mosaic row for the embedded gradients of a batch (0: separate blocks — a single image, or a mosaic past the
    convolution kernels' 2^30 elements per image): the count whose tile-rounded area (32 x 16-pixel tiles) is smallest"""
    if b <= 1:
        return 0
    best, best_area = 0, None
    for gx in range(1, min(b, 8) + 1):
        _, rows, width = ops.mosaic_shape(b, h, w, 2, True, gx)
        if rows * width * c >= (1 << 30):
            continue
        area = -(-rows // 32) * 32 * (-(-width // 16) * 16)
        if best_area is None or area < best_area:
            best, best_area = gx, area
    return best


class ConvStep(torch.autograd.Function):
    """ReflectionPad2d(1) -> Conv2d(3x3) [-> ReLU] [-> MaxPool2d(2,2,ceil_mode=True)], NHWC fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, pool):
        y_act = _conv_fwd(x.detach(), weight, bias.detach().float().contiguous(), relu)
        y = ops.maxpool2x2(y_act) if pool else y_act
        ctx.relu, ctx.pool = relu, pool
        ctx.weight, ctx.weight_version = weight, weight._version     # the parameter itself: packed forms are cached per version
        ctx.save_for_backward(x.detach(), y_act if (relu or pool) else None, y if pool else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y_act, y_pool = ctx.saved_tensors
        w = ctx.weight
        if w._version != ctx.weight_version:
            raise RuntimeError("a convolution weight needed for the backward pass was modified in place after the forward pass")
        b, h, wd, cin = x.shape
        cout = w.shape[0]
        if y_act is None:
            y_act = x.new_empty((b, h, wd, cout))       # shape carrier only (neither ReLU nor pool reads it)
        # the embedded gradients of the batch as ONE image — a mosaic in which neighbours share their zero rows and columns — while
        # it stays under the convolution kernels' 2^30 elements per image: the data-gradient convolution then rounds the batch, not
        # every image, up to whole 32 x 16-pixel tiles (16 maps of 32 x 32: 96 tiles apart, 54 in one column, 45 as 8 rows of 2)
        per_row = _mosaic_per_row(b, h, wd, max(cin, cout))
        tall = per_row > 0
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        fused_bias = need_w and 1024 % cout == 0                    # the bias gradient summed by the mask / pad pass itself
        dap = ops.relu_pool_bwd_pad(dy.contiguous(), y_act, y_pool, ctx.relu, 2, tall=tall, with_bias=fused_bias, per_row=max(per_row, 1))
        dx = dw = db = None                                         # dap: (B, h+4, w+4, Cout) | (1, B(h+2)+2, w+4, Cout)
        if fused_bias:
            dap, db = dap
        if ctx.needs_input_grad[0]:
            if cin % 64 != 0:
                raise NotImplementedError(f"data gradient of a {cin} -> {cout} convolution: the forward kernels need Cout % 64 == 0")
            dxp = _conv_fwd(dap, w, _zero_bias(cin, x.device), False, transposed=True)    # same shape as dap, Cin channels
            dx = ops.fold_reflect(dxp, h, wd, 2, batch=b if tall else 0, per_row=max(per_row, 1))
        if need_w:
            dw = ops.pix_gemm(dap, 2, x, 9, per_row=max(per_row, 1)).view(cout, cin, 3, 3)
            if db is None:
                db = ops.channel_sum(dap)
        return dx, dw, db, None, None


class ConvTStep(torch.autograd.Function):
    """ConvTranspose2d(k=2, s=2) [-> ReLU], NHWC fp32; weight (Cin, Cout, 2, 2)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        w = weight.detach()
        cout = w.shape[1]
        y = ops.convt2x2(x.detach(), ops.pack_convt2x2(w, torch.float32), bias.detach().float().contiguous(), cout, relu)
        ctx.relu = relu
        ctx.save_for_backward(x.detach(), w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        b, h, wd, cin = x.shape
        cout = w.shape[1]
        g = ops.space_to_depth2_relu(dy.contiguous(), y, ctx.relu)                        # (B, h, w, 4*Cout), channel q*Cout + co
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if cin % 64 != 0:
                raise NotImplementedError("data gradient of the transposed convolution needs Cin % 64 == 0")
            # a 1x1 convolution of G: weights (4*Cout, Cin), column ci of row q*Cout + co = w[ci][co][q]
            dx = ops.conv1x1(g, w.permute(2, 3, 1, 0).reshape(4 * cout, cin))
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = ops.pix_gemm(x, 0, g, 1).view(cin, 2, 2, cout).permute(0, 3, 1, 2).contiguous()
            db = ops.channel_sum(g).view(4, cout).sum(0)
        return dx, dw, db, None


class FirstStep(torch.autograd.Function):
    """vgg[0:4]: Conv2d(1,3,1x1) -> ReflPad -> Conv2d(3,64,3x3) -> ReLU; frozen (net.py:167-170): data gradient only."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1):
        y = ops.conv_first(x.detach(), w0.detach(), b0.detach(), w1.detach(), b1.detach(), out_dtype=torch.float32)
        ctx.save_for_backward(w0.detach(), w1.detach(), y)
        return y

    @staticmethod
    def backward(ctx, dy):
        w0, w1, y = ctx.saved_tensors
        if any(ctx.needs_input_grad[1:]):
            raise NotImplementedError("the first encoder layer is frozen on this path (net.py:167-170): no weight gradient")
        da = ops.relu_pool_bwd_pad(dy.contiguous(), y, None, True, 0)
        return ops.conv_first_dgrad(da, w0, w1), None, None, None, None


class LastStep(torch.autograd.Function):
    """decoder[36:38]: ReflPad -> Conv2d(64, 2, 3x3), NHWC in, (B,2,H,W) planes out."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x.detach(), weight.detach())
        return ops.conv_last(x.detach(), weight.detach().float().contiguous(), bias.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        need_dw = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dw, db = ops.conv_last_bwd(dy.contiguous(), x, w, need_dx=ctx.needs_input_grad[0], need_dw=need_dw)
        return dx, dw, db


class MeanStd(torch.autograd.Function):
    """calc_mean_std (function.py:4-12) on NHWC fp32 -> ((B,C), (B,C))."""

    @staticmethod
    def forward(ctx, x, eps):
        xd = x.detach()
        wide = xd.dtype == torch.float32 and xd.shape[1] * xd.shape[2] > ops.WIDE_STATS_MIN_HW
        mean, std = ops.channel_mean_std_wide(xd, eps) if wide else ops.channel_mean_std(xd, eps)
        ctx.save_for_backward(xd, mean, std)
        return mean, std

    @staticmethod
    def backward(ctx, dmean, dstd):
        x, mean, std = ctx.saved_tensors
        return ops.mean_std_bwd(x, mean, std, dmean.contiguous(), dstd.contiguous()), None


class DistanceHead(torch.autograd.Function):
    """Distance_G.forward (net.py:288-308) on (mean, std) of shape (B,512): three Linear [-> Dropout] -> per-row norm -> ReLU, then
    sigmoid(Linear).  Gradients for the eight parameters and the two inputs.  ``keep``: None (eval mode: Dropout is the identity)
    or the three drawn 0/1 masks of training mode, applied to each Linear's output with ``scale`` = 1/(1-p) (net.py:298-304)."""

    @staticmethod
    def forward(ctx, mean, std, w1, b1, w2, b2, w3, b3, wo, bo, keep=None, scale=1.0):
        x0 = torch.cat([mean.detach(), std.detach()], 1).float().contiguous()
        ps = [t.detach().float().contiguous() for t in (w1, b1, w2, b2, w3, b3, wo, bo)]
        drop = (lambda z, k: ops.mask_scale(z, keep[k], scale)) if keep is not None else (lambda z, k: z)
        z1 = drop(ops.linear(x0, ps[0], ps[1]), 0)
        a1 = ops.rownorm_relu(z1)
        z2 = drop(ops.linear(a1, ps[2], ps[3]), 1)
        a2 = ops.rownorm_relu(z2)
        z3 = drop(ops.linear(a2, ps[4], ps[5]), 2)
        a3 = ops.rownorm_relu(z3)
        y = ops.linear1_sigmoid(a3, ps[6], ps[7])
        ctx.keep, ctx.scale = keep, scale
        ctx.save_for_backward(x0, z1, a1, z2, a2, z3, a3, y, ps[0], ps[2], ps[4], ps[6])
        return y

    @staticmethod
    def backward(ctx, dy):
        x0, z1, a1, z2, a2, z3, a3, y, w1, w2, w3, wo = ctx.saved_tensors
        keep, scale = ctx.keep, ctx.scale
        undrop = (lambda dz, k: ops.mask_scale(dz, keep[k], scale)) if keep is not None else (lambda dz, k: dz)
        dz4 = ops.sigmoid_bwd(y, dy.contiguous())
        da3, dwo, dbo = ops.linear_bwd(dz4, a3, wo)
        dz3 = undrop(ops.rownorm_relu_bwd(z3, da3), 2)
        da2, dw3, db3 = ops.linear_bwd(dz3, a2, w3)
        dz2 = undrop(ops.rownorm_relu_bwd(z2, da2), 1)
        da1, dw2, db2 = ops.linear_bwd(dz2, a1, w2)
        dz1 = undrop(ops.rownorm_relu_bwd(z1, da1), 0)
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        dx0, dw1, db1 = ops.linear_bwd(dz1, x0, w1, need_dx=need_in)
        c = x0.shape[1] // 2
        dmean = dx0[:, :c].contiguous() if need_in else None
        dstd = dx0[:, c:].contiguous() if need_in else None
        return dmean, dstd, dw1, db1, dw2, db2, dw3, db3, dwo, dbo, None, None


def needs_graph(x, *modules):
    """True when autograd is recording and the input or one of the parameters asks for a gradient."""
    if not torch.is_grad_enabled():
        return False
    if isinstance(x, torch.Tensor) and x.requires_grad:
        return True
    return any(p.requires_grad for m in modules if m is not None for p in m.parameters())


def run_step(step, x):
    """one engine.Step through its autograd Function (fp32)"""
    conv = step.conv
    if step.kind == "first":
        return FirstStep.apply(x, step.conv0.weight, step.conv0.bias, conv.weight, conv.bias)
    if step.kind == "conv":
        return ConvStep.apply(x, conv.weight, conv.bias, bool(step.relu), bool(step.pool))
    if step.kind == "convt":
        return ConvTStep.apply(x, conv.weight, conv.bias, bool(step.relu))
    return LastStep.apply(x, conv.weight, conv.bias)
