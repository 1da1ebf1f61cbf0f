"""CPU oracle for the holographic field-retrieval hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the algorithm of the reference's inference path so that the HIP
kernels can be checked on a machine where ``/root/reference`` does not exist.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
package never does (the product path raises when the HIP library is missing).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against
  * the reference's own known-answer data (``test_data/``: hologram == |ASM(gt field)|^2), and
  * golden vectors produced by importing the reference modules in the build container with
    seeded synthetic weights (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).

It is all floating point, so the restatement uses plain torch-CPU fp32 ops for the convolutions
(the "plain PyTorch fp32 reference" of a floating-point kernel) and numpy float64 for the
angular-spectrum transfer function, exactly where the reference uses them.

Each function cites the reference lines it follows (paths relative to ``/root/reference``).
"""
from __future__ import annotations

from math import pi

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# layer tables (net.py:76-107 encoder up to relu4_1, net.py:33-74 decoder)
# --------------------------------------------------------------------------------------------
# entries: ("conv", idx, relu) 3x3 reflect-padded conv ; ("pool",) ; ("convt", idx) k2 s2 + ReLU
ENCODER_PLAN = [
    ("conv1x1", 0),
    ("conv", 2, True),
    ("conv", 5, True), ("pool",),
    ("conv", 9, True),
    ("conv", 12, True), ("pool",),
    ("conv", 16, True),
    ("conv", 19, True), ("conv", 22, True), ("conv", 25, True), ("pool",),
    ("conv", 29, True),
]
# index of the last plan entry of enc_1..enc_4 (net.py:147-151 slices children at 4/11/18/31)
ENCODER_STAGE_ENDS = [1, 4, 7, 12]
DECODER_PLAN = [
    ("conv", 1, True), ("conv", 4, True), ("convt", 6),
    ("conv", 9, True), ("conv", 12, True), ("conv", 15, True), ("conv", 18, True), ("convt", 20),
    ("conv", 23, True), ("conv", 26, True), ("convt", 28),
    ("conv", 31, True), ("conv", 34, True), ("conv", 37, False),
]


def _conv3x3_reflect(x, w, b):
    # net.py:78-79 etc.: ReflectionPad2d((1,1,1,1)) then Conv2d(k=3)
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def encode_with_intermediate(x, vgg_sd):
    """net.py:173-179 (+ the layer list net.py:76-107). Returns [relu1_1, relu2_1, relu3_1, relu4_1]."""
    feats = []
    for i, step in enumerate(ENCODER_PLAN):
        if step[0] == "conv1x1":
            x = F.conv2d(x, vgg_sd[f"{step[1]}.weight"], vgg_sd[f"{step[1]}.bias"])
        elif step[0] == "conv":
            x = F.relu(_conv3x3_reflect(x, vgg_sd[f"{step[1]}.weight"], vgg_sd[f"{step[1]}.bias"]))
        else:  # net.py:83: MaxPool2d((2,2),(2,2),(0,0),ceil_mode=True)
            x = F.max_pool2d(x, 2, 2, 0, ceil_mode=True)
        if i in ENCODER_STAGE_ENDS:
            feats.append(x)
    return feats


def encode(x, vgg_sd):
    """net.py:181-184."""
    return encode_with_intermediate(x, vgg_sd)[-1]


def calc_mean_std(feat, eps=1e-5):
    """function.py:4-12: per-(N,C) mean and sqrt(unbiased var + eps)."""
    assert feat.dim() == 4
    n, c = feat.shape[:2]
    var = feat.reshape(n, c, -1).var(dim=2) + eps
    std = var.sqrt().view(n, c, 1, 1)
    mean = feat.reshape(n, c, -1).mean(dim=2).view(n, c, 1, 1)
    return mean, std


def adaptive_instance_normalization(content_feat, style_feat):
    """function.py:15-23."""
    assert content_feat.shape[:2] == style_feat.shape[:2]
    size = content_feat.size()
    s_mean, s_std = calc_mean_std(style_feat)
    c_mean, c_std = calc_mean_std(content_feat)
    normalized = (content_feat - c_mean.expand(size)) / c_std.expand(size)
    return normalized * s_std.expand(size) + s_mean.expand(size)


def adain_with_style_vector(content_feat, style_feats, alpha=1.0):
    """net.py:234-246: stored-style AdaIN followed by the alpha blend."""
    size = content_feat.size()
    b = content_feat.shape[0]
    if style_feats.shape[0] > 2:
        half = style_feats.shape[0] // 2
        s_mean, s_std = style_feats[:half], style_feats[half:]
    else:
        s_mean, s_std = style_feats[:1].repeat(b, 1, 1, 1), style_feats[1:].repeat(b, 1, 1, 1)
    c_mean, c_std = calc_mean_std(content_feat)
    t = (content_feat - c_mean.expand(size)) / c_std.expand(size)
    t = t * s_std.expand(size) + s_mean.expand(size)
    return alpha * t + (1 - alpha) * content_feat


def decoder_forward(t, dec_sd, return_stages=False):
    """net.py:33-74."""
    stages = []
    x = t
    for step in DECODER_PLAN:
        w, b = dec_sd[f"{step[1]}.weight"], dec_sd[f"{step[1]}.bias"]
        if step[0] == "conv":
            x = _conv3x3_reflect(x, w, b)
            if step[2]:
                x = F.relu(x)
        else:  # net.py:40: ConvTranspose2d(C, C, 2, stride=2, padding=0) + ReLU
            x = F.relu(F.conv_transpose2d(x, w, b, stride=2))
        stages.append(x)
    return (x, stages) if return_stages else x


def distance_head(mean, std, dist_sd, return_logit=False):
    """net.py:288-308.  InstanceNorm1d on a (B,F) input normalises each row over its F features
    with biased variance, eps 1e-5, no affine (torch reads it as an unbatched (C=B, L=F) input);
    Dropout is the identity in eval mode."""
    assert mean.shape == std.shape
    b, c = mean.shape[:2]
    x = torch.cat([mean.reshape(b, c), std.reshape(b, c)], dim=1)
    for name in ("l1", "l2", "l3"):
        x = F.linear(x, dist_sd[f"{name}.weight"], dist_sd[f"{name}.bias"])
        mu = x.mean(dim=1, keepdim=True)
        var = x.var(dim=1, unbiased=False, keepdim=True)
        x = F.relu((x - mu) / torch.sqrt(var + 1e-5))
    logit = F.linear(x, dist_sd["out.weight"], dist_sd["out.bias"])
    return logit if return_logit else torch.sigmoid(logit)


def net_field_retrieval(content, style_feats, vgg_sd, dec_sd, dist_sd=None, alpha=1.0,
                        unknown_distance=False):
    """net.py:228-260 (Net.field_retrieval)."""
    assert 0 <= alpha <= 1
    content_feat = encode(content, vgg_sd)
    t = adain_with_style_vector(content_feat, style_feats, alpha)
    g_t = decoder_forward(t, dec_sd)
    amp, ph = g_t[:, :1], g_t[:, 1:]
    if unknown_distance:
        # net.py:255-256: B==1 repeats the row and keeps the first result (same value)
        d = distance_head(*calc_mean_std(content_feat), dist_sd)
        return amp, ph, d
    return amp, ph


def _mse(a, b):
    return ((a - b) ** 2).mean()


def net_forward(content, style, vgg_sd, dec_sd, dist_sd=None, alpha=1.0, field_retrieval=False, unkonwn_distance=False):
    """net.py:199-226 (Net.forward): the training objective's FORWARD values — content loss on relu4_1 of the generated
    amplitude against the AdaIN target, style loss = mean/std MSE over the four encoder stages (net.py:186-197), and with
    ``field_retrieval`` the generated amplitude / phase, the style reconstruction and (``unkonwn_distance``, sic) both
    predicted distances."""
    assert 0 <= alpha <= 1
    style_feats = encode_with_intermediate(style, vgg_sd)
    content_feat = encode(content, vgg_sd)
    t = adaptive_instance_normalization(content_feat, style_feats[-1])
    t = alpha * t + (1 - alpha) * content_feat
    style_re = decoder_forward(style_feats[-1], dec_sd)[:, :1]
    g_t = decoder_forward(t, dec_sd)
    g_t_phase, g_t = g_t[:, 1:], g_t[:, :1]
    g_t_feats = encode_with_intermediate(g_t, vgg_sd)
    loss_c = _mse(g_t_feats[-1], t)
    loss_s = 0.0
    for gf, sf in zip(g_t_feats, style_feats):
        (gm, gs), (sm, ss) = calc_mean_std(gf), calc_mean_std(sf)
        loss_s = loss_s + _mse(gm, sm) + _mse(gs, ss)
    if not field_retrieval:
        return loss_c, loss_s
    if unkonwn_distance:
        d_content = distance_head(*calc_mean_std(content_feat), dist_sd)
        d_style = distance_head(*calc_mean_std(style_feats[-1]), dist_sd)
        return loss_c, loss_s, g_t, g_t_phase, style_re, d_content, d_style
    return loss_c, loss_s, g_t, g_t_phase, style_re


def field_retrieval(content, style_vector, vgg_sd, dec_sd, dist_sd=None, alpha=1.0,
                    unkonwn_distance=False):
    """utils/functions.py:22-29 (sic: misspelt keyword)."""
    assert 0.0 <= alpha <= 1.0
    if unkonwn_distance:
        a, p, d = net_field_retrieval(content, style_vector, vgg_sd, dec_sd, dist_sd, alpha, True)
        return a, p, d.view(-1, 1, 1, 1)
    return net_field_retrieval(content, style_vector, vgg_sd, dec_sd, None, alpha, False)


# --------------------------------------------------------------------------------------------
# angular-spectrum propagation
# --------------------------------------------------------------------------------------------
def transfer_grid(sh, sw, lamb, px, zero_padding):
    """utils/Angular_Spectrum_Method.py:11-23: the float64 grid sqrt(max(0,1-l^2(fx^2+fy^2)))/l."""
    if zero_padding:
        fx = np.arange(sh * 2) / 2 - sh // 2
        fy = np.arange(sw * 2) / 2 - sw // 2
    else:
        fx = np.arange(sh) - sh // 2
        fy = np.arange(sw) - sw // 2
    fx = fx / (sh * px)
    fy = fy / (sw * px)
    g = 1 - (lamb ** 2) * ((fx ** 2)[:, None] + (fy ** 2)[None, :])
    return np.sqrt((g > 0) * 1 * g) / lamb


def asm(O, lamb, d, px, zero_padding=False):
    """utils/Angular_Spectrum_Method.py:7-36.  ``O`` complex (B,C,S,S); ``d`` a float32 tensor
    broadcastable to (B,1,1,1) in metres (or a python float).  The product ``1j*2*pi*d`` is
    complex64 for a float32 ``d`` before it meets the float64 grid, which fixes the rounding of
    the phase constant; the result is complex128 like the reference's."""
    b, c, sh, sw = O.shape
    if zero_padding:
        O = F.pad(O, (sh // 2, sh // 2, sw // 2, sw // 2), mode="replicate")
    g = torch.from_numpy(np.repeat(transfer_grid(sh, sw, lamb, px, zero_padding)[None, None], b, axis=0))
    h = torch.exp(1j * 2 * pi * d * g)
    o_fft = torch.fft.fftshift(torch.fft.fft2(O), dim=(-2, -1))
    out = torch.fft.ifft2(torch.fft.ifftshift(h * o_fft, dim=(-2, -1)))
    nh, nw = out.shape[-2:]
    return out[:, :, (nh - sh) // 2:(nh + sh) // 2, (nw - sh) // 2:(nw + sh) // 2]


class HoloParams:
    """The five attributes Holo_Generator reads from ``args`` (utils/Forward_model.py:9-13),
    with the entry script's defaults (test_field_retrieval_mnist.py:56-60)."""

    def __init__(self, wavelength=532e-9, pixel_size=1.5e-6, distance_normalize=1.0,
                 distance_normalize_constant=0.0, phase_normalize=1.0):
        self.wavelength = wavelength
        self.pixel_size = pixel_size
        self.distance_normalize = distance_normalize
        self.distance_normalize_constant = distance_normalize_constant
        self.phase_normalize = phase_normalize


def holo_generator(params, amplitude, phase, d, return_field=False, complex_number=False):
    """utils/Forward_model.py:16-39 without the scikit-image unwrap branch (parity unpinned for
    unwrap: scikit-image is absent and un-vendored)."""
    d = ((d + params.distance_normalize_constant) * params.distance_normalize) * 1e-3
    phase = phase * params.phase_normalize
    o_low = amplitude * torch.exp(1j * phase)
    o_holo = asm(o_low, params.wavelength, d, params.pixel_size, zero_padding=True)
    if return_field:
        return torch.abs(o_holo).float(), torch.angle(o_holo).float()
    if complex_number:
        return o_holo
    return torch.pow(torch.abs(o_holo), 2).float()


def full_step(content, style_vector, distance_style, vgg_sd, dec_sd, dist_sd, params=None):
    """One pass of the entry loop, test_field_retrieval_mnist.py:125-126 (pre-unwrap):
    returns (amp_field, ph_field, distance_pred, amp_foc, ph_foc, complex focus field)."""
    params = params or HoloParams()
    amp, ph, dist = field_retrieval(content, style_vector, vgg_sd, dec_sd, dist_sd, 1.0, True)
    d_back = -distance_style - 2 * params.distance_normalize_constant
    field = holo_generator(params, amp, ph * params.phase_normalize, d_back, complex_number=True)
    return amp, ph, dist, torch.abs(field).float(), torch.angle(field).float(), field
