"""``torch.library`` registration of the HIP operators (namespace ``holo``), on top of the C-ABI.

BASELINE.json's north_star asks for the hot path "exposed as custom torch ops"; SURVEY.md 8b suggests the names.  Each op is
a ``torch.library.custom_op`` whose ROCm implementation is the ctypes wrapper of ``ops.py`` (one C-ABI call) and whose fake
(meta) implementation gives shapes and dtypes, so the ops are visible to the dispatcher, traceable with fake tensors /
``torch.compile`` and checkable with ``torch.library.opcheck``.  They are registered for ROCm devices ("cuda" in torch) only:
a CPU tensor fails in the dispatcher — there is no CPU kernel on this path.

The network (``engine.Plan``) calls the wrappers directly rather than through the dispatcher: a dispatcher round trip costs
10-20 us per call, 42 calls per step.  ``import style_transfer_based_holographic_imaging_amd.torch_ops`` registers the ops.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from . import _lib, ops

_DEV = "cuda"       # ROCm devices are torch "cuda" devices


def _pooled(h, w, pool):
    return ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)


@torch.library.custom_op("holo::conv_first", mutates_args=(), device_types=_DEV)
def conv_first(x: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor) -> Tensor:
    """vgg[0:4] (net.py:77-80): (B,1,H,W) planes -> NHWC (B,H,W,64)."""
    return ops.conv_first(x, w0, b0, w1, b1)


@conv_first.register_fake
def _(x, w0, b0, w1, b1):
    b, _, h, w = x.shape
    return x.new_empty((b, h, w, 64), dtype=torch.float32)


@torch.library.custom_op("holo::conv3x3_reflect_bias_relu", mutates_args=(), device_types=_DEV)
def conv3x3_reflect_bias_relu(x_nhwc: Tensor, w_packed: Tensor, bias: Tensor, cout: int, relu: bool, pool: bool) -> Tensor:
    """ReflectionPad2d(1) + Conv2d(3x3) [+ ReLU] [+ MaxPool2d(2,2,ceil)] on NHWC, implicit-GEMM MFMA kernel
    (weights from ``ops.pack_conv3x3``)."""
    return ops.conv3x3(x_nhwc, w_packed, bias, cout, relu, pool)


@conv3x3_reflect_bias_relu.register_fake
def _(x_nhwc, w_packed, bias, cout, relu, pool):
    b, h, w, _ = x_nhwc.shape
    ho, wo = _pooled(h, w, pool)
    return x_nhwc.new_empty((b, ho, wo, cout))


@torch.library.custom_op("holo::conv3x3_winograd", mutates_args=(), device_types=_DEV)
def conv3x3_winograd(x_nhwc: Tensor, w_packed: Tensor, bias: Tensor, cout: int, relu: bool, pool: bool, ksplit: int = 1) -> Tensor:
    """The same layer as fp32 Winograd F(2x2,3x3) on the fp32 matrix cores (weights from ``ops.pack_conv3x3_wino``)."""
    return ops.conv3x3_wino(x_nhwc, w_packed, bias, cout, relu, pool, ksplit)


@conv3x3_winograd.register_fake
def _(x_nhwc, w_packed, bias, cout, relu, pool, ksplit=1):
    b, h, w, _ = x_nhwc.shape
    ho, wo = _pooled(h, w, pool)
    return x_nhwc.new_empty((b, ho, wo, cout), dtype=torch.float32)


@torch.library.custom_op("holo::convt2x2s2_bias_relu", mutates_args=(), device_types=_DEV)
def convt2x2s2_bias_relu(x_nhwc: Tensor, w_packed: Tensor, bias: Tensor, cout: int, relu: bool) -> Tensor:
    """ConvTranspose2d(k=2, s=2) [+ ReLU] on NHWC (weights from ``ops.pack_convt2x2``)."""
    return ops.convt2x2(x_nhwc, w_packed, bias, cout, relu)


@convt2x2s2_bias_relu.register_fake
def _(x_nhwc, w_packed, bias, cout, relu):
    b, h, w, _ = x_nhwc.shape
    return x_nhwc.new_empty((b, 2 * h, 2 * w, cout))


@torch.library.custom_op("holo::conv_last", mutates_args=(), device_types=_DEV)
def conv_last(x_nhwc: Tensor, w_oihw: Tensor, bias: Tensor) -> Tensor:
    """decoder[36:38] (net.py:72-73): NHWC (B,H,W,64) -> NCHW fp32 (B,2,H,W) amplitude / phase planes."""
    return ops.conv_last(x_nhwc, w_oihw, bias)


@conv_last.register_fake
def _(x_nhwc, w_oihw, bias):
    b, h, w, _ = x_nhwc.shape
    return x_nhwc.new_empty((b, 2, h, w), dtype=torch.float32)


@torch.library.custom_op("holo::channel_mean_std", mutates_args=(), device_types=_DEV)
def channel_mean_std(x_nhwc: Tensor, eps: float = 1e-5) -> Tuple[Tensor, Tensor]:
    """function.py:4-12 on NHWC: per-(sample, channel) mean and sqrt(unbiased variance + eps), fp32 (B,C)."""
    return ops.channel_mean_std(x_nhwc, eps)


@channel_mean_std.register_fake
def _(x_nhwc, eps=1e-5):
    b, _, _, c = x_nhwc.shape
    return x_nhwc.new_empty((b, c), dtype=torch.float32), x_nhwc.new_empty((b, c), dtype=torch.float32)


@torch.library.custom_op("holo::adain_apply", mutates_args=(), device_types=_DEV)
def adain_apply(x_nhwc: Tensor, mean: Tensor, std: Tensor, style_mean: Tensor, style_std: Tensor, alpha: float = 1.0) -> Tensor:
    """net.py:241-246: (x - mean) / std * style_std + style_mean, blended with x by alpha."""
    return ops.adain_apply(x_nhwc, mean, std, style_mean, style_std, alpha)


@adain_apply.register_fake
def _(x_nhwc, mean, std, style_mean, style_std, alpha=1.0):
    return torch.empty_like(x_nhwc)


@torch.library.custom_op("holo::distance_head", mutates_args=(), device_types=_DEV)
def distance_head(mean: Tensor, std: Tensor, l1_w: Tensor, l1_b: Tensor, l2_w: Tensor, l2_b: Tensor, l3_w: Tensor, l3_b: Tensor,
                  out_w: Tensor, out_b: Tensor) -> Tensor:
    """net.py:288-308 (Distance_G) on (B,C) statistics -> (B,1) in (0,1)."""
    sd = {"l1.weight": l1_w, "l1.bias": l1_b, "l2.weight": l2_w, "l2.bias": l2_b, "l3.weight": l3_w, "l3.bias": l3_b,
          "out.weight": out_w, "out.bias": out_b}
    return ops.distance_head(mean, std, sd)


@distance_head.register_fake
def _(mean, std, l1_w, l1_b, l2_w, l2_b, l3_w, l3_b, out_w, out_b):
    return mean.new_empty((mean.shape[0], 1), dtype=torch.float32)


@torch.library.custom_op("holo::asm_propagate", mutates_args=(), device_types=_DEV)
def asm_propagate(amplitude: Tensor, phase: Tensor, phase_const: Tensor, zero_padding: bool, wavelength: float, pixel_size: float,
                  phase_scale: float, intensity: bool, n_dist: int = 1) -> Tuple[Tensor, Tensor]:
    """Holo_Generator.forward (utils/Forward_model.py:16-39) + ASM (utils/Angular_Spectrum_Method.py:7-36): (B,1,S,S) amplitude
    and phase, float64 phase constants (``ops.phase_constant``) -> (amplitude, phase) of the propagated field, or
    (intensity, empty) with ``intensity``; ``n_dist`` distances per field give (B, n_dist, S, S)."""
    s = amplitude.shape[-1]
    mode = _lib.ASM_OUT_INTENSITY if intensity else _lib.ASM_OUT_AMP_PHASE
    out0, out1 = ops.asm_propagate(amplitude, phase, phase_const, s, zero_padding, wavelength, pixel_size, phase_scale, mode, n_dist)
    return out0, (out1 if out1 is not None else out0.new_empty((0,)))


@asm_propagate.register_fake
def _(amplitude, phase, phase_const, zero_padding, wavelength, pixel_size, phase_scale, intensity, n_dist=1):
    b, c, s, _ = amplitude.shape
    out = amplitude.new_empty((b, c * n_dist, s, s), dtype=torch.float32)
    return out, (amplitude.new_empty((0,), dtype=torch.float32) if intensity else torch.empty_like(out))


@torch.library.custom_op("holo::unwrap2d", mutates_args=(), device_types=_DEV)
def unwrap2d(phase: Tensor) -> Tensor:
    """utils/functions.py:44-59: per-image 2-D phase unwrapping, (B,1,H,W) -> (B,1,H,W)."""
    return ops.unwrap2d(phase)


@unwrap2d.register_fake
def _(phase):
    h, w = phase.shape[-2:]
    return phase.new_empty((phase.numel() // (h * w), 1, h, w), dtype=torch.float32)


OPS = ("conv_first", "conv3x3_reflect_bias_relu", "conv3x3_winograd", "convt2x2s2_bias_relu", "conv_last", "channel_mean_std",
       "adain_apply", "distance_head", "asm_propagate", "unwrap2d")
