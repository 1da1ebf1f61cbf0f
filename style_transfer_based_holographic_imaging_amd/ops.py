"""Tensor-level wrappers over the C-ABI (``include/holo_hip.h``).

PyTorch is plumbing here: it owns device memory and the stream; every computation below is a
hand-written HIP kernel reached through ``_lib.lib``.  All functions require ROCm device tensors
and raise otherwise — there is no CPU or aten fallback.

Activations inside the network are NHWC: a ``(B,H,W,C)`` contiguous tensor.  ``as_nchw_view``
exposes one as the reference's logical ``(B,C,H,W)`` shape without a copy (channels-last strides).
"""
from __future__ import annotations

import math
import os
import threading

import torch

from . import _lib
from ._lib import check, lib


# device of the launch being prepared: set by _req from the tensors of the call, read by _stream / _launch.  A launch
# goes to the current stream OF THE TENSORS' DEVICE and is issued with that device current (the C-ABI launches on
# whatever device is current, and hipFuncSetAttribute is per device), whatever torch.cuda.current_device() says.
class _Launch(threading.local):      # per Python thread: two threads driving two GPUs must not see each other's device
    dev = None


_LAUNCH = _Launch()


def _stream():
    return torch.cuda.current_stream(_LAUNCH.dev).cuda_stream


def _req(*tensors):
    dev = None
    for t in tensors:
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError(
                "holo HIP ops need tensors on a ROCm device (the MI355X path has no CPU fallback); "
                f"got {type(t).__name__} on {getattr(t, 'device', None)}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"holo HIP ops need all tensors of a call on one device; got {dev} and {t.device}")
    _LAUNCH.dev = dev


def set_conv_tile(tile: int):
    """Test / A-B hook (include/holo_hip.h: holo_debug_set_conv_tile): force a tile variant of the 3x3 convolution;
    0 restores the automatic choice."""
    check(lib.holo_debug_set_conv_tile(int(tile)), "holo_debug_set_conv_tile")


def _f32c(t):
    """float32 + contiguous (no-op when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# "f32s" storage: fp32 values as two fp16 halves (hi + lo' * 2^-11), planar per 16 channels, 4 bytes per element
# (csrc/common.h split16).  Carried in torch.int32 tensors: the dtype is only a tag for "4 opaque bytes per element".
F32S = torch.int32
_ACT_DTYPES = (torch.float32, torch.bfloat16, F32S)


def compute_dtype(name_or_dtype):
    """torch.float32 | torch.bfloat16 | "f32s" (or ops.F32S) -> the activation storage dtype."""
    if name_or_dtype in ("f32s", F32S):
        return F32S
    if name_or_dtype in ("f32", torch.float32):
        return torch.float32
    if name_or_dtype in ("bf16", torch.bfloat16):
        return torch.bfloat16
    raise ValueError(f"compute dtype must be float32, bfloat16 or 'f32s', got {name_or_dtype}")


def _sfx(t):
    """C-ABI suffix of an activation tensor's dtype: the bf16 / f32s entry points mirror the f32 ones."""
    if t.dtype == torch.bfloat16:
        return "bf16"
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == F32S:
        return "f32s"
    raise RuntimeError(f"activations must be float32, bfloat16 or f32s (int32 storage), got {t.dtype}")


def _actc(t):
    """activation tensor: keep bf16 / f32s, anything else becomes fp32; contiguous"""
    if t.dtype not in _ACT_DTYPES:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def cast(x: torch.Tensor, dtype) -> torch.Tensor:
    """fp32 <-> bf16 (round to nearest even) and fp32 <-> f32s conversion of contiguous activation tensors by our kernels."""
    _req(x)
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    if F32S in (x.dtype, dtype) and torch.float32 not in (x.dtype, dtype):      # bf16 <-> f32s goes through fp32
        return cast(cast(x, torch.float32), dtype)
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    name = {(torch.float32, torch.bfloat16): "holo_cast_f32_to_bf16", (torch.bfloat16, torch.float32): "holo_cast_bf16_to_f32",
            (torch.float32, F32S): "holo_cast_f32_to_f32s", (F32S, torch.float32): "holo_cast_f32s_to_f32"}[(x.dtype, dtype)]
    _launch(name, lambda: getattr(lib, name)(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), nbytes=6.0 * x.numel())
    return y


class KernelTimer:
    """Per-launch timing with HIP events on the launch stream (torch's current stream is the stream handed to
    the C-ABI).  ONE event is recorded after every launch (plus one when the region opens); a launch's
    duration is the difference of consecutive events, i.e. it includes the idle gap before the kernel —
    half the events of start/stop pairs (pairs cost ~1 % of a 37 ms step).  Launches made by torch itself
    between two of ours (a few tiny element-wise ops) are charged to our next launch.  ``summary()``
    synchronises and returns {name: dict(calls, ms, flops, bytes)} with ALGORITHMIC flops/bytes."""

    def __init__(self):
        self.records = []
        self._last = None

    def __enter__(self):
        global _TIMER
        self._prev, _TIMER = _TIMER, self
        self._last = torch.cuda.Event(enable_timing=True)
        self._last.record()
        return self

    def __exit__(self, *exc):
        global _TIMER
        _TIMER = self._prev
        return False

    def mark(self):
        """restart the clock (e.g. after work that should not be charged to the next launch)"""
        self._last = torch.cuda.Event(enable_timing=True)
        self._last.record()

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, nbytes, e0, e1 in self.records:
            r = out.setdefault(name, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            r["calls"] += 1
            r["ms"] += e0.elapsed_time(e1)
            r["flops"] += flops
            r["bytes"] += nbytes
        return out


_TIMER = None


def _launch(name, fn, flops=0.0, nbytes=0.0, fresh=False):
    """fresh=True: start this launch's clock right before it (used where torch's own small element-wise launches
    precede ours and must not be charged to the kernel's roofline figure)."""
    dev = _LAUNCH.dev
    if dev is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):      # e.g. --device cuda:1 without torch.cuda.set_device
            return _launch(name, fn, flops, nbytes, fresh)
    if _TIMER is None:
        check(fn(), name)
        return
    if fresh:
        _TIMER.mark()
    check(fn(), name)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    _TIMER.records.append((name, float(flops), float(nbytes), _TIMER._last, e1))
    _TIMER._last = e1


# ---------------------------------------------------------------------------------------------
# layout
# ---------------------------------------------------------------------------------------------
def as_nchw_view(x_nhwc: torch.Tensor) -> torch.Tensor:
    return x_nhwc.permute(0, 3, 1, 2)


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Logical (B,C,H,W) tensor -> contiguous (B,H,W,C); free when already channels-last or C == 1."""
    _req(x)
    assert x.dim() == 4
    b, c, h, w = x.shape
    if x.dtype not in _ACT_DTYPES:
        x = x.float()
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    if x.dtype == F32S:
        raise RuntimeError("f32s activations only exist channels-last (their 16-channel groups are contiguous in NHWC)")
    if c == 1:
        return x.contiguous().view(b, h, w, 1)
    if x.dtype == torch.bfloat16:
        return cast(to_nhwc(cast(x, torch.float32)), torch.bfloat16)
    x = x.contiguous()
    y = torch.empty((b, h, w, c), dtype=torch.float32, device=x.device)
    _launch("holo_nchw_to_nhwc_f32", lambda: lib.holo_nchw_to_nhwc_f32(x.data_ptr(), y.data_ptr(), b, h * w, c, _stream()), nbytes=8.0 * b * h * w * c)
    return y


def to_nchw_contiguous(x_nhwc: torch.Tensor) -> torch.Tensor:
    _req(x_nhwc)
    b, h, w, c = x_nhwc.shape
    x_nhwc = cast(x_nhwc, torch.float32) if x_nhwc.dtype == F32S else _f32c(x_nhwc)
    y = torch.empty((b, c, h, w), dtype=torch.float32, device=x_nhwc.device)
    _launch("holo_nhwc_to_nchw_f32", lambda: lib.holo_nhwc_to_nchw_f32(x_nhwc.data_ptr(), y.data_ptr(), b, h * w, c, _stream()), nbytes=8.0 * b * h * w * c)
    return y


# ---------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------
def pack_conv3x3(w_oihw: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    _req(w_oihw)
    cout, cin, kh, kw = w_oihw.shape
    assert (kh, kw) == (3, 3)
    w = _f32c(w_oihw.detach())
    out = torch.empty(lib.holo_conv3x3_packed_elems(cout, cin), dtype=dtype, device=w.device)
    name = "holo_pack_conv3x3_weight_" + _sfx(out)
    _launch(name, lambda: getattr(lib, name)(w.data_ptr(), out.data_ptr(), cout, cin, _stream()))
    return out


def pack_conv3x3_wino(w_oihw: torch.Tensor) -> torch.Tensor:
    """(Cout,Cin,3,3) -> the transformed weights G g G^T of the fp32 Winograd F(2x2,3x3) kernel."""
    _req(w_oihw)
    cout, cin, kh, kw = w_oihw.shape
    assert (kh, kw) == (3, 3)
    w = _f32c(w_oihw.detach())
    out = torch.empty(lib.holo_conv3x3_wino_packed_elems(cout, cin), dtype=torch.float32, device=w.device)
    _launch("holo_pack_conv3x3_wino_weight_f32", lambda: lib.holo_pack_conv3x3_wino_weight_f32(w.data_ptr(), out.data_ptr(), cout, cin, _stream()))
    return out


def pack_convt2x2(w_iohw: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    _req(w_iohw)
    cin, cout, kh, kw = w_iohw.shape
    assert (kh, kw) == (2, 2)
    w = _f32c(w_iohw.detach())
    out = torch.empty(lib.holo_convt2x2_packed_elems(cin, cout), dtype=dtype, device=w.device)
    name = "holo_pack_convt2x2_weight_" + _sfx(out)
    _launch(name, lambda: getattr(lib, name)(w.data_ptr(), out.data_ptr(), cin, cout, _stream()))
    return out


# ---------------------------------------------------------------------------------------------
# convolutions (activation dtype float32 or bfloat16 selects the _f32 / _bf16 entry point)
# ---------------------------------------------------------------------------------------------
def conv_first(x: torch.Tensor, w0, b0, w1, b1, out_dtype=torch.float32, sqrt_input: bool = False) -> torch.Tensor:
    """vgg[0:4] (net.py:77-80): (B,1,H,W) -> NHWC (B,H,W,64).  sqrt_input: x is the hologram intensity and the loader takes its
    square root (test_field_retrieval_mnist.py:119 folded in; bit-identical to torch.sqrt first)."""
    _req(x, w0, b0, w1, b1)
    b, c, h, w = x.shape
    assert c == 1 and tuple(w0.shape) == (3, 1, 1, 1) and tuple(w1.shape) == (64, 3, 3, 3)
    x = _f32c(x)
    y = torch.empty((b, h, w, 64), dtype=out_dtype, device=x.device)
    name = ("holo_conv_first_sqrt_" if sqrt_input else "holo_conv_first_") + _sfx(y)
    w0c, b0c, w1c, b1c = (_f32c(t.detach()) for t in (w0, b0, w1, b1))
    _launch(name, lambda: getattr(lib, name)(x.data_ptr(), w0c.data_ptr(), b0c.data_ptr(), w1c.data_ptr(), b1c.data_ptr(),
                                             y.data_ptr(), b, h, w, _stream()),
            flops=2.0 * (27 * 64 + 3) * b * h * w, nbytes=b * h * w * (4.0 + 64 * y.element_size()))
    return y


def conv3x3(x_nhwc: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, cout: int, relu=True, pool=False):
    """ReflectionPad2d(1)+Conv2d(3x3)[+ReLU][+MaxPool2d(2,2,ceil_mode=True)] on NHWC."""
    _req(x_nhwc, w_packed, bias)
    b, h, w, cin = x_nhwc.shape
    x_nhwc = _actc(x_nhwc)
    if w_packed.dtype != x_nhwc.dtype:
        raise RuntimeError(f"packed weights are {w_packed.dtype} but activations {x_nhwc.dtype}")
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    y = torch.empty((b, ho, wo, cout), dtype=x_nhwc.dtype, device=x_nhwc.device)
    name = "holo_conv3x3_" + _sfx(x_nhwc)
    es = x_nhwc.element_size()
    _launch(name, lambda: getattr(lib, name)(x_nhwc.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(), b, h, w,
                                             cin, cout, int(relu), int(pool), _stream()),
            flops=2.0 * 9 * cin * cout * b * h * w, nbytes=float(es) * b * (h * w * cin + ho * wo * cout))
    return y


def conv3x3_wino(x_nhwc: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, cout: int, relu=True, pool=False, ksplit: int = 1):
    """conv3x3 as fp32 Winograd F(2x2,3x3) (weights from pack_conv3x3_wino).  FLOP are counted as the direct
    convolution's (the algorithmic work), so a rate above the fp32 MFMA peak is the algorithm's gain.
    ``ksplit`` > 1: split-K for launches that would leave most of the chip idle (holo_conv3x3_wino_splitk_f32)."""
    _req(x_nhwc, w_packed, bias)
    b, h, w, cin = x_nhwc.shape
    if x_nhwc.dtype != torch.float32 or w_packed.dtype != torch.float32:
        raise RuntimeError("the Winograd kernel is fp32 only")
    x_nhwc = _actc(x_nhwc)
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    y = torch.empty((b, ho, wo, cout), dtype=torch.float32, device=x_nhwc.device)
    if ksplit > 1:
        ws_bytes = lib.holo_conv3x3_wino_splitk_workspace_bytes(b, h, w, cout, int(ksplit))
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x_nhwc.device)
        _launch("holo_conv3x3_wino_splitk_f32", lambda: lib.holo_conv3x3_wino_splitk_f32(
            x_nhwc.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(), b, h, w, cin, cout, int(relu), int(pool), int(ksplit),
            ws.data_ptr(), ws_bytes, _stream()),
                flops=2.0 * 9 * cin * cout * b * h * w, nbytes=4.0 * b * (h * w * cin + ho * wo * cout))
        return y
    _launch("holo_conv3x3_wino_f32", lambda: lib.holo_conv3x3_wino_f32(x_nhwc.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                                     b, h, w, cin, cout, int(relu), int(pool), _stream()),
            flops=2.0 * 9 * cin * cout * b * h * w, nbytes=4.0 * b * (h * w * cin + ho * wo * cout))
    return y


def convt2x2(x_nhwc: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, cout: int, relu=True):
    """ConvTranspose2d(k=2,s=2)[+ReLU] on NHWC: (B,H,W,Cin) -> (B,2H,2W,Cout)."""
    _req(x_nhwc, w_packed, bias)
    b, h, w, cin = x_nhwc.shape
    x_nhwc = _actc(x_nhwc)
    if w_packed.dtype != x_nhwc.dtype:
        raise RuntimeError(f"packed weights are {w_packed.dtype} but activations {x_nhwc.dtype}")
    y = torch.empty((b, 2 * h, 2 * w, cout), dtype=x_nhwc.dtype, device=x_nhwc.device)
    name = "holo_convt2x2_" + _sfx(x_nhwc)
    es = x_nhwc.element_size()
    _launch(name, lambda: getattr(lib, name)(x_nhwc.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(), b, h, w,
                                             cin, cout, int(relu), _stream()),
            flops=2.0 * cin * 4 * cout * b * h * w, nbytes=float(es) * b * h * w * (cin + 4 * cout))
    return y


def conv_last(x_nhwc: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """decoder[36:38] (net.py:72-73): NHWC (B,H,W,64) -> NCHW fp32 (B,2,H,W)."""
    _req(x_nhwc, w_oihw, bias)
    b, h, w, cin = x_nhwc.shape
    assert cin == 64 and tuple(w_oihw.shape) == (2, 64, 3, 3)
    x_nhwc = _actc(x_nhwc)
    y = torch.empty((b, 2, h, w), dtype=torch.float32, device=x_nhwc.device)
    name = "holo_conv_last_" + _sfx(x_nhwc)
    wc, bc = _f32c(w_oihw.detach()), _f32c(bias.detach())
    _launch(name, lambda: getattr(lib, name)(x_nhwc.data_ptr(), wc.data_ptr(), bc.data_ptr(), y.data_ptr(), b, h, w, _stream()),
            flops=2.0 * 9 * 64 * 2 * b * h * w, nbytes=b * h * w * (64.0 * x_nhwc.element_size() + 8))
    return y


# ---------------------------------------------------------------------------------------------
# statistics / AdaIN
# ---------------------------------------------------------------------------------------------
def channel_mean_std(x_nhwc: torch.Tensor, eps: float = 1e-5):
    """function.py:4-12 on NHWC: returns fp32 (mean, std) of shape (B,C)."""
    _req(x_nhwc)
    b, h, w, c = x_nhwc.shape
    x_nhwc = _actc(x_nhwc)
    mean = torch.empty((b, c), dtype=torch.float32, device=x_nhwc.device)
    std = torch.empty_like(mean)
    name = "holo_channel_mean_std_" + _sfx(x_nhwc)
    _launch(name, lambda: getattr(lib, name)(x_nhwc.data_ptr(), mean.data_ptr(), std.data_ptr(), b, h * w, c, float(eps), _stream()),
            nbytes=float(x_nhwc.element_size()) * b * h * w * c)
    return mean, std


def adain_apply(x_nhwc, mean, std, style_mean, style_std, alpha: float = 1.0):
    """net.py:241-246.  style_mean/style_std: (1,C) (broadcast) or (B,C)."""
    _req(x_nhwc, mean, std, style_mean, style_std)
    b, h, w, c = x_nhwc.shape
    x_nhwc = _actc(x_nhwc)
    style_mean = _f32c(style_mean.reshape(-1, c))
    style_std = _f32c(style_std.reshape(-1, c))
    assert style_mean.shape == style_std.shape and style_mean.shape[0] in (1, b)
    stride = 0 if style_mean.shape[0] == 1 and b != 1 else c
    y = torch.empty_like(x_nhwc)
    name = "holo_adain_apply_" + _sfx(x_nhwc)
    _launch(name, lambda: getattr(lib, name)(x_nhwc.data_ptr(), mean.data_ptr(), std.data_ptr(), style_mean.data_ptr(),
                                             style_std.data_ptr(), stride, float(alpha), y.data_ptr(), b, h * w, c, _stream()),
            nbytes=2.0 * x_nhwc.element_size() * b * h * w * c)
    return y


def adain_fused(x_nhwc, style_mean, style_std, alpha: float = 1.0, eps: float = 1e-5):
    """channel_mean_std + adain_apply in one launch: returns (y, mean, std), bit-identical to the two calls."""
    _req(x_nhwc, style_mean, style_std)
    b, h, w, c = x_nhwc.shape
    x_nhwc = _actc(x_nhwc)
    style_mean = _f32c(style_mean.reshape(-1, c))
    style_std = _f32c(style_std.reshape(-1, c))
    assert style_mean.shape == style_std.shape and style_mean.shape[0] in (1, b)
    stride = 0 if style_mean.shape[0] == 1 and b != 1 else c
    y = torch.empty_like(x_nhwc)
    mean = torch.empty((b, c), dtype=torch.float32, device=x_nhwc.device)
    std = torch.empty_like(mean)
    name = "holo_adain_fused_" + _sfx(x_nhwc)
    _launch(name, lambda: getattr(lib, name)(x_nhwc.data_ptr(), mean.data_ptr(), std.data_ptr(), style_mean.data_ptr(), style_std.data_ptr(),
                                             stride, float(alpha), y.data_ptr(), b, h * w, c, float(eps), _stream()),
            nbytes=3.0 * x_nhwc.element_size() * b * h * w * c)
    return y, mean, std


# ---------------------------------------------------------------------------------------------
# distance head
# ---------------------------------------------------------------------------------------------
def linear(x, w, bias):
    _req(x, w, bias)
    x = _f32c(x)
    bsz, k = x.shape
    n = w.shape[0]
    y = torch.empty((bsz, n), dtype=torch.float32, device=x.device)
    _launch("holo_linear_f32", lambda: lib.holo_linear_f32(x.data_ptr(), _f32c(w.detach()).data_ptr(), _f32c(bias.detach()).data_ptr(), y.data_ptr(),
                              bsz, k, n, _stream()), flops=2.0 * bsz * k * n, nbytes=4.0 * (n * k + bsz * (k + n)))
    return y


def rownorm_relu(x, eps: float = 1e-5):
    _req(x)
    x = _f32c(x)
    y = torch.empty_like(x)
    _launch("holo_rownorm_relu_f32", lambda: lib.holo_rownorm_relu_f32(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], float(eps), _stream()), nbytes=8.0 * x.numel())
    return y


def linear1_sigmoid(x, w, bias):
    _req(x, w, bias)
    x = _f32c(x)
    y = torch.empty((x.shape[0], 1), dtype=torch.float32, device=x.device)
    _launch("holo_linear1_sigmoid_f32", lambda: lib.holo_linear1_sigmoid_f32(x.data_ptr(), _f32c(w.detach()).data_ptr(), _f32c(bias.detach()).data_ptr(),
                                       y.data_ptr(), x.shape[0], x.shape[1], _stream()), nbytes=4.0 * x.numel())
    return y


def distance_head(mean, std, sd, eps: float = 1e-5):
    """net.py:288-308 on (B,C) statistics; ``sd`` maps l1/l2/l3/out .weight/.bias to device tensors.  The reference's widths
    (C = 512) run as four launches (holo_distance_head_f32); any other width layer by layer."""
    _req(mean, std)
    bsz, c = mean.shape
    shapes = [tuple(sd[f"{n}.weight"].shape) for n in ("l1", "l2", "l3", "out")]
    if c % 256 == 0 and 2 * c <= 1024 and shapes == [(2 * c, 2 * c), (2 * c, 2 * c), (c, 2 * c), (1, c)]:
        mean, std = _f32c(mean), _f32c(std)
        prm = [_f32c(sd[f"{n}.{p}"].detach()) for n in ("l1", "l2", "l3", "out") for p in ("weight", "bias")]
        _req(*prm)
        ws_bytes = lib.holo_distance_head_workspace_bytes(bsz, c)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=mean.device)
        y = torch.empty((bsz, 1), dtype=torch.float32, device=mean.device)
        _launch("holo_distance_head_f32", lambda: lib.holo_distance_head_f32(
            mean.data_ptr(), std.data_ptr(), *[t.data_ptr() for t in prm], y.data_ptr(), bsz, c, float(eps), ws.data_ptr(), ws_bytes, _stream()),
                flops=2.0 * bsz * (2 * (2 * c) ** 2 + 2 * c * c + c), nbytes=4.0 * (2 * (2 * c) ** 2 + 2 * c * c + c))
        return y
    x = torch.cat([mean, std], dim=1)
    for name in ("l1", "l2", "l3"):
        x = rownorm_relu(linear(x, sd[f"{name}.weight"], sd[f"{name}.bias"]))
    return linear1_sigmoid(x, sd["out.weight"], sd["out.bias"])


# ---------------------------------------------------------------------------------------------
# angular spectrum
# ---------------------------------------------------------------------------------------------
# fields per launch group: keeps the two S x P complex64 intermediates inside the 256 MiB Infinity Cache
_ASM_CACHE_BUDGET = int(os.environ.get("HOLO_ASM_GROUP_MB", "320")) << 20      # (the variable is a tuning knob for scripts/bench_asm.py)


def phase_constant(d, n_fields: int, device) -> torch.Tensor:
    """float64 constant c of theta = c*G per field, with the reference's rounding
    (utils/Angular_Spectrum_Method.py:29): ``1j*2*pi*d`` is complex64 when d is a float32 tensor
    — c = fl32(fl32(2 pi) * d) — and a float64 value when d is a Python float / float64 tensor."""
    if isinstance(d, torch.Tensor):
        if d.dtype == torch.float64:
            c = (2 * math.pi) * d.to(device)
        else:
            c = (d.to(device=device, dtype=torch.float32) * (2 * math.pi)).double()  # fp32 multiply by fl32(2 pi)
        c = c.reshape(-1)
        if c.numel() == 1:
            c = c.expand(n_fields)
        assert c.numel() == n_fields, f"distance tensor has {c.numel()} entries for {n_fields} fields"
        return c.contiguous()
    return torch.full((n_fields,), 2 * math.pi * float(d), dtype=torch.float64, device=device)


def phase_constant_from_distance(d: torch.Tensor, add: float, mul: float, n_out: int, period: int) -> torch.Tensor:
    """The same constants as ``phase_constant(((d + add) * mul) * 1e-3, ...)`` for a float32 ROCm tensor ``d`` of ``period``
    distances in the reference's units (millimetres after normalisation), in ONE launch with the identical fp32 rounding chain
    (holo_asm_phase_constants_f32): out[i] = c(d[i % period])."""
    _req(d)
    d = d.reshape(-1)
    assert d.dtype == torch.float32 and d.numel() in (1, period)
    stride = 0 if d.numel() == 1 else d.stride(0)
    out = torch.empty(n_out, dtype=torch.float64, device=d.device)
    _launch("holo_asm_phase_constants_f32", lambda: lib.holo_asm_phase_constants_f32(
        d.data_ptr(), stride, int(period), float(add), float(mul), out.data_ptr(), int(n_out), _stream()))
    return out


def _field_stride(t: torch.Tensor, s: int):
    """Element stride between consecutive (b,c) fields of a (B,C,S,S) tensor whose planes are
    contiguous, or None when the fields are not evenly spaced."""
    b, c = t.shape[:2]
    if t.stride(-1) != 1 or t.stride(-2) != s:
        return None
    if c == 1:
        return t.stride(0) if b > 1 else s * s
    if b == 1:
        return t.stride(1)
    return t.stride(1) if t.stride(0) == c * t.stride(1) else None


def asm_propagate(in0, in1, phase_c, s, zero_padding, wavelength, pixel_size, phase_scale, out_mode, n_dist=1):
    """Low-level launcher.  in0/in1: (B,C,S,S) float32 amplitude/phase (in1 not None) or in0 complex64.
    phase_c: float64 (n_fields*n_dist,).  Returns (B*n_dist... ) shaped like (B,C*n_dist,S,S) outputs."""
    _req(in0, phase_c)
    b, ch = in0.shape[:2]
    n_fields = b * ch
    dev = in0.device
    complex_in = in1 is None
    if complex_in:
        if in0.dtype != torch.complex64:
            in0 = in0.to(torch.complex64)
        st0 = _field_stride(in0, s)
        if st0 is None:
            in0 = in0.contiguous()
            st0 = s * s
        st1 = 0
        p0, p1 = in0.data_ptr(), 0
        in_mode = _lib.ASM_IN_COMPLEX
    else:
        _req(in1)
        in0 = in0 if in0.dtype == torch.float32 else in0.float()
        in1 = in1 if in1.dtype == torch.float32 else in1.float()
        st0, st1 = _field_stride(in0, s), _field_stride(in1, s)
        if st0 is None:
            in0 = in0.contiguous()
            st0 = s * s
        if st1 is None:
            in1 = in1.contiguous()
            st1 = s * s
        p0, p1 = in0.data_ptr(), in1.data_ptr()
        in_mode = _lib.ASM_IN_AMP_PHASE
    n_out = n_fields * n_dist
    shape = (b, ch * n_dist, s, s)
    if out_mode == _lib.ASM_OUT_COMPLEX:
        out0 = torch.empty(shape, dtype=torch.complex64, device=dev)
        out1 = None
    else:
        out0 = torch.empty(shape, dtype=torch.float32, device=dev)
        out1 = torch.empty(shape, dtype=torch.float32, device=dev) if out_mode == _lib.ASM_OUT_AMP_PHASE else None
    p = 2 * s if zero_padding else s
    per_field = s * p * 8 * (1 + n_dist)
    group = max(1, min(n_fields, _ASM_CACHE_BUDGET // per_field, 65535 // n_dist))
    group = -(-n_fields // -(-n_fields // group))      # near-equal groups: 64 fields at 20 per group run as 4 x 16, not 20 + 20 + 20 + 4
    ws_bytes = lib.holo_asm_workspace_bytes(group, n_dist, s, int(bool(zero_padding)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    esz_in = 8 if complex_in else 4
    esz_out = 8 if out_mode == _lib.ASM_OUT_COMPLEX else 4
    plane = s * s
    for f0 in range(0, n_fields, group):
        nf = min(group, n_fields - f0)
        _launch("holo_asm_propagate", lambda: lib.holo_asm_propagate(
            p0 + f0 * st0 * esz_in, (p1 + f0 * st1 * 4) if p1 else 0, in_mode, st0, st1,
            out0.data_ptr() + f0 * n_dist * plane * esz_out,
            (out1.data_ptr() + f0 * n_dist * plane * 4) if out1 is not None else 0, out_mode, plane, plane,
            phase_c.data_ptr() + f0 * n_dist * 8, nf, n_dist, s, int(bool(zero_padding)),
            float(wavelength), float(pixel_size), float(phase_scale), ws.data_ptr(), ws_bytes, _stream()),
                nbytes=(24.0 + 56.0 * n_dist) * plane * nf, fresh=True)
    return out0, out1


_UNWRAP_WS_BUDGET = 4 << 30


def unwrap2d(phase: torch.Tensor) -> torch.Tensor:
    """Per-image 2-D phase unwrapping of (B,1,H,W) / (B,H,W) / (H,W) float32 phase maps on the device
    (utils/functions.py:44-59).  Returns a new (B,1,H,W) float32 tensor, like the reference's wrapper."""
    _req(phase)
    h, w = phase.shape[-2:]
    x = phase.reshape(-1, h, w)
    x = x if x.dtype == torch.float32 else x.float()
    b = x.shape[0]
    if not (x.stride(2) == 1 and x.stride(1) == w and (b == 1 or x.stride(0) >= h * w)):
        x = x.contiguous()
    st = x.stride(0) if b > 1 else h * w     # batch-strided views such as field[:, 1:] need no copy
    out = torch.empty((b, 1, h, w), dtype=torch.float32, device=x.device)
    group = max(1, min(b, _UNWRAP_WS_BUDGET // (24 * h * w), 65535))
    ws_bytes = lib.holo_unwrap2d_workspace_bytes(group, h, w)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    for i0 in range(0, b, group):
        nb = min(group, b - i0)
        _launch("holo_unwrap2d_f32", lambda: lib.holo_unwrap2d_f32(
            x.data_ptr() + i0 * st * 4, st, out.data_ptr() + i0 * h * w * 4, h * w, nb, h, w,
            ws.data_ptr(), ws_bytes, _stream()), nbytes=8.0 * h * w * nb, fresh=True)
    return out


# ---------------------------------------------------------------------------------------------
# backward pass (include/holo_hip.h ABI 5, csrc/backward.hip; SURVEY.md 8f N4: net.py:199-226)
# ---------------------------------------------------------------------------------------------
def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


def maxpool2x2(x_nhwc: torch.Tensor) -> torch.Tensor:
    """MaxPool2d(2,2,ceil_mode=True) on NHWC fp32, unfused (training keeps the pre-pool activation)."""
    _req(x_nhwc)
    x_nhwc = _f32c(x_nhwc)
    b, h, w, c = x_nhwc.shape
    y = torch.empty((b, (h + 1) // 2, (w + 1) // 2, c), dtype=torch.float32, device=x_nhwc.device)
    _launch("holo_maxpool2x2_f32", lambda: lib.holo_maxpool2x2_f32(x_nhwc.data_ptr(), y.data_ptr(), b, h, w, c, _stream()),
            nbytes=5.0 * b * h * w * c)
    return y


def mosaic_shape(b: int, h: int, w: int, border: int, tall: bool, per_row: int = 1):
    """(pitch_rows, rows, width) of the buffer holo_relu_pool_bwd_pad_f32 writes: separate blocks (tall False), one tall image
    (neighbours share their zero rows), or a mosaic with ``per_row`` images side by side (sharing zero columns too)."""
    gx = per_row if per_row and per_row > 1 else 1
    assert gx == 1 or tall, "images side by side come with the shared-row layout"
    pitch = h + border if tall else h + 2 * border
    rows = ((b + gx - 1) // gx - 1) * pitch + h + 2 * border if tall else h + 2 * border
    width = (gx - 1) * (w + border) + w + 2 * border
    return pitch, rows, width


def relu_pool_bwd_pad(dy, y_act, y_pool, relu: bool, border: int, tall: bool = False, with_bias: bool = False, per_row: int = 1):
    """Gradient of a conv step's pre-activation embedded in a zero border: dy routed through the (optional) max-pool and masked
    by the (optional) ReLU.  y_act: (B,h,w,c) post-ReLU pre-pool output; dy, y_pool: pooled shape when y_pool is given.
    tall: ONE image (1, rows, width, c) in which neighbouring images share their zero rows (and, with per_row > 1 images side by
    side, their zero columns: mosaic_shape) instead of (B, h+2*border, w+2*border, c).
    with_bias: also returns the bias gradient (c), summed in the same pass."""
    _req(dy, y_act)
    dy, y_act = _f32c(dy), _f32c(y_act)
    b, h, w, c = y_act.shape
    yp = _f32c(y_pool) if y_pool is not None else None
    pitch, rows, width = mosaic_shape(b, h, w, border, tall, per_row)
    shape = (1, rows, width, c) if tall else (b, rows, width, c)
    out = torch.empty(shape, dtype=torch.float32, device=dy.device)
    dbias = torch.empty((c,), dtype=torch.float32, device=dy.device) if with_bias else None
    n = lib.holo_relu_pool_bwd_pad_workspace_bytes(c) if with_bias else 0
    ws = _ws(n, dy.device) if with_bias else None
    _launch("holo_relu_pool_bwd_pad_f32",
            lambda: lib.holo_relu_pool_bwd_pad_f32(dy.data_ptr(), y_act.data_ptr(), yp.data_ptr() if yp is not None else 0, out.data_ptr(),
                                                   b, h, w, c, int(bool(relu)), int(border), int(pitch), int(per_row),
                                                   dbias.data_ptr() if with_bias else 0, ws.data_ptr() if with_bias else 0, n, _stream()),
            nbytes=12.0 * b * h * w * c)
    return (out, dbias) if with_bias else out


def fold_reflect(dxp, h: int, w: int, border: int, batch: int = 0, per_row: int = 1) -> torch.Tensor:
    """Adjoint of ReflectionPad2d(1): (B,h+2*border,w+2*border,c) -> (B,h,w,c); with ``batch`` the input is the tall / mosaic
    layout of relu_pool_bwd_pad (one image holding the whole batch)."""
    _req(dxp)
    dxp = _f32c(dxp)
    b, hp, wp, c = dxp.shape
    pitch, rows, width = mosaic_shape(batch or b, h, w, border, bool(batch), per_row)
    assert (b == 1 if batch else True) and hp == rows and wp == width, (dxp.shape, rows, width)
    b = batch or b
    dx = torch.empty((b, h, w, c), dtype=torch.float32, device=dxp.device)
    _launch("holo_fold_reflect_f32", lambda: lib.holo_fold_reflect_f32(dxp.data_ptr(), dx.data_ptr(), b, h, w, c, int(border), int(pitch), int(per_row),
                                                                       _stream()),
            nbytes=8.0 * b * h * w * c)
    return dx


def conv1x1(x_nhwc, w_kn, bias=None, relu: bool = False) -> torch.Tensor:
    """(B,h,w,cin) . w_kn (cin, cout) [+ bias] [ReLU] -> (B,h,w,cout), fp32, on the transposed convolution's one-tap MFMA GEMM."""
    _req(x_nhwc, w_kn)
    x_nhwc, w_kn = _f32c(x_nhwc), _f32c(w_kn.detach())
    b, h, w, cin = x_nhwc.shape
    cout = w_kn.shape[1]
    assert w_kn.shape[0] == cin and cout % 4 == 0
    # the packer's (Cin, Cout', 2, 2) layout with GEMM column (2dy+dx)*Cout' + c  (a re-layout of the weights: no arithmetic)
    w_fake = w_kn.view(cin, 4, cout // 4).permute(0, 2, 1).reshape(cin, cout // 4, 2, 2).contiguous()
    wp = pack_convt2x2(w_fake, torch.float32)
    bias = torch.zeros(cout, dtype=torch.float32, device=x_nhwc.device) if bias is None else _f32c(bias)
    y = torch.empty((b, h, w, cout), dtype=torch.float32, device=x_nhwc.device)
    _launch("holo_conv1x1_f32", lambda: lib.holo_conv1x1_f32(x_nhwc.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr(), b, h, w, cin, cout,
                                                             int(bool(relu)), _stream()),
            flops=2.0 * b * h * w * cin * cout, nbytes=4.0 * (x_nhwc.numel() + y.numel()))
    return y


def space_to_depth2_relu(dy, y, relu: bool) -> torch.Tensor:
    """(B,2h,2w,c) gradient of a transposed-conv step's output -> (B,h,w,4c), masked by its ReLU."""
    _req(dy)
    dy = _f32c(dy)
    b, h2, w2, c = dy.shape
    yy = _f32c(y) if relu else None
    g = torch.empty((b, h2 // 2, w2 // 2, 4 * c), dtype=torch.float32, device=dy.device)
    _launch("holo_space_to_depth2_relu_f32",
            lambda: lib.holo_space_to_depth2_relu_f32(dy.data_ptr(), yy.data_ptr() if yy is not None else 0, g.data_ptr(), b, h2 // 2, w2 // 2, c,
                                                      int(bool(relu)), _stream()), nbytes=12.0 * dy.numel())
    return g


def channel_sum(x_rows_c: torch.Tensor) -> torch.Tensor:
    """sum over all leading dimensions of a (..., c) fp32 tensor (bias gradients); c % 64 == 0"""
    _req(x_rows_c)
    x_rows_c = _f32c(x_rows_c)
    c = x_rows_c.shape[-1]
    rows = x_rows_c.numel() // c
    out = torch.empty((c,), dtype=torch.float32, device=x_rows_c.device)
    n = lib.holo_channel_sum_workspace_bytes(rows, c)
    ws = _ws(n, x_rows_c.device)
    _launch("holo_channel_sum_f32", lambda: lib.holo_channel_sum_f32(x_rows_c.data_ptr(), out.data_ptr(), rows, c, ws.data_ptr(), n, _stream()),
            nbytes=4.0 * x_rows_c.numel())
    return out


def pix_gemm(p_padded, p_border: int, q, taps: int, per_row: int = 1) -> torch.Tensor:
    """out[m][n][tap] = sum over pixels P[pixel][m] * Q[reflect(pixel + tap)][n] (weight gradients; fp32 MFMA, deterministic).
    P: (B, h+2*border, w+2*border, mc), or the tall / mosaic layout of relu_pool_bwd_pad (one image holding the whole batch,
    ``per_row`` images side by side)."""
    _req(p_padded, q)
    p_padded, q = _f32c(p_padded), _f32c(q)
    b, h, w, nc = q.shape
    mc = p_padded.shape[-1]
    tall = not (p_padded.shape[0] == b and p_padded.shape[1] == h + 2 * p_border and per_row <= 1)
    pitch, rows, width = mosaic_shape(b, h, w, p_border, tall, per_row)
    assert p_padded.shape[0] == (1 if tall else b) and p_padded.shape[1] == rows and p_padded.shape[2] == width, (p_padded.shape, rows, width)
    out = torch.empty((mc, nc, taps), dtype=torch.float32, device=q.device)
    n = lib.holo_pix_gemm_workspace_bytes(b, h, w, mc, nc, taps)
    ws = _ws(n, q.device)
    _launch("holo_pix_gemm_f32", lambda: lib.holo_pix_gemm_f32(p_padded.data_ptr(), int(p_border), int(pitch), int(per_row), q.data_ptr(), out.data_ptr(),
                                                               b, h, w, mc, nc, int(taps), ws.data_ptr(), n, _stream()),
            flops=2.0 * taps * mc * nc * b * h * w, nbytes=4.0 * (p_padded.numel() + q.numel()))
    return out


def conv_last_bwd(dy_planes, x_nhwc, w_oihw, need_dx=True, need_dw=True):
    """decoder[36:38] backward: returns (dx (B,h,w,64) or None, dw (2,64,3,3) or None, db (2) or None)."""
    _req(dy_planes, x_nhwc, w_oihw)
    dy_planes, x_nhwc, w_oihw = _f32c(dy_planes), _f32c(x_nhwc), _f32c(w_oihw.detach())
    b, h, w, _ = x_nhwc.shape
    dev = x_nhwc.device
    dxp = torch.empty((b, h + 2, w + 2, 64), dtype=torch.float32, device=dev) if need_dx else None
    dw = torch.empty((2, 64, 3, 3), dtype=torch.float32, device=dev) if need_dw else None
    db = torch.empty((2,), dtype=torch.float32, device=dev) if need_dw else None
    n = lib.holo_conv_last_bwd_workspace_bytes(b, h, w)
    ws = _ws(n, dev)
    _launch("holo_conv_last_bwd_f32",
            lambda: lib.holo_conv_last_bwd_f32(dy_planes.data_ptr(), x_nhwc.data_ptr(), w_oihw.data_ptr(), dxp.data_ptr() if need_dx else 0,
                                               dw.data_ptr() if need_dw else 0, db.data_ptr() if need_dw else 0, b, h, w, ws.data_ptr(), n, _stream()),
            flops=2.0 * 2 * 64 * 9 * b * h * w * (int(need_dx) + int(need_dw)), nbytes=4.0 * x_nhwc.numel())
    dx = fold_reflect(dxp, h, w, 1) if need_dx else None
    return dx, dw, db


def conv_first_dgrad(da_nhwc, w0, w1) -> torch.Tensor:
    """vgg[0:4] backward, data gradient: (B,h,w,64) masked gradient -> (B,1,h,w)."""
    _req(da_nhwc, w0, w1)
    da_nhwc = _f32c(da_nhwc)
    b, h, w, _ = da_nhwc.shape
    din = torch.empty((b, 1, h, w), dtype=torch.float32, device=da_nhwc.device)
    n = b * (h + 2) * (w + 2) * 16
    ws = _ws(n, da_nhwc.device)
    w0c, w1c = _f32c(w0.detach()), _f32c(w1.detach())
    _launch("holo_conv_first_dgrad_f32", lambda: lib.holo_conv_first_dgrad_f32(da_nhwc.data_ptr(), w0c.data_ptr(), w1c.data_ptr(), din.data_ptr(),
                                                                               b, h, w, ws.data_ptr(), n, _stream()),
            flops=2.0 * 27 * 64 * b * h * w, nbytes=4.0 * da_nhwc.numel())
    return din


WIDE_STATS_MIN_HW = 2048


def channel_mean_std_wide(x_nhwc: torch.Tensor, eps: float = 1e-5):
    """function.py:4-12 on a large NHWC fp32 map, the pixels split over blocks (training losses at small batch)."""
    _req(x_nhwc)
    x_nhwc = _f32c(x_nhwc)
    b, h, w, c = x_nhwc.shape
    mean = torch.empty((b, c), dtype=torch.float32, device=x_nhwc.device)
    std = torch.empty_like(mean)
    n = lib.holo_channel_mean_std_wide_workspace_bytes(b, h * w, c)
    ws = _ws(n, x_nhwc.device)
    _launch("holo_channel_mean_std_wide_f32", lambda: lib.holo_channel_mean_std_wide_f32(x_nhwc.data_ptr(), mean.data_ptr(), std.data_ptr(), b, h * w, c,
                                                                                         float(eps), ws.data_ptr(), n, _stream()),
            nbytes=4.0 * x_nhwc.numel())
    return mean, std


def mean_std_bwd(x_nhwc, mean, std, dmean, dstd) -> torch.Tensor:
    """calc_mean_std backward on NHWC fp32: mean/std/dmean/dstd (B,C)."""
    _req(x_nhwc, mean, std, dmean, dstd)
    x_nhwc = _f32c(x_nhwc)
    b, h, w, c = x_nhwc.shape
    m, s, dm, ds = (_f32c(t.reshape(b, c)) for t in (mean, std, dmean, dstd))
    dx = torch.empty_like(x_nhwc)
    _launch("holo_mean_std_bwd_f32", lambda: lib.holo_mean_std_bwd_f32(x_nhwc.data_ptr(), m.data_ptr(), s.data_ptr(), dm.data_ptr(), ds.data_ptr(),
                                                                       dx.data_ptr(), b, h * w, c, _stream()), nbytes=8.0 * x_nhwc.numel())
    return dx


def mask_scale(x, keep, scale: float) -> torch.Tensor:
    """x * keep * scale (training-mode Dropout and its gradient), fp32."""
    _req(x, keep)
    x, keep = _f32c(x), _f32c(keep)
    assert x.shape == keep.shape
    y = torch.empty_like(x)
    _launch("holo_mask_scale_f32", lambda: lib.holo_mask_scale_f32(x.data_ptr(), keep.data_ptr(), float(scale), y.data_ptr(), x.numel(), _stream()),
            nbytes=12.0 * x.numel())
    return y


def linear_bwd(dy, x, w, need_dx=True, need_dw=True):
    _req(dy, x, w)
    dy, x, w = _f32c(dy), _f32c(x), _f32c(w.detach())
    b, fout = dy.shape
    fin = x.shape[1]
    dx = torch.empty((b, fin), dtype=torch.float32, device=dy.device) if need_dx else None
    dw = torch.empty((fout, fin), dtype=torch.float32, device=dy.device) if need_dw else None
    db = torch.empty((fout,), dtype=torch.float32, device=dy.device) if need_dw else None
    _launch("holo_linear_bwd_f32", lambda: lib.holo_linear_bwd_f32(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr() if need_dx else 0,
                                                                   dw.data_ptr() if need_dw else 0, db.data_ptr() if need_dw else 0, b, fin, fout, _stream()))
    return dx, dw, db


def rownorm_relu_bwd(z, dy, eps=1e-5) -> torch.Tensor:
    _req(z, dy)
    z, dy = _f32c(z), _f32c(dy)
    b, f = z.shape
    dz = torch.empty_like(z)
    _launch("holo_rownorm_relu_bwd_f32", lambda: lib.holo_rownorm_relu_bwd_f32(z.data_ptr(), dy.data_ptr(), dz.data_ptr(), b, f, float(eps), _stream()))
    return dz


def sigmoid_bwd(y, dy) -> torch.Tensor:
    _req(y, dy)
    y, dy = _f32c(y), _f32c(dy)
    dz = torch.empty_like(y)
    _launch("holo_sigmoid_bwd_f32", lambda: lib.holo_sigmoid_bwd_f32(y.data_ptr(), dy.data_ptr(), dz.data_ptr(), y.numel(), _stream()))
    return dz
