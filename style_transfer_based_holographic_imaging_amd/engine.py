"""Layer plans: turn the reference's ``nn.Sequential`` layer lists into fused HIP launches.

``net.vgg`` / ``net.decoder`` keep the reference's child layout (``net.py:33-130``) so that
``load_state_dict``, ``list(vgg.children())[:31]`` and the slicing in ``Net.__init__`` work
unchanged; the children are only parameter holders.  ``compile_plan`` pattern-matches a child list
into steps, each one C-ABI call:

    Conv2d(1,3,1x1) . ReflectionPad2d . Conv2d(3,64,3x3) . ReLU           -> conv_first
    ReflectionPad2d . Conv2d(3x3) [. ReLU] [. MaxPool2d(2,2,ceil)]        -> conv3x3 (+relu, +pool)
    ConvTranspose2d(k2,s2) [. ReLU]                                       -> convt2x2
    ReflectionPad2d . Conv2d(64,2,3x3)                                    -> conv_last (NCHW planes out)

Packed weights are cached per plan and rebuilt when a parameter changes (version counter / data pointer of every
parameter of the step; ``load_state_dict`` and ``_apply`` also drop the cache).  Writes through ``param.data`` bump no
version counter: call ``invalidate_packed_weights()`` after one.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class Step:
    __slots__ = ("kind", "conv", "conv0", "relu", "pool")

    def __init__(self, kind, conv, conv0=None, relu=False, pool=False):
        self.kind, self.conv, self.conv0, self.relu, self.pool = kind, conv, conv0, relu, pool

    def __repr__(self):
        c = self.conv
        return f"Step({self.kind}, {tuple(c.weight.shape)}, relu={self.relu}, pool={self.pool})"


def _is_reflpad1(m):
    return isinstance(m, nn.ReflectionPad2d) and tuple(m.padding) == (1, 1, 1, 1)


def _is_pool(m):
    def two(v):
        return v == 2 or tuple(v) == (2, 2)
    return (isinstance(m, nn.MaxPool2d) and two(m.kernel_size) and two(m.stride) and m.ceil_mode
            and (m.padding == 0 or tuple(m.padding) == (0, 0)))


def compile_plan(children):
    """children: list of nn.Module in reference order -> list[Step]; raises on anything unsupported."""
    steps, i, n = [], 0, len(children)
    while i < n:
        m = children[i]
        if isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1):
            # vgg[0]: only the 1->3 pre-conv feeding the first 3x3 exists on this path
            ok = (m.in_channels == 1 and m.out_channels == 3 and i + 2 < n and _is_reflpad1(children[i + 1])
                  and isinstance(children[i + 2], nn.Conv2d) and children[i + 2].kernel_size == (3, 3)
                  and children[i + 2].in_channels == 3 and children[i + 2].out_channels == 64
                  and i + 3 < n and isinstance(children[i + 3], nn.ReLU))
            if not ok:
                raise NotImplementedError(f"unsupported 1x1 convolution at child {i}: {m}")
            steps.append(Step("first", children[i + 2], conv0=m, relu=True))
            i += 4
        elif _is_reflpad1(m):
            if not (i + 1 < n and isinstance(children[i + 1], nn.Conv2d) and children[i + 1].kernel_size == (3, 3)
                    and children[i + 1].stride == (1, 1) and children[i + 1].padding == (0, 0)):
                raise NotImplementedError(f"ReflectionPad2d at child {i} is not followed by a 3x3 Conv2d")
            conv = children[i + 1]
            i += 2
            relu = i < n and isinstance(children[i], nn.ReLU)
            i += int(relu)
            pool = relu and i < n and _is_pool(children[i])
            i += int(pool)
            if conv.out_channels == 2 and conv.in_channels == 64 and not relu:
                steps.append(Step("last", conv))
            else:
                if conv.in_channels % 16 or conv.out_channels % 64:
                    raise NotImplementedError(
                        f"3x3 conv {conv.in_channels}->{conv.out_channels}: the MFMA kernel needs Cin%16==0, Cout%64==0")
                steps.append(Step("conv", conv, relu=relu, pool=pool))
        elif isinstance(m, nn.ConvTranspose2d):
            if not (m.kernel_size == (2, 2) and m.stride == (2, 2) and m.padding == (0, 0)
                    and m.in_channels % 16 == 0 and m.out_channels % 64 == 0):
                raise NotImplementedError(f"unsupported ConvTranspose2d at child {i}: {m}")
            i += 1
            relu = i < n and isinstance(children[i], nn.ReLU)
            i += int(relu)
            steps.append(Step("convt", m, relu=relu))
        elif _is_pool(m):
            raise NotImplementedError(f"MaxPool2d at child {i} does not follow a conv+ReLU (it is fused into one)")
        else:
            raise NotImplementedError(f"unsupported layer at child {i}: {m}")
    return steps


class Plan:
    """Executable plan of one child list; owns the packed-weight cache."""

    def __init__(self, children, dtype=torch.float32, conv_algo="auto"):
        self.steps = compile_plan(list(children))
        self.dtype = dtype          # activation / packed-weight dtype: float32 (exact fp32 MFMA), bfloat16, or ops.F32S
        self.conv_algo = conv_algo  # fp32 3x3 layers: "direct" (implicit GEMM), "winograd" (F(2x2,3x3)), "auto" (by size)
        self._packed = {}

    # Winograd work items are 32 x 16 pixels x 32 channels on a persistent grid of one block per CU (256): measured over batch
    # 1..64 at 128^2 and 256^2 (scripts/algo_sweep.sh) the kernel wins from about half a wave of items up; below that the direct
    # kernel's smaller tiles fill the chip better.  The two kernels differ by fp32 rounding (3e-5 of the outputs at 256^2), so the
    # choice must not follow the exact batch size — a hologram would change with the shard it arrives in.  "auto" therefore
    # decides from PER-IMAGE quantities (H, W, Cin, Cout) and the batch the launch is PLANNED for: the top of the bucket
    # [1], [2], [3..5], [6..8], [9..12], [13..] that holds the batch (16 for the last one).  Every batch of 13 or more — all shards
    # of a multi-GPU job, every micro-batch of the bench — makes the same choice, and so do the batches inside each smaller bucket;
    # planning for the top means the real launch never has more work items than the plan assumed.
    WINO_MIN_BLOCKS = 128
    BATCH_TOPS = (1, 2, 5, 8, 12, 16)

    @classmethod
    def batch_bucket(cls, b):
        return next((t for t in cls.BATCH_TOPS if b <= t), cls.BATCH_TOPS[-1])

    # split-K (ops.conv3x3_wino(ksplit=...)): a layer short of work items deals the Cin/16 channel chunks of every tile to k <= 8
    # items (parts of at least 2 chunks, sizes differing by at most one).  k minimises a cost model of the launch, in microseconds
    # as measured on MI355X (scripts/wino_ksplit_sweep.py, scripts/wino_stamps.py: an item costs 8 us + 4.4 us per chunk whether
    # the grid is full or not; the reduction pass 3 us + its traffic at 4 TB/s): rounds of the persistent grid x the longest item,
    # plus the reduction.  Typical outcome: the k that brings the item count closest to the CU count from below.
    CUS = 256
    KSPLIT_MAX = 8
    KSPLIT_MIN_CHUNKS = 2
    ITEM_US, CHUNK_US, REDUCE_US, REDUCE_BYTES_PER_US = 8.0, 4.4, 3.0, 4.0e6

    @classmethod
    def _ksplit(cls, items, chunks, out_bytes):
        """k in [1, KSPLIT_MAX] with the lowest modelled time for `items` tiles x channel blocks of `chunks` chunks each."""
        best_k, best_t = 1, -(-items // cls.CUS) * (cls.ITEM_US + cls.CHUNK_US * chunks)
        for k in range(2, cls.KSPLIT_MAX + 1):
            if chunks // k < cls.KSPLIT_MIN_CHUNKS:
                break
            t = (-(-items * k // cls.CUS) * (cls.ITEM_US + cls.CHUNK_US * -(-chunks // k))
                 + cls.REDUCE_US + (k + 1) * out_bytes / cls.REDUCE_BYTES_PER_US)
            if t < best_t:
                best_k, best_t = k, t
        return best_k

    def _wino_choice(self, step, x):
        """(use the Winograd kernel, ksplit) for a 3x3 step on input x (NHWC)."""
        if self.dtype != torch.float32 or step.kind != "conv" or self.conv_algo == "direct" or step.conv.in_channels < 32:
            return False, 1
        if self.conv_algo == "winograd":
            return True, 1
        b, h, w, _ = x.shape
        planned = self.batch_bucket(b)
        items = planned * ((h + 31) // 32) * ((w + 15) // 16) * (step.conv.out_channels // 32)
        # the open-ended top bucket is never split: a batch of 256 would carry the reduction pass a batch of 16 asked for
        k = 1 if planned == self.BATCH_TOPS[-1] else self._ksplit(items, step.conv.in_channels // 16,
                                                                  4.0 * planned * h * w * step.conv.out_channels)
        if items * k >= self.WINO_MIN_BLOCKS:
            return True, k
        return False, 1

    def _use_winograd(self, step, x):
        return self._wino_choice(step, x)[0]

    def _weights(self, idx, step, wino=False):
        conv = step.conv
        key = tuple(v for m in (conv, step.conv0) if m is not None
                    for v in (m.weight._version, m.weight.data_ptr(), m.bias._version, m.bias.data_ptr()))
        idx = (idx, "wino") if wino else idx
        hit = self._packed.get(idx)
        if hit is not None and hit[0] == key:
            return hit[1]
        w = conv.weight.detach()
        b = conv.bias.detach().float().contiguous()
        if step.kind == "conv" and wino:
            vals = (ops.pack_conv3x3_wino(w), b)
        elif step.kind == "conv":
            vals = (ops.pack_conv3x3(w, self.dtype), b)
        elif step.kind == "convt":
            vals = (ops.pack_convt2x2(w, self.dtype), b)
        elif step.kind == "first":
            c0 = step.conv0
            vals = (c0.weight.detach().float().contiguous(), c0.bias.detach().float().contiguous(),
                    w.float().contiguous(), b)
        else:
            vals = (w.float().contiguous(), b)
        self._packed[idx] = (key, vals)
        return vals

    def invalidate(self):
        """Drop the packed weights.  The cache key (version counter + data pointer of every parameter of a step) sees
        in-place ops, ``load_state_dict`` and ``.to()``; a write THROUGH ``param.data`` (``p.data.copy_()``,
        ``p.data[...] = ``) bumps no version counter, so call this (or ``HoloSequential.invalidate_packed_weights``)
        after one."""
        self._packed.clear()

    @property
    def ends_in_planes(self):
        return bool(self.steps) and self.steps[-1].kind == "last"

    def run_step(self, idx, x, intensity=False):
        st = self.steps[idx]
        if st.kind == "conv" and x.dtype == torch.float32:
            use_wino, ksplit = self._wino_choice(st, x)
            if use_wino:
                wts = self._weights(idx, st, wino=True)
                return ops.conv3x3_wino(x, wts[0], wts[1], st.conv.out_channels, st.relu, st.pool, ksplit=ksplit)
        wts = self._weights(idx, st)
        if st.kind == "first":
            if x.dim() != 4 or x.shape[1] != 1:
                raise RuntimeError(f"the encoder expects a (B,1,H,W) hologram amplitude, got {tuple(x.shape)}")
            return ops.conv_first(x, *wts, out_dtype=self.dtype, sqrt_input=intensity)
        if st.kind in ("conv", "convt") and x.dtype != self.dtype:
            x = ops.cast(x, self.dtype)
        if st.kind == "conv":
            return ops.conv3x3(x, wts[0], wts[1], st.conv.out_channels, st.relu, st.pool)
        if st.kind == "convt":
            return ops.convt2x2(x, wts[0], wts[1], st.conv.out_channels, st.relu)
        return ops.conv_last(x, wts[0], wts[1])

    def run(self, x, intensity=False):
        """x: logical (B,C,H,W) for a plan that starts with conv_first, else NHWC (B,H,W,C).
        intensity: x is the hologram INTENSITY; the first layer's loader takes the square root (inference plan only).
        Returns NHWC, or NCHW (B,2,H,W) planes when the plan ends with conv_last.
        While autograd is recording and the input or a parameter of the plan asks for a gradient, every step runs through its
        ``autograd`` Function (fp32 only): the backward pass of the training objective."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for st in self.steps for m in (st.conv, st.conv0)
                                                                if m is not None for p in m.parameters())):
            if self.dtype != torch.float32:
                raise NotImplementedError("the backward pass exists in fp32 only: set_compute_dtype(torch.float32) for training")
            if intensity:
                raise NotImplementedError("intensity input is an inference-plan option: pass the amplitude while autograd records")
            from . import autograd as _ag
            for st in self.steps:
                x = _ag.run_step(st, x)
            return x
        if intensity and not (self.steps and self.steps[0].kind == "first"):
            raise RuntimeError("intensity input needs a plan that starts with the first encoder layer")
        for idx in range(len(self.steps)):
            x = self.run_step(idx, x, intensity and idx == 0)
        return x


class HoloSequential(nn.Sequential):
    """``nn.Sequential`` whose forward runs the fused HIP plan of its children.

    Input: the reference's logical NCHW tensor on a ROCm device.  Output: logical NCHW — a
    channels-last view of the NHWC result, or the contiguous (B,2,H,W) planes of the last decoder
    layer.  CPU tensors raise: this path has no CPU fallback."""

    def set_compute_dtype(self, dtype):
        """float32 (default: exact fp32 MFMA, the 1e-4 parity path) or bfloat16 (BASELINE config 3: bf16 convs,
        fp32 accumulation, statistics, distance head and propagator)."""
        dtype = ops.compute_dtype(dtype)      # also "f32s": fp32 values as split fp16 halves on the fp16 matrix cores
        object.__setattr__(self, "_holo_dtype", dtype)
        return self

    def invalidate_packed_weights(self):
        """Forget the packed (MFMA-layout) copies of the weights; they are rebuilt on the next forward."""
        plan = getattr(self, "_holo_plan", None)
        if plan is not None:
            plan.invalidate()
        from . import autograd as _ag       # the training path's per-version cache (writes through ``.data`` do not bump a version)
        _ag.forget_packed_weights()
        return self

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed_weights()
        return out

    def _apply(self, fn, *args, **kwargs):       # .to() / .cuda() / .float() ...
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_packed_weights()
        return out

    def set_conv_algo(self, algo):
        """fp32 3x3 layers: "auto" (default: Winograd F(2x2,3x3) where the layer fills the chip, else the direct
        implicit-GEMM kernel), "direct" or "winograd".  Both are exact fp32 arithmetic; they differ by rounding only."""
        if algo not in ("auto", "direct", "winograd"):
            raise ValueError(f"conv algo must be 'auto', 'direct' or 'winograd', got {algo!r}")
        object.__setattr__(self, "_holo_algo", algo)
        return self

    def _plan(self):
        dtype = getattr(self, "_holo_dtype", torch.float32)
        algo = getattr(self, "_holo_algo", "auto")
        sig = (tuple(id(m) for m in self), dtype, algo)
        if getattr(self, "_holo_plan_sig", None) != sig:
            object.__setattr__(self, "_holo_plan", Plan(list(self), dtype, algo))
            object.__setattr__(self, "_holo_plan_sig", sig)
        return self._holo_plan

    def run_nhwc(self, x):
        """NHWC in (or NCHW single-channel planes for a plan starting at vgg[0]) -> plan output."""
        return self._plan().run(x)

    def forward(self, input, intensity=False):
        if not input.is_cuda:
            raise RuntimeError("this network runs on MI355X HIP kernels only: move the input to a ROCm device "
                               "(the CPU restatement lives in oracle/ and is test infrastructure)")
        plan = self._plan()
        x = input if plan.steps and plan.steps[0].kind == "first" else ops.to_nhwc(input)
        y = plan.run(x, intensity)
        return y if plan.ends_in_planes else ops.as_nchw_view(y)
