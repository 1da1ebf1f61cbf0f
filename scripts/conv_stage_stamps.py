#!/usr/bin/env python3
"""Where one wave's conv stage spends its cycles (needs a library built with -DHOLO_CONV_STAMPS=1):

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DHOLO_CONV_STAMPS=1 -o /tmp/libholo_stamps.so \\
        style_transfer_based_holographic_imaging_amd/csrc/*.hip
  HOLO_HIP_LIB=/tmp/libholo_stamps.so python scripts/conv_stage_stamps.py

Prints, for block 1000 / thread 0 of a 256->256 @64x64 layer at batch 64, the cycle-counter ticks per stage, of issuing the
staging loads, and of the wait + barrier at the stage end, for the fp32, bf16 and f32s kernels (ratios are the message)."""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_based_holographic_imaging_amd import ops  # noqa: E402
from style_transfer_based_holographic_imaging_amd._lib import LIB_PATH  # noqa: E402

lib = ctypes.CDLL(LIB_PATH)
if not hasattr(lib, "holo_debug_read"):
    raise SystemExit("this library was not built with -DHOLO_CONV_STAMPS=1 (see the docstring)")
dev = torch.device("cuda:0")
for dt, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16"), (ops.F32S, "f32s")):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 256, 64, 64, generator=g)
    wt = torch.randn(256, 256, 3, 3, generator=g) * math.sqrt(2 / (9 * 256))
    bias = torch.zeros(256, device=dev)
    xa, wp = ops.cast(ops.to_nhwc(x.to(dev)), dt), ops.pack_conv3x3(wt.to(dev), dt)
    ops.conv3x3(xa, wp, bias, 256, True, False)
    torch.cuda.synchronize()
    lib.holo_debug_reset()
    ops.conv3x3(xa, wp, bias, 256, True, False)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib.holo_debug_read(out)
    n = max(out[3], 1)
    print(f"{name}: {out[3]} stages, ticks/stage {out[0] / n:.0f}, staging-load issue {out[1] / n:.0f} ({100 * out[1] / max(out[0], 1):.0f} %), "
          f"wait + barrier {out[2] / n:.0f} ({100 * out[2] / max(out[0], 1):.0f} %)")
