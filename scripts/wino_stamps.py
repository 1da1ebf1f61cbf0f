#!/usr/bin/env python3
"""Where block 0 / thread 0 of the Winograd kernel spends its cycles (needs a side build with -DWINO_STAMPS=1):
  python -c "from style_transfer_based_holographic_imaging_amd import build; build.build(extra_flags=('-DWINO_STAMPS=1',), out='scratch/libholo_wstamps.so')"
  HOLO_HIP_LIB=$PWD/scratch/libholo_wstamps.so python scripts/wino_stamps.py"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_based_holographic_imaging_amd import ops  # noqa: E402
from style_transfer_based_holographic_imaging_amd._lib import LIB_PATH  # noqa: E402

lib = ctypes.CDLL(LIB_PATH)
if not hasattr(lib, "holo_wino_debug_read"):
    raise SystemExit("this library was not built with -DWINO_STAMPS=1")
dev = torch.device("cuda:0")
names = ["step0", "sync0", "step1", "sync1", "loop top (+plan)", "epilogue", "zero+switch"]
shapes = ((64, 64, 256), (64, 256, 64), (64, 512, 32))
if len(sys.argv) > 1:     # e.g. 5,256,32 5,512,16  (batch, channels, side)
    shapes = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
for b, c, hw in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, c, hw, hw, generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) * math.sqrt(2 / (9 * c))
    bias = torch.zeros(c, device=dev)
    xa, wp = ops.to_nhwc(x.to(dev)), ops.pack_conv3x3_wino(wt.to(dev))
    ops.conv3x3_wino(xa, wp, bias, c, True, False)
    torch.cuda.synchronize()
    lib.holo_wino_debug_reset()
    ops.conv3x3_wino(xa, wp, bias, c, True, False)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.holo_wino_debug_read(out)
    chunks, tiles = max(out[7], 1), max(out[8], 1)
    tot = sum(out[i] for i in range(7))
    print(f"{c}ch {hw}x{hw}: {tiles} tiles, {chunks} chunks, total {tot} ticks = {tot / tiles:.0f} per tile")
    for i, n in enumerate(names):
        per = out[i] / (chunks if i < 5 else tiles)
        print(f"   {n:18s} {out[i]:12d} ticks  {100 * out[i] / tot:5.1f} %   {per:9.0f} per {'chunk' if i < 5 else 'tile'}")
    # inside the epilogue (the "epilogue" line above is only its last part when these are present)
    for i, n in ((9, "epilogue: addressing, bias request, nops"), (10, "epilogue: accumulator reads + output transform")):
        print(f"   {n:48s} {out[i] / tiles:9.0f} per tile")
