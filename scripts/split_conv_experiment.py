"""EXPERIMENT: fp32-accurate 3x3 conv on split-fp16 operands (holo_conv3x3_f32s) — accuracy vs float64 and speed."""
import ctypes, math, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from style_transfer_based_holographic_imaging_amd import ops
from style_transfer_based_holographic_imaging_amd._lib import LIB_PATH
lib = ctypes.CDLL(LIB_PATH)
P, I = ctypes.c_void_p, ctypes.c_int
lib.holo_conv3x3_f32s.restype = I
lib.holo_conv3x3_f32s.argtypes = [P, P, P, P, I, I, I, I, I, I, I, P]
dev = torch.device('cuda:0')

def split_planar(t):                      # (..., C) fp32 -> (..., C) 4-byte elements, planar per 16 channels
    hi = t.half()
    lo = ((t - hi.float()) * 2048.0).half()
    c = t.shape[-1]
    g = torch.stack([hi.reshape(*t.shape[:-1], c // 16, 16), lo.reshape(*t.shape[:-1], c // 16, 16)], dim=-2)   # (..., C/16, 2, 16)
    return g.reshape(*t.shape[:-1], c * 2).contiguous().view(torch.int32)                                      # C dwords

def pack_w(w):                            # OIHW -> [Cin/32][tap][Cout][32]
    co, ci = w.shape[:2]
    t = w.reshape(co, ci // 32, 32, 9).permute(1, 3, 0, 2).contiguous()     # (chunk, tap, co, 32)
    return split_planar(t)

def run(b, h, wd, cin, cout, reps=20, relu=True, pool=False):
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(b, cin, h, wd, generator=g).abs() * 0.7            # post-ReLU-like activations
    w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    bias = torch.randn(cout, generator=g) * 0.1
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    xs, ws = split_planar(x_nhwc).to(dev), pack_w(w).to(dev)
    biasd = bias.to(dev)
    ho, wo = ((h + 1) // 2, (wd + 1) // 2) if pool else (h, wd)
    y = torch.empty(b, ho, wo, cout, device=dev)
    def call():
        rc = lib.holo_conv3x3_f32s(xs.data_ptr(), ws.data_ptr(), biasd.data_ptr(), y.data_ptr(), b, h, wd, cin, cout, int(relu), int(pool), None)
        assert rc == 0
    call(); torch.cuda.synchronize()
    ref64 = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode='reflect'), w.double(), bias.double())
    if relu: ref64 = F.relu(ref64)
    if pool: ref64 = F.max_pool2d(ref64, 2, 2, 0, ceil_mode=True)
    got = y.permute(0, 3, 1, 2).cpu().double()
    e_split = float((got - ref64).abs().max() / ref64.abs().max())
    # the fp32 MFMA kernel on the same data
    xp = ops.to_nhwc(x.to(dev)); wp = ops.pack_conv3x3(w.to(dev))
    y32 = ops.conv3x3(xp, wp, biasd, cout, relu, pool)
    e_f32 = float((ops.as_nchw_view(y32).cpu().double() - ref64).abs().max() / ref64.abs().max())
    e_cpu = float((F.relu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w, bias)).double() - (F.relu(F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode='reflect'), w.double(), bias.double())))).abs().max() / ref64.abs().max()) if not pool else float('nan')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): call()
    torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): ops.conv3x3(xp, wp, biasd, cout, relu, pool)
    torch.cuda.synchronize(); t32 = (time.perf_counter() - t0) / reps
    fl = 2.0 * 9 * cin * cout * b * h * wd
    print(f"B={b} {cin}->{cout} @{h}x{wd} pool={pool}: err vs fp64  split {e_split:.2e}  fp32-MFMA {e_f32:.2e}  torch-CPU-fp32 {e_cpu:.2e} | "
          f"split {ts*1e3:.3f} ms = {fl/ts/1e12:.0f} TFLOP/s-equiv, fp32 MFMA {t32*1e3:.3f} ms = {fl/t32/1e12:.0f} TFLOP/s", flush=True)

run(2, 24, 40, 64, 128, reps=3)
run(1, 21, 37, 128, 64, reps=3, pool=True)
run(64, 64, 64, 256, 256)
run(64, 256, 256, 64, 64)
run(64, 32, 32, 512, 512)
run(64, 128, 128, 128, 128)
