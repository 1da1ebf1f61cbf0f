#!/usr/bin/env python3
"""What the reference's own modules would do on an MI355X through stock PyTorch-ROCm (MIOpen convolutions,
hipFFT): the same step as bench.py (field_retrieval + back-propagation, pre-unwrap) with aten ops on `cuda`.
Diagnostic only (test infrastructure: it drives the oracle's functional restatement on the device); the
number is quoted in DESIGN.md next to the HIP path's.

  python tests/diagnostics/torch_rocm_baseline.py [--batch 64] [--size 256] [--steps 10]
"""
import argparse
import json
import os
import sys
import time
from math import pi

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import holo_oracle as orc  # noqa: E402
from style_transfer_based_holographic_imaging_amd import synth  # noqa: E402


def asm_device(O, lamb, d, px):
    """utils/Angular_Spectrum_Method.py:7-36 with the float64 grid built on the host and uploaded per call,
    as the reference does."""
    b, c, sh, sw = O.shape
    O = F.pad(O, (sh // 2, sh // 2, sw // 2, sw // 2), mode="replicate")
    g = torch.from_numpy(np.repeat(orc.transfer_grid(sh, sw, lamb, px, True)[None, None], b, axis=0)).to(O.device)
    h = torch.exp(1j * 2 * pi * d * g)
    o_fft = torch.fft.fftshift(torch.fft.fft2(O), dim=(-2, -1))
    out = torch.fft.ifft2(torch.fft.ifftshift(h * o_fft, dim=(-2, -1)))
    nh, nw = out.shape[-2:]
    return out[:, :, (nh - sh) // 2:(nh + sh) // 2, (nw - sh) // 2:(nw + sh) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    b, s = args.batch, args.size
    sds = synth.synthetic_state_dicts(0, full_vgg=True)
    style = torch.load(os.path.join(REPO, "style_representation", "MNIST", "style_vector.pt"), map_location="cpu").float().to(dev)
    hp = orc.HoloParams()
    ph = synth.synthetic_phase_objects(b, s, seed=1234)
    content = (0.3 + 0.6 * ph).to(dev)          # timing is data-independent; no CPU propagation at this size
    d_style = torch.full((b, 1, 1, 1), 0.2, device=dev)
    out = {}
    for label, dtype, cl in (("fp32_nchw", torch.float32, False), ("fp32_channels_last", torch.float32, True),
                             ("bf16_channels_last", torch.bfloat16, True)):
        dsd = [{k: v.to(dev, dtype) for k, v in sd.items()} for sd in sds]
        if cl:
            dsd = [{k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v) for k, v in sd.items()} for sd in dsd]
        x = content.to(dtype)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        st = style.to(dtype)

        def step():
            with torch.no_grad():
                amp, p, d = orc.field_retrieval(x, st, *dsd, 1.0, True)
                dd = ((-d_style + hp.distance_normalize_constant) * hp.distance_normalize) * 1e-3
                o = amp.float() * torch.exp(1j * p.float())
                f = asm_device(o, hp.wavelength, dd, hp.pixel_size)
                return torch.abs(f).float(), torch.angle(f).float()

        for bench_flag in (False, True):
            torch.backends.cudnn.benchmark = bench_flag
            try:
                for _ in range(args.warmup):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / args.steps
                out[f"{label}{'_tuned' if bench_flag else ''}"] = {"holograms_per_s": round(b / dt, 1), "ms_per_step": round(dt * 1e3, 2)}
            except Exception as e:  # noqa: BLE001
                out[f"{label}{'_tuned' if bench_flag else ''}"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
            print(json.dumps(out), flush=True)
    print(json.dumps({"torch": torch.__version__, "batch": b, "size": s, "result": out}))


if __name__ == "__main__":
    main()
