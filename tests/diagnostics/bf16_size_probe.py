"""Diagnostic (GPU): how far is the bf16 conv path from its CPU bf16-rounding emulation and from fp32, per image size and
per encoder stage?  Decides what the config-3-sized bf16 test can ask.  Usage: python tests/diagnostics/bf16_size_probe.py"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from hip_helpers import DEV, bf16_emulated_field_retrieval, build_network, holo_generator, orc  # noqa: E402
from style_transfer_based_holographic_imaging_amd import synth  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def l2(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())


def main():
    network, sds = build_network(0)
    hg = holo_generator()
    style = torch.load(os.path.join(os.path.dirname(os.path.dirname(HERE)), "style_representation", "MNIST", "style_vector.pt"),
                       map_location="cpu").float()
    for size in (128, 256, 512):
        ph = synth.synthetic_phase_objects(1, size, seed=1235)
        d_c = synth.synthetic_distances(1, seed=1235)
        with torch.no_grad():
            content = torch.sqrt(hg(torch.full_like(ph, 0.6).to(DEV), ph.to(DEV), d_c.to(DEV)))
            a32, p32, _ = network.field_retrieval(content, style.to(DEV), 1.0, True)
            f32 = [f.float().cpu() for f in network.encode_with_intermediate(content)]
            network.set_compute_dtype(torch.bfloat16)
            a16, p16, _ = network.field_retrieval(content, style.to(DEV), 1.0, True)
            f16 = [f.float().cpu() for f in network.encode_with_intermediate(content)]
            network.set_compute_dtype(torch.float32)
            ea, ep, _ = bf16_emulated_field_retrieval(content.cpu(), style, sds)
            oa, op_, _ = orc.net_field_retrieval(content.cpu(), style, *sds, 1.0, True)
            # emulated encoder stages
            vgg = sds[0]

            def r(t):
                return t.to(torch.bfloat16).float()
            x = content.cpu()
            emu, bounds = [], {4: 0, 11: 1, 18: 2, 31: 3}
            plan = orc.ENCODER_PLAN
            for i, step in enumerate(plan):
                if step[0] == "conv1x1":
                    x = F.conv2d(x, vgg[f"{step[1]}.weight"], vgg[f"{step[1]}.bias"])
                elif step[0] == "conv":
                    w = vgg[f"{step[1]}.weight"]
                    w = w if step[1] == 2 else r(w)
                    x = F.relu(orc._conv3x3_reflect(x, w, vgg[f"{step[1]}.bias"]))
                    nxt_pool = i + 1 < len(plan) and plan[i + 1][0] == "pool"
                    if not nxt_pool:
                        x = r(x)
                else:
                    x = r(F.max_pool2d(x, 2, 2, 0, ceil_mode=True))
                if step[0] == "conv" and step[1] in (2, 9, 16, 29):
                    emu.append(x.clone())
        print(f"size {size}: amp  hip16-vs-emu {l2(a16, ea):.3f}  hip16-vs-fp32 {l2(a16, a32):.3f}  emu-vs-fp32 {l2(ea, oa):.3f}   "
              f"phase {l2(p16, ep):.3f} / {l2(p16, p32):.3f} / {l2(ep, op_):.3f}")
        for k in range(4):
            print(f"   enc{k + 1}: hip16-vs-emu L2 {l2(f16[k], emu[k]):.4f} max {float((f16[k] - emu[k]).abs().max() / emu[k].abs().max()):.4f}   "
                  f"hip16-vs-fp32 {l2(f16[k], f32[k]):.4f}   emu-vs-fp32 {l2(emu[k], f32[k]):.4f}")


if __name__ == "__main__":
    main()
