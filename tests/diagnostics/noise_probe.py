#!/usr/bin/env python3
"""Where does fp32 rounding noise enter?  Per stage: HIP fp32 and CPU-oracle fp32, each against a
float64 evaluation of the same network (test infrastructure; run on the GPU box)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import holo_oracle as orc  # noqa: E402
from hip_helpers import DEV, build_network  # noqa: E402
from style_transfer_based_holographic_imaging_amd import ops, synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = 2


def rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())


def main():
    network, sds = build_network(0)
    style = torch.load(os.path.join(REPO, "style_representation", "MNIST", "style_vector.pt"), map_location="cpu").float()
    hp = orc.HoloParams()
    ph = synth.synthetic_phase_objects(B, S, seed=77)
    d_c = synth.synthetic_distances(B, seed=77)
    dbl = [{k: v.double() for k, v in sd.items()} for sd in sds]
    rows = []
    with torch.no_grad():
        content = torch.sqrt(orc.holo_generator(hp, torch.full_like(ph, 0.6), ph, d_c)).float()
        f32 = orc.encode_with_intermediate(content, sds[0])
        f64 = orc.encode_with_intermediate(content.double(), dbl[0])
        fh = network.encode_with_intermediate(content.to(DEV))
        for k in range(4):
            rows.append((f"enc{k + 1}", rel(fh[k], f64[k]), rel(f32[k], f64[k])))
        # from here on feed every path the SAME float64-rounded input so stages are isolated
        feat = f64[3].float()
        m32, s32 = orc.calc_mean_std(feat)
        m64, s64 = orc.calc_mean_std(feat.double())
        mh, sh = ops.channel_mean_std(ops.to_nhwc(feat.to(DEV)))
        rows.append(("mean", rel(mh.view_as(m64), m64), rel(m32, m64)))
        rows.append(("std", rel(sh.view_as(s64), s64), rel(s32, s64)))
        t64 = orc.adain_with_style_vector(feat.double(), style.double(), 1.0)
        t32 = orc.adain_with_style_vector(feat, style, 1.0)
        th = ops.adain_apply(ops.to_nhwc(feat.to(DEV)), mh, sh, style[:1].to(DEV), style[1:].to(DEV), 1.0)
        rows.append(("adain", rel(ops.as_nchw_view(th), t64), rel(t32, t64)))
        t_in = t64.float()
        o64, st64 = orc.decoder_forward(t_in.double(), dbl[1], return_stages=True)
        o32, st32 = orc.decoder_forward(t_in, sds[1], return_stages=True)
        plan = network.decoder._plan()
        x = ops.to_nhwc(t_in.to(DEV))
        for idx, st in enumerate(plan.steps):
            x = plan.run_step(idx, x)
            got = x if st.kind == "last" else ops.as_nchw_view(x)
            rows.append((f"dec{idx}:{st.kind}", rel(got, st64[idx]), rel(st32[idx], st64[idx])))
        # one isolated 512->512 conv on identical input
        w, b = sds[1]["1.weight"], sds[1]["1.bias"]
        c64 = torch.relu(orc._conv3x3_reflect(t_in.double(), w.double(), b.double()))
        c32 = torch.relu(orc._conv3x3_reflect(t_in, w, b))
        ch = ops.conv3x3(ops.to_nhwc(t_in.to(DEV)), ops.pack_conv3x3(w.to(DEV)), b.to(DEV), 512, True, False)
        rows.append(("single conv512 (K=4608)", rel(ops.as_nchw_view(ch), c64), rel(c32, c64)))
    print(f"S={S}  stage: HIP-fp32 vs fp64 | CPU-fp32 vs fp64   (max|d|/max|ref|)")
    for name, a, b in rows:
        print(f"  {name:26s} {a:10.2e} | {b:10.2e}")


if __name__ == "__main__":
    main()
