#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference's modules are imported from where they lie (never copied); scikit-image is absent
there, so an empty ``skimage.restoration`` stub is placed in ``sys.modules`` to let
``utils/Forward_model.py`` import (its ``unwrap`` branch is never taken here).  Weights are the
seeded He-normal tensors of ``style_transfer_based_holographic_imaging_amd.synth`` loaded through
``load_state_dict`` into the reference's own ``net.vgg`` / ``net.decoder`` / ``net.Distance_G``.

Outputs (float32 unless noted, all small):
  golden_net_s128.npz    full entry-loop outputs for test_data file 0 (5 holograms, 128x128)
  golden_net_sums.npz    checksums of the same outputs for test_data files 1..19
  golden_small.npz       S=32 case with intermediates (encoder stages, stats, AdaIN, decoder)
  golden_edges.npz       B=1, multi-style, alpha 0/0.5, odd size 36, other style vectors
  golden_asm.npz         ASM / Holo_Generator modes on seeded fields (incl. zero_padding=False)
  golden_forward.npz     Net.forward (the training objective's forward values: losses, g_t, style reconstruction, distances)
  golden_backward.npz    gradients of a scalar objective built from all seven results of Net.forward (torch.autograd on the
                         reference): per parameter of the decoder and the distance head its sum, L2 norm and 256 fixed samples
"""
import argparse
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

warnings.filterwarnings("ignore")


def import_reference():
    """Import the reference's modules from where they lie.  The repo has same-named drop-in modules (``net``, ``function``
    and a regular ``utils`` package, which would beat the reference's namespace package wherever it sits on ``sys.path``), so
    ``utils`` is bound to the reference's directory explicitly and EVERY imported module is checked to come from there."""
    stub = types.ModuleType("skimage")
    stub.restoration = types.ModuleType("skimage.restoration")
    stub.restoration.unwrap_phase = None
    sys.modules["skimage"] = stub
    sys.modules["skimage.restoration"] = stub.restoration
    for name in [m for m in sys.modules if m in ("net", "function", "utils") or m.startswith("utils.")]:
        del sys.modules[name]        # e.g. the repo's own modules imported earlier in this process
    ref_utils = types.ModuleType("utils")
    ref_utils.__path__ = [os.path.join(REF, "utils")]
    sys.modules["utils"] = ref_utils
    sys.path.insert(0, REF)          # reference's net / function win over the repo's
    if REPO not in sys.path:
        sys.path.append(REPO)        # for the synth module only
    import net as ref_net
    import function as ref_function
    import utils.Forward_model as ref_fm
    import utils.Angular_Spectrum_Method as ref_asm
    import utils.functions as ref_fn
    for mod in (ref_net, ref_function, ref_fm, ref_asm, ref_fn):
        assert os.path.realpath(mod.__file__).startswith(REF + os.sep), f"{mod.__name__} came from {mod.__file__}, not the reference"
    return ref_net, ref_function, ref_fm.Holo_Generator, ref_asm.ASM, ref_fn.field_retrieval


def build_network(ref_net, seed):
    import copy
    from style_transfer_based_holographic_imaging_amd import synth
    vgg_sd, dec_sd, dist_sd = synth.synthetic_state_dicts(seed, full_vgg=True)
    vgg = copy.deepcopy(ref_net.vgg)
    vgg.load_state_dict(vgg_sd)
    vgg = torch.nn.Sequential(*list(vgg.children())[:31])   # test_field_retrieval_mnist.py:89
    dec = copy.deepcopy(ref_net.decoder)
    dec.load_state_dict(dec_sd)
    dist = ref_net.Distance_G()
    dist.load_state_dict(dist_sd)
    network = ref_net.Net(vgg, dec, None, dist).eval()
    return network


def sums(t):
    """size-independent checksums of one tensor: mean, L2, max|.|, strided 16x16 subsample."""
    t = t.detach().double()
    sub = t[..., :: max(1, t.shape[-2] // 16), :: max(1, t.shape[-1] // 16)][..., :16, :16]
    return (np.array([t.mean().item(), t.norm().item(), t.abs().max().item()], np.float64),
            sub.float().numpy())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)

    ref_net, ref_function, Holo_Generator, ASM, field_retrieval = import_reference()
    from style_transfer_based_holographic_imaging_amd import synth

    hp = types.SimpleNamespace(wavelength=532e-9, pixel_size=1.5e-6, distance_normalize=1.0,
                               distance_normalize_constant=0.0, phase_normalize=1.0)
    model_forward = Holo_Generator(hp)
    network = build_network(ref_net, args.seed)
    style = torch.load(f"{REF}/style_representation/MNIST/style_vector.pt", map_location="cpu").float()

    def entry_step(content_holo, distance_style, style_vec=style):
        content = torch.sqrt(content_holo).float()
        with torch.no_grad():
            amp, ph, dist = field_retrieval(network, content, style_vec, 1.0, True)
            field = model_forward(amp, ph * hp.phase_normalize,
                                  -distance_style - 2 * hp.distance_normalize_constant,
                                  complex_number=True)
            amp_foc, ph_foc = model_forward(amp, ph * hp.phase_normalize,
                                            -distance_style - 2 * hp.distance_normalize_constant,
                                            return_field=True, unwrap=False)
        return dict(amp=amp, ph=ph, dist=dist, amp_foc=amp_foc, ph_foc=ph_foc,
                    field_re=field.real.float(), field_im=field.imag.float())

    # ---- 1. full outputs for file 0, checksums for files 1..19 -------------------------------
    def load(i, kind):
        return torch.load(f"{REF}/test_data/test_{kind}_{i}.pt", map_location="cpu")

    out0 = entry_step(load(0, "content_holo"), load(0, "distance_style").float())
    np.savez_compressed(os.path.join(args.out, "golden_net_s128.npz"),
                        **{k: v.contiguous().numpy() for k, v in out0.items()}, seed=args.seed)
    sums_d = {}
    for i in range(1, 20):
        o = entry_step(load(i, "content_holo"), load(i, "distance_style").float())
        for k, v in o.items():
            if k == "dist":
                sums_d[f"f{i}_dist"] = v.numpy()
            else:
                s, sub = sums(v)
                sums_d[f"f{i}_{k}_stats"] = s
                sums_d[f"f{i}_{k}_sub"] = sub
    np.savez_compressed(os.path.join(args.out, "golden_net_sums.npz"), **sums_d, seed=args.seed)

    # ---- 2. S=32 case with intermediates ------------------------------------------------------
    small = {}
    ph_obj = synth.synthetic_phase_objects(2, 32, seed=11)
    d_c = torch.tensor([0.4, 0.7]).view(-1, 1, 1, 1)
    with torch.no_grad():
        holo = model_forward(torch.full_like(ph_obj, 0.6), ph_obj, d_c)
        content = torch.sqrt(holo).float()
        feats = network.encode_with_intermediate(content)
        mean, std = ref_function.calc_mean_std(feats[-1])
        amp, ph, dist = network.field_retrieval(content, style, 1.0, True)
        size = feats[-1].size()
        t = (feats[-1] - mean.expand(size)) / std.expand(size)
        t = t * style[1:].repeat(2, 1, 1, 1).expand(size) + style[:1].repeat(2, 1, 1, 1).expand(size)
        x = t
        dec_stage = {}
        for idx, layer in enumerate(network.decoder):
            x = layer(x)
            if idx in (7, 21, 29):      # after the ReLU that follows each ConvTranspose2d
                dec_stage[idx] = x
        adain_img = ref_function.adaptive_instance_normalization(feats[-1], feats[-1].flip(0))
    small.update(content=content, holo=holo, enc1=feats[0], enc2=feats[1], enc3=feats[2], enc4=feats[3],
                 mean=mean, std=std, t=t, dec7=dec_stage[7], dec21=dec_stage[21], dec29=dec_stage[29],
                 g_t=x, amp=amp, ph=ph, dist=dist, adain_img=adain_img, phase_obj=ph_obj, d_c=d_c)
    np.savez_compressed(os.path.join(args.out, "golden_small.npz"),
                        **{k: v.contiguous().numpy() for k, v in small.items()}, seed=args.seed)

    # ---- 3. edge cases ----------------------------------------------------------------------------
    edges = {}
    with torch.no_grad():
        c32 = content
        # B == 1 path (net.py:255-256)
        a1, p1, d1 = network.field_retrieval(c32[:1], style, 1.0, True)
        edges.update(b1_amp=a1, b1_ph=p1, b1_dist=d1)
        # multiple-style method (net.py:235-236): first half rows = means, second half = stds
        sv2 = synth.synthetic_style_vector(5)
        multi = torch.cat([style[:1], sv2[:1], style[1:], sv2[1:]], 0)
        am, pm, dm = network.field_retrieval(c32, multi, 1.0, True)
        edges.update(multi_style=multi, multi_amp=am, multi_ph=pm, multi_dist=dm)
        # alpha blend (net.py:246)
        for al, tag in ((0.0, "a0"), (0.5, "a05")):
            aa, pa = network.field_retrieval(c32, style, al)
            edges[f"{tag}_amp"], edges[f"{tag}_ph"] = aa, pa
        # odd size: ceil_mode pooling 36 -> 18 -> 9 -> 5, decoder 5 -> 10 -> 20 -> 40
        c36 = torch.sqrt(model_forward(torch.full((2, 1, 36, 36), 0.6),
                                       synth.synthetic_phase_objects(2, 36, seed=12), d_c)).float()
        f36 = network.encode_with_intermediate(c36)
        a36, p36, d36 = network.field_retrieval(c36, style, 1.0, True)
        edges.update(odd_content=c36, odd_enc3=f36[2], odd_enc4=f36[3], odd_amp=a36, odd_ph=p36, odd_dist=d36)
        # the other two bundled style vectors
        for name in ("polystyrene_bead", "red_blood_cell"):
            sv = torch.load(f"{REF}/style_representation/{name}/style_vector.pt", map_location="cpu").float()
            a_s, p_s, d_s = network.field_retrieval(c32, sv, 1.0, True)
            edges[f"{name}_amp"], edges[f"{name}_ph"], edges[f"{name}_dist"] = a_s, p_s, d_s
        # distance head alone on seeded statistics (B=3) and its B=1 equality
        g = torch.Generator().manual_seed(3)
        mu = torch.randn(3, 512, 1, 1, generator=g).abs()
        sd = torch.rand(3, 512, 1, 1, generator=g) + 0.1
        edges.update(dh_mean=mu, dh_std=sd, dh_out=network.distance_g((mu, sd)))
    np.savez_compressed(os.path.join(args.out, "golden_edges.npz"),
                        **{k: v.contiguous().numpy() for k, v in edges.items()}, seed=args.seed)

    # ---- 4. ASM / Holo_Generator modes on seeded fields ------------------------------------------
    asm_d = {}
    g = torch.Generator().manual_seed(21)
    for s in (16, 64):
        amp_in = torch.rand(3, 1, s, s, generator=g) + 0.2
        ph_in = (torch.rand(3, 1, s, s, generator=g) - 0.5) * 4.0
        d_mm = torch.tensor([0.35, -0.2, 0.8]).view(-1, 1, 1, 1)
        with torch.no_grad():
            fld = model_forward(amp_in, ph_in, d_mm, complex_number=True)
            a_o, p_o = model_forward(amp_in, ph_in, d_mm, return_field=True)
            inten = model_forward(amp_in, ph_in, d_mm)
            o_low = amp_in * torch.exp(1j * ph_in)
            nopad = ASM(o_low, hp.wavelength, d_mm * 1e-3, hp.pixel_size, zero_padding=False)
            scal = ASM(o_low, hp.wavelength, 3.0e-4, hp.pixel_size, zero_padding=True)
        asm_d.update({f"s{s}_amp_in": amp_in, f"s{s}_ph_in": ph_in, f"s{s}_d_mm": d_mm,
                      f"s{s}_field_re": fld.real.float(), f"s{s}_field_im": fld.imag.float(),
                      f"s{s}_amp": a_o, f"s{s}_ph": p_o, f"s{s}_intensity": inten,
                      f"s{s}_nopad_re": nopad.real.float(), f"s{s}_nopad_im": nopad.imag.float(),
                      f"s{s}_scalar_re": scal.real.float(), f"s{s}_scalar_im": scal.imag.float()})
    # multi-channel field (C=2) goes through the same code (batch, c, Sh, Sw = O.shape)
    with torch.no_grad():
        o2 = torch.complex(torch.randn(2, 2, 16, 16, generator=g), torch.randn(2, 2, 16, 16, generator=g))
        r2 = ASM(o2, hp.wavelength, torch.tensor([1e-4, -3e-4]).view(-1, 1, 1, 1), hp.pixel_size, zero_padding=True)
    asm_d.update(c2_in_re=o2.real, c2_in_im=o2.imag, c2_re=r2.real.float(), c2_im=r2.imag.float())
    np.savez_compressed(os.path.join(args.out, "golden_asm.npz"),
                        **{k: v.contiguous().numpy() for k, v in asm_d.items()})

    # ---- 5. Net.forward: the training objective's forward values (net.py:199-226) ----------------------
    # content / style holograms of the 32x32 case, alpha 1.0 and 0.6; every return form of the method
    fwd = {}
    ph_style = synth.synthetic_phase_objects(2, 32, seed=13)
    with torch.no_grad():
        style_img = torch.sqrt(model_forward(torch.full_like(ph_style, 0.6), ph_style, torch.tensor([0.2, 0.2]).view(-1, 1, 1, 1))).float()
        for al, tag in ((1.0, "a1"), (0.6, "a06")):
            lc, ls, g_t, g_ph, s_re, d_c_pred, d_s_pred = network(content, style_img, al, True, True)
            fwd.update({f"{tag}_loss_c": lc, f"{tag}_loss_s": ls, f"{tag}_g_t": g_t, f"{tag}_g_t_phase": g_ph,
                        f"{tag}_style_re": s_re, f"{tag}_d_content": d_c_pred, f"{tag}_d_style": d_s_pred})
        lc2, ls2 = network(content, style_img)
        five = network(content, style_img, 1.0, True)
        fwd.update(plain_loss_c=lc2, plain_loss_s=ls2, five_style_re=five[4], style_img=style_img, content=content)
    np.savez_compressed(os.path.join(args.out, "golden_forward.npz"),
                        **{k: v.contiguous().numpy() for k, v in fwd.items()}, seed=args.seed)

    # ---- 6. Net.forward backward: the reference's own gradients (torch.autograd over its aten graph) ------------
    # objective = loss_c + 10 loss_s + <g_t, u1> + <g_t_phase, u2> + <style_re, u3> + <d_content, u4> + <d_style, u5> with fixed
    # cosine patterns u (no RNG): every output of the method feeds the gradient.  alpha 0.6 exercises the blend.
    def pattern(t, k):
        idx = torch.arange(t.numel(), dtype=torch.float64).view(t.shape)
        return torch.cos(idx * (0.37 + 0.11 * k) + k).float()

    for p in network.decoder.parameters():
        p.requires_grad_(True)
    for p in network.distance_g.parameters():
        p.requires_grad_(True)
    network.zero_grad()
    lc, ls, g_t, g_ph, s_re, d_c_pred, d_s_pred = network(content, style_img, 0.6, True, True)
    objective = (lc + 10.0 * ls + (g_t * pattern(g_t, 1)).mean() + (g_ph * pattern(g_ph, 2)).mean() + (s_re * pattern(s_re, 3)).mean()
                 + (d_c_pred * pattern(d_c_pred, 4)).sum() + (d_s_pred * pattern(d_s_pred, 5)).sum())
    objective.backward()
    bwd = {"objective": objective.detach()}
    for prefix, module in (("decoder", network.decoder), ("distance_g", network.distance_g)):
        for name, p in module.named_parameters():
            gflat = p.grad.detach().reshape(-1)
            pick = torch.linspace(0, gflat.numel() - 1, min(256, gflat.numel())).round().long()
            bwd[f"{prefix}.{name}.sum"] = gflat.double().sum().float()
            bwd[f"{prefix}.{name}.norm"] = gflat.double().norm().float()
            bwd[f"{prefix}.{name}.samples"] = gflat[pick].clone()
    np.savez_compressed(os.path.join(args.out, "golden_backward.npz"),
                        **{k: v.contiguous().numpy() for k, v in bwd.items()}, seed=args.seed)

    for f in sorted(os.listdir(args.out)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(args.out, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
