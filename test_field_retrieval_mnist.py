#!/usr/bin/env python3
"""Entry point of the field-retrieval demo, drop-in for the reference's script of the same name.

Same flags and defaults as /root/reference/test_field_retrieval_mnist.py:39-62, same wiring
(:76-94) and the same loop over the 20 bundled files (:111-142), running on the MI355X HIP path:

    python test_field_retrieval_mnist.py --device cuda

Differences, all additive:
  * `--device` defaults to `cuda`: this implementation is HIP-only (`--device cpu` stops with a message).
  * the reference's trained weights are not distributed with it (`.MISSING_LARGE_BLOBS`); when the files
    under --decoder_root / --vgg are absent, `--synthetic-weights SEED` runs the identical pipeline on
    seeded He-normal weights (numbers are then meaningless as reconstructions, but every kernel runs).
  * tensors are loaded with map_location (the bundled .pt files carry a cuda:0 storage tag).
  * `--compute-dtype bf16|f32s` selects the optional conv-stack arithmetic (default f32: exact fp32 MFMA).
  * the focused phase is unwrapped on the GPU (`holo_unwrap2d_f32`) instead of a host loop over scikit-image;
    `--unwrap off` keeps it wrapped.
"""
import argparse
import copy
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn


# (flag, type, default) — names and defaults of the reference's argparse block (test_field_retrieval_mnist.py:39-62);
# only --device differs (this implementation is HIP-only)
REFERENCE_FLAGS = (
    ("data_name", str, "MNIST"), ("decoder_root", str, "./models/MNIST"), ("vgg", str, "./models/vgg_normalised.pth"),
    ("device", str, "cuda"), ("exp_name", str, "MNIST_test"), ("save_dir", str, "./output"), ("save_ext", str, ".png"),
    ("img_size", int, 128), ("n_layer", int, 4), ("unknown_distance", int, 1),
    ("wavelength", float, 532e-9), ("pixel_size", float, 1.5e-6), ("phase_normalize", float, 1.0),
    ("distance_normalize", float, 1.0), ("distance_normalize_constant", float, 0.0),
)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for name, kind, default in REFERENCE_FLAGS:
        ap.add_argument(f"--{name}", type=kind, default=default)
    extra = ap.add_argument_group("additions of this implementation")
    extra.add_argument("--synthetic-weights", type=int, default=None, metavar="SEED",
                       help="use seeded He-normal weights when the trained files are absent")
    extra.add_argument("--unwrap", choices=["on", "off"], default="on")
    extra.add_argument("--compute-dtype", choices=["f32", "bf16", "f32s"], default="f32",
                       help="conv-stack arithmetic: exact fp32 MFMA (default), bf16, or fp32 values as split fp16 halves")
    extra.add_argument("--data_root", default="./test_data")
    extra.add_argument("--no-images", action="store_true", help="skip the per-sample PNG mosaics")
    return ap.parse_args(argv)


def load_weights(args, device):
    import net
    from style_transfer_based_holographic_imaging_amd import synth
    decoder = copy.deepcopy(net.decoder)
    decoder_ph = copy.deepcopy(net.decoder)
    distance_g = net.Distance_G()
    vgg = copy.deepcopy(net.vgg)
    files = [args.decoder, args.decoder_ph, args.distance_g, args.vgg]
    if all(os.path.isfile(f) for f in files):
        decoder.load_state_dict(torch.load(args.decoder, map_location='cpu'))
        decoder_ph.load_state_dict(torch.load(args.decoder_ph, map_location='cpu'))
        distance_g.load_state_dict(torch.load(args.distance_g, map_location='cpu'))
        state_dict = torch.load(args.vgg, map_location='cpu')
        state_dict['0.weight'] = state_dict['0.weight'].sum(dim=1, keepdim=True)   # RGB pre-conv -> 1 channel
        vgg.load_state_dict(state_dict)
    elif args.synthetic_weights is not None:
        vgg_sd, dec_sd, dist_sd = synth.synthetic_state_dicts(args.synthetic_weights, full_vgg=True)
        vgg.load_state_dict(vgg_sd)
        decoder.load_state_dict(dec_sd)
        decoder_ph.load_state_dict(dec_sd)
        distance_g.load_state_dict(dist_sd)
        print(f"[note] trained weights not found under {args.decoder_root!r} / {args.vgg!r}: "
              f"using synthetic weights, seed {args.synthetic_weights}")
    else:
        missing = [f for f in files if not os.path.isfile(f)]
        raise FileNotFoundError(f"missing weight files {missing}; pass --synthetic-weights SEED to run without them")
    vgg = nn.Sequential(*list(vgg.children())[:31])
    network = net.Net(vgg, decoder, decoder_ph, distance_g)
    network.eval()
    network.to(device)
    return network


def main(argv=None):
    args = parse_args(argv)
    if not args.device.startswith('cuda'):
        raise SystemExit("this implementation runs on MI355X HIP kernels only: use --device cuda "
                         "(the CPU restatement in oracle/ is test infrastructure)")
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device visible")
    from utils.Forward_model import Holo_Generator
    from utils.functions import field_retrieval
    from utils import metrics

    device = torch.device(args.device)
    # make `--device cuda:N` the current device too: torch's own allocations / streams follow the current device
    # (the HIP ops launch on their tensors' device either way)
    torch.cuda.set_device(device if device.index is not None else torch.cuda.current_device())
    args.save_dir = args.save_dir + '/%s' % args.exp_name
    output_dir = Path(args.save_dir)
    output_dir.mkdir(exist_ok=True, parents=True)
    args.decoder = os.path.join(args.decoder_root, 'decoder_iter_80000.pth.tar')
    args.decoder_ph = os.path.join(args.decoder_root, 'decoder_ph_iter_80000.pth.tar')
    args.distance_g = os.path.join(args.decoder_root, 'distance_g_iter_80000.pth.tar')
    args.style_path = os.path.join('./style_representation', 'MNIST', 'style_vector.pt')

    network = load_weights(args, device)
    if args.compute_dtype != "f32":
        network.set_compute_dtype(args.compute_dtype)
    model_forward = Holo_Generator(args).to(device)
    if not os.path.isfile(args.style_path):
        raise FileNotFoundError(args.style_path)
    style_vector = torch.load(args.style_path, map_location='cpu').to(device).float()

    use_unwrap = args.unwrap == 'on'

    psnr_list, mae_list, distance_list = [], [], []
    vis_idx = 0
    for i in range(20):
        def load(kind):
            return torch.load(os.path.join(args.data_root, f"test_{kind}_{i}.pt"), map_location='cpu')
        content_holo, distance_style = load("content_holo"), load("distance_style")
        distance_content, gt_amplitude = load("distance_content"), load("gt_amplitude")
        gt_phase = load("gt_phase").to(device)
        content_images = torch.sqrt(content_holo).to(device).float().detach()
        distance_style = distance_style.to(device).float()
        batch_size = content_images.shape[0]
        with torch.no_grad():
            amp_field, ph_field, distance_content_pred = field_retrieval(network, content_images, style_vector, 1.0, True)
            amp_foc, ph_foc = model_forward(amp_field, ph_field * args.phase_normalize,
                                            -distance_style - 2 * args.distance_normalize_constant,
                                            return_field=True, unwrap=use_unwrap)
            gt_phase -= torch.mean(gt_phase, dim=(-2, -1), keepdim=True).expand_as(gt_phase)
            ph_field -= torch.mean(ph_field, dim=(-2, -1), keepdim=True).expand_as(amp_field)
            ph_foc -= torch.mean(ph_foc, dim=(-2, -1), keepdim=True).expand_as(amp_foc)
            psnr_list.append(metrics.psnr(ph_foc, gt_phase))
            mae_list.append(metrics.mae(ph_foc, gt_phase))
            for j in range(batch_size):
                vis_idx += 1
                distance_list.append([distance_content.squeeze()[j].item(), distance_content_pred[j].squeeze().item()])
                if args.no_images:
                    continue
                top = torch.cat([content_images[j:j + 1].cpu(), amp_field[j:j + 1].cpu(), gt_amplitude[j:j + 1].cpu(),
                                 amp_foc[j:j + 1].cpu()], dim=3)
                bot = torch.cat([torch.zeros_like(content_images[j:j + 1]).cpu(), ph_field[j:j + 1].cpu(),
                                 gt_phase[j:j + 1].cpu(), ph_foc[j:j + 1].cpu()], dim=3)
                metrics.save_image(torch.cat([top, bot], dim=2), str(output_dir / '{:d}_test{:s}'.format(vis_idx, args.save_ext)))

    print("Mean PSNR: ", np.mean(psnr_list))
    print("Mean MAE: ", np.mean(mae_list))
    dl = (np.array(distance_list) + args.distance_normalize_constant) * args.distance_normalize * 1000
    distance_true, distance_pred = dl[:, 0], dl[:, 1]
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        distance = {np.round(n, 2): [] for n in np.unique(distance_true)}
        for n, m in zip(distance_true, distance_pred):
            distance[np.round(n, 2)].append(m)
        plt.figure(figsize=(12, 3))
        plt.boxplot(list(distance.values()), vert=0)
        plt.savefig(output_dir / 'distance_prediction.png')
        plt.close()
    except ImportError:
        print("[note] matplotlib is not installed: distance_prediction.png skipped")
    print("R2 score: ", metrics.r2_score(distance_true, distance_pred))
    return 0


if __name__ == "__main__":
    sys.exit(main())
