"""Drop-in for the synthesis half of ``utils/Data_loader.py``: ``mnist_loader`` (Data_loader.py:10-36) and
``mnist_loader_test`` (:38-69) — batches of phase objects turned into style / content holograms by forward propagation
(SURVEY.md 8f N1).  Same signatures, same random draws in the same order; the holograms come from the fused HIP propagator
(``model_forward`` = ``utils.Forward_model.Holo_Generator``) and everything stays on the device.

torchvision is not in the image, so the reference's ``transforms.Compose([RandomHorizontalFlip(), RandomVerticalFlip(),
RandomAffine(degrees=0, translate=[0.1, 0.1], fill=0.0)])`` is restated with torch ops (``random_flip_translate``): one draw
``torch.rand(1) < 0.5`` per flip, then RandomAffine.get_params' draws — the angle (a degenerate uniform_(-0, 0) that still
advances the generator), ``tx = int(round(uniform(-0.1 W, 0.1 W)))``, ``ty`` likewise — applied to the WHOLE batch tensor
(torchvision transforms treat a (B,1,H,W) tensor as one image); a zero-angle integer translation under nearest-neighbour
sampling is an exact shift, ``out[y, x] = in[y - ty, x - tx]``, with fill 0.  **Parity unpinned** against torchvision itself
(absent, un-vendored, unpinned in the reference); tests hold the restatement to its definition.  ``Holo_loader`` (measured
.mat files of the tissue / bead / blood-cell sets) is not on this path and is not provided."""
import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.data as data


def shift_zero_fill(x, tx, ty):
    """out[..., y, x] = in[..., y - ty, x - tx], zero outside: a positive tx moves the content to the right, ty down."""
    h, w = x.shape[-2:]
    out = torch.zeros_like(x)
    if abs(tx) >= w or abs(ty) >= h:
        return out
    ys, yd = (slice(0, h - ty), slice(ty, h)) if ty >= 0 else (slice(-ty, h), slice(0, h + ty))
    xs, xd = (slice(0, w - tx), slice(tx, w)) if tx >= 0 else (slice(-tx, w), slice(0, w + tx))
    out[..., yd, xd] = x[..., ys, xs]
    return out


def random_flip_translate(x, translate=(0.1, 0.1), p=0.5):
    """The reference's transform pipeline on a batch tensor, consuming torch's global generator like torchvision does."""
    if torch.rand(1) < p:                      # RandomHorizontalFlip
        x = x.flip(-1)
    if torch.rand(1) < p:                      # RandomVerticalFlip
        x = x.flip(-2)
    h, w = x.shape[-2:]
    torch.empty(1).uniform_(-0.0, 0.0)         # RandomAffine.get_params: angle (degrees = 0)
    max_dx, max_dy = float(translate[0] * w), float(translate[1] * h)
    tx = int(round(torch.empty(1).uniform_(-max_dx, max_dx).item()))
    ty = int(round(torch.empty(1).uniform_(-max_dy, max_dy).item()))
    return shift_zero_fill(x, tx, ty)


def _first_batch(dataset, batch_size, device):
    return next(iter(data.DataLoader(dataset, batch_size=batch_size, shuffle=True)))[0].to(device)


def _draw_distances(holo_list, n, device):
    picks = [holo_list[np.random.randint(low=0, high=len(holo_list), size=1)[0]] for _ in range(n)]
    return torch.Tensor(picks).reshape([-1, 1, 1, 1]).to(device).float()


def mnist_loader(args, dataset, train_holo_list_style, train_holo_list_content, model_forward, device, return_gt=False):
    """Data_loader.py:10-36: independent style / content image batches, one transform draw each."""
    style_images = _first_batch(dataset, args.batch_size, device)
    content_images = _first_batch(dataset, args.batch_size, device)
    distance_style = _draw_distances(train_holo_list_style, args.batch_size, device)
    distance_content = _draw_distances(train_holo_list_content, args.batch_size, device)
    distance_style = -args.distance_normalize_constant + distance_style / args.distance_normalize
    distance_content = -args.distance_normalize_constant + distance_content / args.distance_normalize
    phase_style = F.pad(style_images, (32, 32, 32, 32), 'constant', 0).to(device).float()
    amplitude = torch.ones_like(phase_style) * 0.6
    phase_style = random_flip_translate(phase_style)
    phase_content = F.pad(content_images, (32, 32, 32, 32), 'constant', 0).to(device).float()
    phase_content = random_flip_translate(phase_content)
    style_holo = model_forward(amplitude, phase_style, distance_style).to(device).float().detach()
    content_holo = model_forward(amplitude, phase_content, distance_content).to(device).float().detach()
    if return_gt:
        return style_holo, content_holo, distance_style, distance_content, amplitude, phase_content
    return style_holo, content_holo, distance_style, distance_content


def mnist_loader_test(args, dataset, holo_list_style, holo_list_content, model_forward, device, test_interpolation=False):
    """Data_loader.py:38-69: ONE image batch, transformed twice (content first, then style); with ``test_interpolation``
    every listed style distance is applied to the batch repeated along dim 0 (the sweep idiom, :59-62)."""
    style_images = _first_batch(dataset, args.batch_size, device)
    if test_interpolation:
        distance_style = torch.Tensor(holo_list_style).reshape([-1, 1, 1, 1]).to(device).float()
    else:
        distance_style = _draw_distances(holo_list_style, args.batch_size, device)
    distance_content = _draw_distances(holo_list_content, args.batch_size, device)
    distance_style = -args.distance_normalize_constant + distance_style / args.distance_normalize
    distance_content = -args.distance_normalize_constant + distance_content / args.distance_normalize
    phase_style = F.pad(style_images, (32, 32, 32, 32), 'constant', 0).to(device).float()
    amplitude = torch.ones_like(phase_style) * 0.6
    phase_content = random_flip_translate(phase_style)
    phase_style = random_flip_translate(phase_style)
    if test_interpolation:
        b = distance_style.shape[0]
        style_holo = model_forward(amplitude, phase_style.repeat(b, 1, 1, 1), distance_style).to(device).float().detach()
        content_holo = model_forward(amplitude, phase_content.repeat(b, 1, 1, 1), distance_content.repeat(b, 1, 1, 1)).to(device).float().detach()
        return style_holo, content_holo, distance_style, distance_content, amplitude, phase_content
    style_holo = model_forward(amplitude, phase_style, distance_style).to(device).float().detach()
    content_holo = model_forward(amplitude, phase_content, distance_content).to(device).float().detach()
    return style_holo, content_holo, distance_style, distance_content
