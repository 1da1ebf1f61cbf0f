"""Seeded synthetic weights and inputs.

The reference ships no trained weights (``/root/reference/.MISSING_LARGE_BLOBS:1-4``), so parity
and timing use He-normal weights drawn from a framework-independent generator
(numpy ``default_rng``): the same seed gives the same tensors in this container, on the GPU box
and inside the golden-vector script, without shipping 50 MB of parameters.

State-dict key surface mirrors the reference modules:
  * ``net.vgg``      conv children at 0,2,5,9,12,16,19,22,25,29 (``net.py:76-107``; 53 children in
    the full net, the first 31 are used, ``test_field_retrieval_mnist.py:89``)
  * ``net.decoder``  children 1,4,6,9,12,15,18,20,23,26,28,31,34,37 (``net.py:33-74``)
  * ``net.Distance_G``  ``l1,l2,l3,out`` (``net.py:266-286``)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.utils.data

# (child index, kind, Cin, Cout); kind: "c1" 1x1 conv, "c3" 3x3 conv, "ct" ConvTranspose2d k2 s2.
VGG_CONVS = [
    (0, "c1", 1, 3), (2, "c3", 3, 64), (5, "c3", 64, 64), (9, "c3", 64, 128), (12, "c3", 128, 128),
    (16, "c3", 128, 256), (19, "c3", 256, 256), (22, "c3", 256, 256), (25, "c3", 256, 256),
    (29, "c3", 256, 512),
    # relu4_2 .. relu5_4: present in net.vgg (53 children) but sliced off by the entry script.
    (32, "c3", 512, 512), (35, "c3", 512, 512), (38, "c3", 512, 512), (42, "c3", 512, 512),
    (45, "c3", 512, 512), (48, "c3", 512, 512), (51, "c3", 512, 512),
]
DECODER_CONVS = [
    (1, "c3", 512, 512), (4, "c3", 512, 256), (6, "ct", 256, 256), (9, "c3", 256, 256),
    (12, "c3", 256, 256), (15, "c3", 256, 256), (18, "c3", 256, 128), (20, "ct", 128, 128),
    (23, "c3", 128, 128), (26, "c3", 128, 64), (28, "ct", 64, 64), (31, "c3", 64, 64),
    (34, "c3", 64, 64), (37, "c3", 64, 2),
]
DISTANCE_LINEARS = [("l1", 1024, 1024), ("l2", 1024, 1024), ("l3", 1024, 512), ("out", 512, 1)]


def _he(rng, shape, fan_in):
    return (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)


def _bias(rng, n):
    return (rng.standard_normal(n) * 0.1).astype(np.float32)


def _conv_state(rng, table):
    sd = {}
    for idx, kind, cin, cout in table:
        if kind == "c1":
            w = _he(rng, (cout, cin, 1, 1), cin)
        elif kind == "c3":
            w = _he(rng, (cout, cin, 3, 3), cin * 9)
        else:  # ConvTranspose2d weight is (Cin, Cout, 2, 2); every output pixel sees Cin terms
            w = _he(rng, (cin, cout, 2, 2), cin)
        sd[f"{idx}.weight"] = torch.from_numpy(w)
        sd[f"{idx}.bias"] = torch.from_numpy(_bias(rng, cout))
    return sd


def synthetic_state_dicts(seed: int = 0, full_vgg: bool = False):
    """Return ``(vgg_sd, decoder_sd, distance_sd)`` of float32 CPU tensors.

    ``full_vgg=False`` gives the 20 tensors of ``vgg[:31]``; ``True`` all 34 of the 53-layer net.
    The stream is drawn layer by layer so the first 31 layers are identical in both modes.
    """
    rng = np.random.default_rng(seed)
    vgg_all = _conv_state(rng, VGG_CONVS)
    if not full_vgg:
        keep = {f"{i}." for i, *_ in VGG_CONVS if i < 31}
        vgg_all = {k: v for k, v in vgg_all.items() if k[: k.index(".") + 1] in keep}
    dec = _conv_state(rng, DECODER_CONVS)
    dist = {}
    for name, fin, fout in DISTANCE_LINEARS:
        dist[f"{name}.weight"] = torch.from_numpy(_he(rng, (fout, fin), fin))
        dist[f"{name}.bias"] = torch.from_numpy(_bias(rng, fout))
    return vgg_all, dec, dist


def synthetic_style_vector(seed: int = 0):
    """A ``(2,512,1,1)`` [mean; std] style vector with positive std, like the bundled ones."""
    rng = np.random.default_rng(seed + 7919)
    mu = np.abs(rng.standard_normal((1, 512, 1, 1))).astype(np.float32) * 0.4
    sd = (0.2 + np.abs(rng.standard_normal((1, 512, 1, 1))) * 0.4).astype(np.float32)
    return torch.from_numpy(np.concatenate([mu, sd], 0))


def synthetic_phase_objects(batch: int, size: int, seed: int = 1234):
    """Digit-like phase objects in [0,1]: smooth random blobs of side ``size-64`` zero-padded by
    32 px per side, the recipe of ``utils/Data_loader.py:54`` with MNIST replaced by seeded blobs
    (SURVEY.md §8d). Sizes below 96 use a proportional border instead of 32."""
    rng = np.random.default_rng(seed)
    border = 32 if size >= 96 else size // 4
    inner = size - 2 * border
    out = np.zeros((batch, 1, size, size), np.float32)
    yy, xx = np.mgrid[0:inner, 0:inner].astype(np.float32)
    for b in range(batch):
        img = np.zeros((inner, inner), np.float32)
        for _ in range(int(rng.integers(3, 7))):
            cy, cx = rng.uniform(0.2, 0.8, 2) * inner
            sy, sx = rng.uniform(0.04, 0.16, 2) * inner
            img += rng.uniform(0.5, 1.0) * np.exp(-((yy - cy) ** 2 / (2 * sy**2) + (xx - cx) ** 2 / (2 * sx**2)))
        img = np.clip(img, 0.0, None)
        img /= max(float(img.max()), 1e-6)
        img[img < 0.05] = 0.0
        out[b, 0, border:border + inner, border:border + inner] = img
    return torch.from_numpy(out)


def synthetic_distances(batch: int, seed: int = 1234):
    """Content distances in mm drawn from the bundled set {0.4..0.8} (``test_data``), (B,1,1,1)."""
    rng = np.random.default_rng(seed + 1)
    d = rng.choice(np.array([0.4, 0.5, 0.6, 0.7, 0.8], np.float32), size=batch)
    return torch.from_numpy(d.astype(np.float32)).view(-1, 1, 1, 1)


def job_inputs(lo: int, hi: int, size: int, seed: int = 1236, chunk: int = 64):
    """Phase objects and content distances of holograms [lo, hi) of a fixed-total job (BASELINE configs[3]): hologram i is
    a function of i ALONE — chunk i // chunk is drawn from seed + chunk index — so any sharding of the job over ranks gives
    the same holograms.  Returns ((hi-lo,1,S,S) phase objects, (hi-lo,1,1,1) distances in mm)."""
    phs, ds = [], []
    if hi > lo:
        for c in range(lo // chunk, (hi - 1) // chunk + 1):
            c_lo, c_hi = max(lo, chunk * c) - chunk * c, min(hi, chunk * c + chunk) - chunk * c
            phs.append(synthetic_phase_objects(chunk, size, seed=seed + c)[c_lo:c_hi])
            ds.append(synthetic_distances(chunk, seed=seed + c)[c_lo:c_hi])
    if not phs:
        return torch.zeros((0, 1, size, size)), torch.zeros((0, 1, 1, 1))
    return torch.cat(phs, 0), torch.cat(ds, 0)


class SyntheticDigits(torch.utils.data.Dataset):
    """Stand-in for ``torchvision.datasets.MNIST`` (absent offline): ``n`` seeded 28 x 28 digit-like blobs in [0, 1] with a
    label, indexable like the dataset ``utils.Data_loader.mnist_loader(_test)`` draws its first shuffled batch from."""

    def __init__(self, n: int = 256, side: int = 28, seed: int = 4321):
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:side, 0:side].astype(np.float32)
        imgs = np.zeros((n, 1, side, side), np.float32)
        for i in range(n):
            img = np.zeros((side, side), np.float32)
            for _ in range(int(rng.integers(2, 5))):
                cy, cx = rng.uniform(0.25, 0.75, 2) * side
                sy, sx = rng.uniform(0.06, 0.2, 2) * side
                img += rng.uniform(0.5, 1.0) * np.exp(-((yy - cy) ** 2 / (2 * sy**2) + (xx - cx) ** 2 / (2 * sx**2)))
            img /= max(float(img.max()), 1e-6)
            img[img < 0.1] = 0.0
            imgs[i, 0] = img
        self.images = torch.from_numpy(imgs)
        self.labels = torch.from_numpy(rng.integers(0, 10, n).astype(np.int64))

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        return self.images[i], int(self.labels[i])
