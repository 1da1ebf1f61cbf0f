"""Host-side post-processing of the entry script (test_field_retrieval_mnist.py:105-106,128-159):
PSNR / MAE as torchmetrics defines them, R^2 as scikit-learn does, and the 8-bit mosaic writer that
stands in for torchvision.utils.save_image.  Plain numpy/zlib; nothing here is on the GPU hot path."""
import struct
import zlib

import numpy as np
import torch


def psnr(pred, target):
    """torchmetrics PeakSignalNoiseRatio() defaults: data_range = max(target) - min(target), base 10,
    mean over all elements of the batch."""
    pred, target = pred.detach().double().cpu(), target.detach().double().cpu()
    data_range = float(target.max() - target.min())
    mse = float(((pred - target) ** 2).mean())
    return 10.0 * np.log10(data_range ** 2 / mse)


def mae(pred, target):
    return float((pred.detach().double().cpu() - target.detach().double().cpu()).abs().mean())


def r2_score(y_true, y_pred):
    y_true, y_pred = np.asarray(y_true, np.float64), np.asarray(y_pred, np.float64)
    ss_res = ((y_true - y_pred) ** 2).sum()
    ss_tot = ((y_true - y_true.mean()) ** 2).sum()
    return 1.0 - ss_res / ss_tot


def save_image(tensor, path):
    """save_image of ONE (1,1,H,W) tensor: grey replicated to RGB, x*255+0.5 clamped to uint8, PNG."""
    img = tensor.detach().float().cpu()
    while img.dim() > 2:
        img = img[0]
    u8 = img.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).numpy()
    rgb = np.repeat(u8[:, :, None], 3, axis=2)
    h, w = u8.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) \
        + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)
