"""CPU: the entry script's host-side post-processing (utils/metrics.py) against closed-form values.

torchmetrics / torchvision / scikit-learn are not in the image, so the values are pinned to the DEFINITIONS the reference
relies on (test_field_retrieval_mnist.py:105-106,128-159): ``PeakSignalNoiseRatio()`` with its default ``data_range=None``
(range = max(target) - min(target) of the batch, base 10, mean over all elements), ``MeanAbsoluteError()``,
``sklearn.metrics.r2_score`` (1 - SS_res / SS_tot), and ``torchvision.utils.save_image`` (x * 255 + 0.5, clamp, uint8, RGB)."""
import struct
import zlib

import numpy as np
import pytest
import torch

from utils import metrics


def test_psnr_uses_the_targets_range_like_torchmetrics_default():
    target = torch.tensor([[0.0, 2.0], [4.0, 1.0]])                 # range 4
    pred = target + torch.tensor([[0.5, -0.5], [0.5, -0.5]])        # MSE 0.25
    assert metrics.psnr(pred, target) == pytest.approx(10.0 * np.log10(16.0 / 0.25), rel=1e-12)
    # scale covariance: PSNR is unchanged when prediction and target are scaled together (the range scales with them)
    assert metrics.psnr(3.0 * pred, 3.0 * target) == pytest.approx(metrics.psnr(pred, target), rel=1e-12)
    # NOT a fixed data_range of 1: a target spanning [0, 0.5] with the same absolute error scores 10*log10(0.25/mse)
    t2 = torch.tensor([0.0, 0.5, 0.25, 0.125])
    p2 = t2 + 0.05
    assert metrics.psnr(p2, t2) == pytest.approx(10.0 * np.log10(0.25 / 0.0025), rel=1e-6)     # fp32 inputs
    # batch semantics: one range and one mean over the whole (B,1,H,W) batch, as the reference calls it per file
    g = torch.Generator().manual_seed(0)
    tb, pb = torch.rand(5, 1, 8, 8, generator=g), torch.rand(5, 1, 8, 8, generator=g)
    want = 10.0 * np.log10(float(tb.max() - tb.min()) ** 2 / float(((pb - tb).double() ** 2).mean()))
    assert metrics.psnr(pb, tb) == pytest.approx(want, rel=1e-6)


def test_mae_and_r2():
    a, b = torch.tensor([1.0, -2.0, 3.0]), torch.tensor([0.0, 0.0, 0.0])
    assert metrics.mae(a, b) == pytest.approx(2.0)
    y = np.array([1.0, 2.0, 3.0, 4.0])
    assert metrics.r2_score(y, y) == 1.0
    assert metrics.r2_score(y, np.full(4, y.mean())) == pytest.approx(0.0)          # predicting the mean scores 0
    assert metrics.r2_score(y, y[::-1]) == pytest.approx(1.0 - 20.0 / 5.0)           # SS_res 20, SS_tot 5
    assert metrics.r2_score([0.4, 0.5, 0.6], [0.41, 0.5, 0.58]) == pytest.approx(1.0 - 0.0005 / 0.02)


def _decode_png(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF, tag
        chunks.append((tag, body))
        pos += 12 + n
    assert [t for t, _ in chunks][0] == b"IHDR" and chunks[-1][0] == b"IEND"
    w, h, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, filt, inter) == (8, 2, 0, 0, 0)                       # 8-bit RGB, no interlace
    data = zlib.decompress(b"".join(b for t, b in chunks if t == b"IDAT"))
    rows = np.frombuffer(data, np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()                                                    # filter type 0 on every scanline
    return rows[:, 1:].reshape(h, w, 3)


def test_mosaic_writer_round_trips(tmp_path):
    """save_image semantics (torchvision.utils.save_image of one image): x*255 + 0.5, clamped to [0, 255], truncated to
    uint8, grey replicated to RGB; decoded back chunk by chunk (CRCs included)."""
    g = torch.Generator().manual_seed(3)
    img = torch.rand(1, 1, 6, 10, generator=g) * 1.4 - 0.2          # values below 0 and above 1 exercise the clamp
    img[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.5, 127.4 / 255.0])
    path = tmp_path / "m.png"
    metrics.save_image(img, str(path))
    rgb = _decode_png(path)
    assert rgb.shape == (6, 10, 3)
    want = (img[0, 0] * 255 + 0.5).clamp(0, 255).to(torch.uint8).numpy()
    assert (rgb[..., 0] == want).all() and (rgb[..., 1] == want).all() and (rgb[..., 2] == want).all()
    assert list(rgb[0, :4, 0]) == [0, 255, 128, 127]
    # the reference's 2 x 4 mosaic of 128 x 128 panels: 512 wide, 256 tall
    panels = torch.rand(1, 1, 256, 512, generator=g)
    metrics.save_image(panels, str(path))
    assert _decode_png(path).shape == (256, 512, 3)
