"""Every BASELINE config at its FULL batch (GPU): the activation tensors of configs[2] and configs[4] pass 2^32 bytes
(256 x 512^2 x 64 ch x 2 B = 8.6 GB; 32 x 1024^2 x 64 ch x 4 B = 8.6 GB), which is where a 32-bit image base or a slip in a
persistent kernel's tile id would show.  No oracle is needed at these sizes: a hologram's result is bit-exact whatever
batch of the same bucket it arrives in (tests/test_hip_network.py::test_full_size_properties), so hologram k of the full
batch must EQUAL hologram k run inside a batch of 16 — first, middle and last k; every hologram of the batch is distinct,
so an image that lands in the wrong slot cannot pass.  The arithmetic itself is pinned at these sizes on one or two
holograms against the oracle (test_hip_network.py, test_hip_bf16.py, test_hip_propagator.py).

Reference path: test_field_retrieval_mnist.py:125-126; the sweep idiom utils/Data_loader.py:59-62."""
import os

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from hip_helpers import DEV, build_network, holo_generator
    assert torch.cuda.is_available()
    network, sds = build_network(0)
    return dict(dev=DEV, net=network, sds=sds, hg=holo_generator())


def _distinct_holograms(env, batch, size, seed):
    """`batch` DISTINCT network inputs on the device: 16 seeded synthetic holograms (the configs' generator),
    hologram k = base[k % 16] scaled by a factor of its own in (0.7, 1.0]."""
    from style_transfer_based_holographic_imaging_amd import synth
    dev, hg = env["dev"], env["hg"]
    ph = synth.synthetic_phase_objects(16, size, seed=seed).to(dev)
    d_c = synth.synthetic_distances(16, seed=seed).to(dev)
    with torch.no_grad():
        base = torch.sqrt(hg(torch.full_like(ph, 0.6), ph, d_c))
    k = torch.arange(batch, device=dev)
    scale = (1.0 - 0.3 * k.float() / batch).view(-1, 1, 1, 1)
    content = (base[k % 16] * scale).contiguous()
    flat = content.view(batch, -1)
    assert len({float(v) for v in flat[:, flat.shape[1] // 2 + size // 2].cpu()}) == batch      # all distinct
    return content


def _slices(batch, ks, piece=16):
    """16-hologram windows of the batch that contain the holograms `ks` (each window inside the batch)."""
    out = []
    for k in ks:
        lo = min(max(0, k - piece // 2), batch - piece)
        out.append((k, lo, lo + piece))
    return out


def _retrieve(env, content, style):
    from utils.functions import field_retrieval
    with torch.no_grad():
        return field_retrieval(env["net"], content, style, 1.0, True)


def test_config1_fp32_batch64_at_256(env, mnist_style):
    """BASELINE configs[1]: 64 x 256^2 fp32 under `auto` (what bench.py times): holograms 0, 31, 63 of the full batch equal
    the same holograms inside a batch of 16 (bucket 13+: the same kernels), network outputs and focused field."""
    dev, hg = env["dev"], env["hg"]
    style = mnist_style.to(dev)
    content = _distinct_holograms(env, 64, 256, 1234)
    d_s = torch.full((64, 1, 1, 1), -0.2, device=dev)
    amp, ph, dist = _retrieve(env, content, style)
    with torch.no_grad():
        a_f, p_f = hg(amp, ph * 1.0, d_s, return_field=True)
    assert torch.isfinite(amp).all() and torch.isfinite(ph).all() and torch.isfinite(a_f).all()
    for k, lo, hi in _slices(64, (0, 31, 63)):
        a2, p2, d2 = _retrieve(env, content[lo:hi], style)
        j = k - lo
        assert torch.equal(amp[k], a2[j]) and torch.equal(ph[k], p2[j]) and torch.equal(dist[k], d2[j]), k
        with torch.no_grad():
            a1, p1 = hg(amp[k:k + 1], ph[k:k + 1] * 1.0, d_s[:1], return_field=True)
        assert torch.equal(a_f[k:k + 1], a1) and torch.equal(p_f[k:k + 1], p1), k
    # the holograms differ, so must the results (a slot that received another image would repeat one)
    assert not torch.equal(amp[0], amp[16]) and not torch.equal(amp[31], amp[63])


def test_config2_bf16_batch256_at_512(env, mnist_style):
    """BASELINE configs[2]: 256 x 512^2, bf16 convolutions / fp32 propagator.  The 64-channel full-resolution activations are
    8.6 GB (past 2^32 bytes, past 2^31 elements): holograms 0, 127, 255 of the full batch equal the same holograms inside a
    batch of 16, and the fp32 propagator (P = 1024, run in groups) on them equals per-hologram calls."""
    dev, hg, network = env["dev"], env["hg"], env["net"]
    style = mnist_style.to(dev)
    content = _distinct_holograms(env, 256, 512, 1235)
    d_s = torch.full((256, 1, 1, 1), -0.2, device=dev)
    try:
        network.set_compute_dtype(torch.bfloat16)
        amp, ph, dist = _retrieve(env, content, style)
        with torch.no_grad():
            a_f, p_f = hg(amp, ph * 1.0, d_s, return_field=True)
        assert amp.shape == (256, 1, 512, 512) and torch.isfinite(amp).all() and torch.isfinite(ph).all()
        for k, lo, hi in _slices(256, (0, 127, 255)):
            a2, p2, d2 = _retrieve(env, content[lo:hi], style)
            j = k - lo
            assert torch.equal(amp[k], a2[j]) and torch.equal(ph[k], p2[j]) and torch.equal(dist[k], d2[j]), k
            with torch.no_grad():
                a1, p1 = hg(amp[k:k + 1], ph[k:k + 1] * 1.0, d_s[:1], return_field=True)
            assert torch.equal(a_f[k:k + 1], a1) and torch.equal(p_f[k:k + 1], p1), k
        assert not torch.equal(amp[0], amp[16]) and not torch.equal(amp[127], amp[255])
    finally:
        network.set_compute_dtype(torch.float32)


def test_config4_fp32_batch32_at_1024_with_sweep(env):
    """BASELINE configs[4]: 32 frames of 1024^2, red_blood_cell style vector, 8 distances per frame (fp32; the 64-channel
    full-resolution activations are 8.6 GB).  Frames 0 and 31 of the full batch equal the same frames inside a batch of 16
    (network), and their 8-distance sweeps (P = 2048, shared forward transform) equal the sweep of that frame alone."""
    dev, hg = env["dev"], env["hg"]
    style = torch.load(os.path.join(REPO, "style_representation", "red_blood_cell", "style_vector.pt"), map_location="cpu").float().to(dev)
    content = _distinct_holograms(env, 32, 1024, 1237)
    sweep_d = torch.linspace(-0.1, -0.8, 8, device=dev)
    amp, ph, dist = _retrieve(env, content, style)
    with torch.no_grad():
        a_s, p_s = hg.sweep(amp, ph * 1.0, sweep_d)
    assert a_s.shape == (32, 8, 1024, 1024) and torch.isfinite(a_s).all() and torch.isfinite(p_s).all()
    for k, lo, hi in _slices(32, (0, 31)):
        a2, p2, d2 = _retrieve(env, content[lo:hi], style)
        j = k - lo
        assert torch.equal(amp[k], a2[j]) and torch.equal(ph[k], p2[j]) and torch.equal(dist[k], d2[j]), k
        with torch.no_grad():
            a1, p1 = hg.sweep(amp[k:k + 1], ph[k:k + 1] * 1.0, sweep_d)
        assert torch.equal(a_s[k:k + 1], a1) and torch.equal(p_s[k:k + 1], p1), k
    assert not torch.equal(amp[0], amp[16])


def test_config3_total2048_on_one_rank(env, mnist_style):
    """BASELINE configs[3] as one rank sees it: a fixed total of 2048 holograms of 256^2 walked in micro-batches of 64 by
    shard.sharded_pass (what `bench.py --total 2048` times); rows 0, 1023, 2047 of the gathered block equal a direct call on
    a batch of 16 that contains the hologram.  (More than one rank: tests/test_shard_gloo.py, tests/test_hip_rccl.py.)"""
    from style_transfer_based_holographic_imaging_amd import shard
    dev, hg = env["dev"], env["hg"]
    style = mnist_style.to(dev)
    total, s = 2048, 256
    content = _distinct_holograms(env, total, s, 1236)
    d_s = torch.full((64, 1, 1, 1), -0.2, device=dev)

    def one_batch(x):
        amp, ph, dist = _retrieve(env, x, style)
        with torch.no_grad():
            a_f, p_f = hg(amp, ph * 1.0, d_s[:x.shape[0]], return_field=True)
        return amp, ph, a_f, p_f, dist

    def process(m0, m1, rows):
        shard.pack_outputs(*one_batch(content[m0:m1]), out=rows)

    block = shard.sharded_pass(total, 64, process, shard.packed_width(s), dev)
    assert block.shape == (total, shard.packed_width(s)) and torch.isfinite(block).all()
    for k, lo, hi in _slices(total, (0, 1023, 2047)):
        want = shard.pack_outputs(*one_batch(content[lo:hi]))
        assert torch.equal(block[k], want[k - lo]), k
    amp, ph, a_f, p_f, dist = shard.unpack_outputs(block[1023:1024], s)
    assert amp.shape == (1, 1, s, s) and 0.0 < float(dist) < 1.0
