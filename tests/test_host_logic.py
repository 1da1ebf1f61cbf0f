"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/holo_hip.h
declares, argument validation rejects bad calls before any launch, the plan compiler fuses the
reference's layer lists as documented, state-dict key compatibility, and loud failures on CPU tensors."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn as nn

from conftest import REPO

HEADER = os.path.join(REPO, "include", "holo_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(holo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from style_transfer_based_holographic_imaging_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/holo_hip.h but not exported"
    assert set(declared) == set(_lib.SIGNATURES), "ctypes SIGNATURES and the header disagree"
    assert _lib.lib.holo_abi_version() == 5
    assert _lib.lib.holo_target_arch() == b"gfx950"


def test_argument_validation_needs_no_gpu():
    """Every entry point validates before launching: these return error codes on a GPU-less host."""
    from style_transfer_based_holographic_imaging_amd._lib import lib
    assert lib.holo_conv3x3_f32(None, None, None, None, 1, 4, 4, 16, 64, 1, 0, None) == 1
    assert b"null pointer" in lib.holo_last_error()
    fake = ctypes.c_void_p(16)
    assert lib.holo_conv3x3_f32(fake, fake, fake, fake, 1, 1, 4, 16, 64, 1, 0, None) == 1       # H < 2
    assert b"reflection padding" in lib.holo_last_error()
    assert lib.holo_conv3x3_f32(fake, fake, fake, fake, 1, 4, 4, 24, 64, 1, 0, None) == 2       # Cin % 16
    assert lib.holo_conv3x3_f32(fake, fake, fake, fake, 1, 4, 4, 16, 64, 0, 1, None) == 2       # pool without relu
    assert lib.holo_asm_propagate(fake, fake, 0, 1, 1, fake, fake, 0, 1, 1, fake, 1, 1, 4000, 1, 5e-7, 1e-6, 1.0,
                                  fake, 1 << 40, None) == 2                                      # P = 8000
    assert b"outside [2,4096]" in lib.holo_last_error()
    assert lib.holo_asm_propagate(fake, fake, 0, 1, 1, fake, fake, 0, 1, 1, fake, 1, 1, 41, 1, 5e-7, 1e-6, 1.0,
                                  fake, 1 << 30, None) == 2                                      # odd S, padded
    assert b"even field side" in lib.holo_last_error()
    assert lib.holo_asm_propagate(fake, fake, 0, 1, 1, fake, fake, 0, 1, 1, fake, 1, 1, 64, 1, 5e-7, 1e-6, 1.0,
                                  fake, 16, None) == 3                                           # workspace
    assert lib.holo_asm_propagate(fake, None, 0, 1, 1, fake, fake, 0, 1, 1, fake, 1, 1, 64, 1, 5e-7, 1e-6, 1.0,
                                  fake, 1 << 30, None) == 1                                      # missing phase
    assert lib.holo_adain_apply_f32(fake, fake, fake, fake, fake, 0, 1.5, fake, 1, 4, 64, None) == 1  # alpha
    # tables of the generic path + T (2 fields) + U (6 output fields) + one 65 x 65 transfer-function table per output field
    assert lib.holo_asm_workspace_bytes(2, 3, 64, 1) == 3 * 4096 * 8 + 64 * 128 * 8 * (2 + 6) + 6 * 65 * 65 * 8
    assert lib.holo_asm_workspace_bytes(200, 1, 64, 1) == 3 * 4096 * 8 + 64 * 128 * 8 * 400 + 64 * 65 * 65 * 8   # at most 64 table slots
    assert lib.holo_asm_workspace_bytes(1, 1, 1024, 1) == 3 * 4096 * 8 + 1024 * 2048 * 8 * 2                     # P = 2048: row-major path, no table
    assert lib.holo_conv3x3_packed_elems(128, 64) == 9 * 64 * 128
    assert lib.holo_unwrap2d_workspace_bytes(3, 10, 20) == 3 * 200 * 24
    assert lib.holo_unwrap2d_f32(fake, 100, fake, 100, 1, 10, 10, None, 0, None) == 1             # no workspace
    assert lib.holo_unwrap2d_f32(fake, 50, fake, 100, 1, 10, 10, fake, 1 << 20, None) == 1        # stride < image
    assert b"shorter than an image" in lib.holo_last_error()
    assert lib.holo_unwrap2d_f32(fake, 1 << 26, fake, 1 << 26, 1, 8192, 4096, fake, 1 << 40, None) == 2
    assert lib.holo_unwrap2d_f32(fake, 100, fake, 100, 2, 10, 10, fake, 100, None) == 3           # workspace


def test_state_dict_surface_matches_reference_keys():
    import net
    from style_transfer_based_holographic_imaging_amd import synth
    vgg_sd, dec_sd, dist_sd = synth.synthetic_state_dicts(1, full_vgg=True)
    assert len(list(net.vgg.children())) == 53 and len(list(net.decoder.children())) == 38
    assert set(net.vgg.state_dict()) == set(vgg_sd) and len(vgg_sd) == 34
    assert set(net.decoder.state_dict()) == set(dec_sd) and len(dec_sd) == 28
    assert set(net.Distance_G().state_dict()) == set(dist_sd) == {f"{n}.{p}" for n in ("l1", "l2", "l3", "out")
                                                                    for p in ("weight", "bias")}
    for k, v in net.decoder.state_dict().items():
        assert v.shape == dec_sd[k].shape, k
    # weight surgery of the entry script (test_field_retrieval_mnist.py:87) keeps working
    assert net.vgg.state_dict()["0.weight"].shape == (3, 1, 1, 1)
    # ConvTranspose2d weights are (Cin, Cout, 2, 2)
    assert net.decoder.state_dict()["6.weight"].shape == (256, 256, 2, 2)


def test_plan_compiler_fuses_reference_layer_lists():
    import net
    from style_transfer_based_holographic_imaging_amd.engine import compile_plan
    enc = nn.Sequential(*list(net.vgg.children())[:31])
    network = net.Net(enc, net.decoder, None, net.Distance_G())
    assert network.n_layer == 4
    kinds = [[(s.kind, s.conv.out_channels, s.relu, s.pool) for s in getattr(network, f"enc_{k}")._plan().steps]
             for k in range(1, 5)]
    assert kinds[0] == [("first", 64, True, False)]
    assert kinds[1] == [("conv", 64, True, True), ("conv", 128, True, False)]
    assert kinds[2] == [("conv", 128, True, True), ("conv", 256, True, False)]
    assert kinds[3] == [("conv", 256, True, False)] * 2 + [("conv", 256, True, True), ("conv", 512, True, False)]
    dec = [(s.kind, s.conv.out_channels) for s in network.decoder._plan().steps]
    assert dec == [("conv", 512), ("conv", 256), ("convt", 256), ("conv", 256), ("conv", 256), ("conv", 256),
                   ("conv", 128), ("convt", 128), ("conv", 128), ("conv", 64), ("convt", 64), ("conv", 64),
                   ("conv", 64), ("last", 2)]
    # the full 53-layer encoder gives a fifth stage (net.py:152-153)
    full = net.Net(net.vgg, net.decoder)
    assert full.n_layer == 5 and len(full.enc_5._plan().steps) == 4
    for p in network.enc_1.parameters():
        assert p.requires_grad is False          # net.py:167-170
    with pytest.raises(NotImplementedError):
        compile_plan([nn.Conv2d(3, 8, 5)])
    with pytest.raises(NotImplementedError):
        compile_plan([nn.ReflectionPad2d(1), nn.Conv2d(24, 64, 3), nn.ReLU()])


def test_cpu_tensors_fail_loudly():
    import function
    import net
    from utils.Angular_Spectrum_Method import ASM
    from utils.Forward_model import Holo_Generator
    from utils.functions import field_retrieval
    import types
    enc = nn.Sequential(*list(net.vgg.children())[:31])
    network = net.Net(enc, net.decoder, None, net.Distance_G()).eval()
    x = torch.rand(1, 1, 16, 16)
    sv = torch.rand(2, 512, 1, 1)
    with pytest.raises(RuntimeError, match="ROCm device|HIP kernels only"):
        field_retrieval(network, x, sv, 1.0, True)
    with pytest.raises(RuntimeError, match="ROCm device"):
        function.calc_mean_std(torch.rand(1, 64, 4, 4))
    with pytest.raises(AssertionError):
        function.calc_mean_std(torch.rand(64, 4, 4))
    hp = types.SimpleNamespace(wavelength=532e-9, pixel_size=1.5e-6, distance_normalize=1.0,
                               distance_normalize_constant=0.0, phase_normalize=1.0)
    with pytest.raises(RuntimeError, match="ROCm device"):
        Holo_Generator(hp)(torch.ones(1, 1, 16, 16), torch.zeros(1, 1, 16, 16), 0.3)
    with pytest.raises(RuntimeError, match="square"):
        ASM(torch.zeros(1, 1, 8, 16, dtype=torch.complex64), 5e-7, 1e-4, 1e-6)
    with pytest.raises(AssertionError):
        field_retrieval(network, x, sv, 1.2)
    with pytest.raises(RuntimeError, match="ROCm device|HIP kernels only"):
        network(x, x)                              # Net.forward (loss values): the same HIP-only passes
    with pytest.raises(AssertionError):
        network(x, x, 1.5)


def test_phase_constant_rounding_chain():
    """c = fl32(fl32(2 pi) * d) for a float32 distance tensor, float64 2*pi*d for a Python float
    (utils/Angular_Spectrum_Method.py:29)."""
    import math
    import numpy as np
    from style_transfer_based_holographic_imaging_amd import ops
    d = torch.tensor([4e-4, -2e-4, 8e-4], dtype=torch.float32)
    c = ops.phase_constant(d, 3, torch.device("cpu"))
    expect = (np.float32(2 * math.pi) * d.numpy().astype(np.float32)).astype(np.float64)
    assert c.dtype == torch.float64 and np.array_equal(c.numpy(), expect)
    ref = (1j * 2 * math.pi * d).imag.double()        # the reference's own expression
    assert torch.equal(c, ref)
    c2 = ops.phase_constant(3.0e-4, 2, torch.device("cpu"))
    assert c2.tolist() == [2 * math.pi * 3.0e-4] * 2
    with pytest.raises(AssertionError):
        ops.phase_constant(d, 4, torch.device("cpu"))


def test_synthetic_generators_are_deterministic():
    from style_transfer_based_holographic_imaging_amd import synth
    a = synth.synthetic_state_dicts(3)
    b = synth.synthetic_state_dicts(3)
    assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and len(a[0]) == 20
    full = synth.synthetic_state_dicts(3, full_vgg=True)[0]
    assert all(torch.equal(full[k], a[0][k]) for k in a[0])
    p = synth.synthetic_phase_objects(2, 64, seed=5)
    assert p.shape == (2, 1, 64, 64) and float(p.min()) == 0.0 and float(p.max()) <= 1.0
    assert torch.equal(p, synth.synthetic_phase_objects(2, 64, seed=5))
    assert (p[:, :, :16] == 0).all() and (p[:, :, :, -16:] == 0).all()   # zero border


def test_unwrap_has_no_host_fallback():
    """Device path only: ops.unwrap2d refuses host tensors; the drop-in wrapper takes the reference's own
    scikit-image route for them and says so when that library is absent."""
    from style_transfer_based_holographic_imaging_amd import ops
    from utils.functions import unwrap
    with pytest.raises(RuntimeError, match="ROCm device"):
        ops.unwrap2d(torch.zeros(1, 1, 8, 8))
    try:
        import skimage.restoration  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="scikit-image"):
            unwrap(torch.zeros(1, 1, 8, 8))


# ---- round-2 advisor items ----------------------------------------------------------------------------------------
def test_library_carries_the_hash_of_its_sources_and_a_stale_one_is_refused(monkeypatch):
    """The hash of sources + flags + ROCm version is compiled into the library; build.py reads it from the file and
    _lib compares it on every load (every rank): a library built from other sources does not load silently."""
    from style_transfer_based_holographic_imaging_amd import _lib, build
    here = build.source_hash()
    assert build.built_hash() == here == _lib.lib.holo_source_hash().decode()
    assert build.source_hash(("-DHOLO_DEBUG",)) != here            # flags are part of the hash
    assert not build.needs_build()
    monkeypatch.setattr(build, "source_hash", lambda *a: "0" * 64)
    monkeypatch.delenv("HOLO_HIP_LIB", raising=False)
    with pytest.raises(_lib.HoloLibraryError, match="built from other sources"):
        _lib._load()


def test_debug_tile_hook_validates_its_argument():
    from style_transfer_based_holographic_imaging_amd._lib import lib
    assert lib.holo_debug_set_conv_tile(17) == 1 and b"outside [0,9]" in lib.holo_last_error()
    assert lib.holo_debug_set_conv_tile(0) == 0


def test_ops_reject_tensors_on_different_devices_and_follow_the_tensors_device():
    """ops._req: every tensor of a call must sit on ONE ROCm device, and that device (not torch's current one) picks
    the stream and is made current around the launch."""
    from unittest import mock
    from style_transfer_based_holographic_imaging_amd import ops

    def fake(idx):
        t = mock.MagicMock(spec=torch.Tensor)
        t.is_cuda, t.device = True, torch.device("cuda", idx)
        return t
    ops._req(fake(1), fake(1))
    assert ops._LAUNCH.dev == torch.device("cuda", 1)
    with pytest.raises(RuntimeError, match="one device"):
        ops._req(fake(0), fake(1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops._req(fake(0), torch.zeros(1))
    seen = []

    def current_stream(d=None):
        seen.append(d)
        return mock.MagicMock(cuda_stream=0)
    with mock.patch.object(torch.cuda, "current_stream", current_stream):
        ops._req(fake(1))
        ops._stream()
    assert seen == [torch.device("cuda", 1)]
    # a launch for a tensor on cuda:1 while cuda:0 is current switches the device around the call
    entered = []

    class Ctx:
        def __init__(self, d):
            entered.append(d)

        def __enter__(self):
            cur[0] = 1

        def __exit__(self, *a):
            cur[0] = 0
    cur = [0]
    with mock.patch.object(torch.cuda, "current_device", lambda: cur[0]), mock.patch.object(torch.cuda, "device", Ctx):
        ops._req(fake(1))
        ran = []
        ops._launch("x", lambda: ran.append(cur[0]) or 0)
    assert entered == [torch.device("cuda", 1)] and ran == [1]


def test_packed_weight_cache_sees_the_pre_conv_and_can_be_invalidated():
    """engine.Plan: the 'first' step's cache key covers vgg[0] (the 1x1 pre-conv) as well as the 3x3 conv; writes
    through .data need invalidate()."""
    import net
    from style_transfer_based_holographic_imaging_amd.engine import HoloSequential, Plan
    import copy
    children = [copy.deepcopy(m) for m in list(net.vgg.children())[:4]]
    plan = Plan(children)
    st = plan.steps[0]
    assert st.kind == "first" and st.conv0 is children[0]
    a = plan._weights(0, st)
    assert plan._weights(0, st) is a                       # cached
    with torch.no_grad():
        children[0].weight.add_(1.0)                       # in-place op on the PRE-conv: version counter moves
    b = plan._weights(0, st)
    assert b is not a and torch.equal(b[0], children[0].weight)
    children[0].weight.data.mul_(2.0)                      # write through .data: no version bump -> stale until invalidated
    assert plan._weights(0, st) is b
    plan.invalidate()
    c = plan._weights(0, st)
    assert c is not b and torch.equal(c[0], children[0].weight)
    # load_state_dict / _apply on the module drop the cache too
    seq = HoloSequential(*children)
    seq._plan()._weights(0, seq._plan().steps[0])
    assert seq._plan()._packed
    seq.load_state_dict(seq.state_dict())
    assert not seq._plan()._packed
    seq._plan()._weights(0, seq._plan().steps[0])
    seq.float()
    assert not seq._plan()._packed


def test_auto_conv_algo_does_not_follow_the_exact_batch():
    """engine.Plan: "auto" chooses Winograd / direct and the split from per-image quantities and the batch the launch is PLANNED for
    (the top of its bucket: 1 | 2 | 3-5 | 6-8 | 9-12 | 13 and more), so that a hologram's arithmetic does not depend on the shard or
    micro-batch it arrives in (ragged shards of a multi-GPU job, --batch)."""
    import net
    from style_transfer_based_holographic_imaging_amd.engine import Plan
    assert [Plan.batch_bucket(b) for b in (1, 2, 3, 4, 5, 6, 8, 9, 12, 13, 16, 17, 255, 256, 2048)] == \
        [1, 2, 5, 5, 5, 8, 8, 12, 12, 16, 16, 16, 16, 16, 16]
    for seq in (net.decoder, net.vgg):
        plan = Plan(list(seq.children())[:31] if seq is net.vgg else list(seq.children()))
        for st in (s for s in plan.steps if s.kind == "conv"):
            for side in (16, 32, 64, 128, 256):
                def pick(b):
                    return plan._wino_choice(st, torch.empty(b, side, side, st.conv.in_channels, device="meta"))
                assert len({pick(b) for b in (13, 16, 17, 63, 64, 255, 256)}) == 1
                assert pick(64)[1] == 1                                   # the open-ended bucket is never split
                for bucket in ((3, 4, 5), (6, 7, 8), (9, 10, 11, 12)):
                    assert len({pick(b) for b in bucket}) == 1


def test_auto_splits_k_for_layers_short_of_blocks():
    """engine.Plan: a layer with fewer work items than CUs deals its channel chunks to k <= 8 items per tile (parts of at least 2
    chunks); k comes from the cost model (rounds of the persistent grid x the longest item + the reduction pass), i.e. about the k
    that fills the 256 CUs once.  Only a layer that stays below half a wave of items even so takes the direct kernel."""
    import net
    from style_transfer_based_holographic_imaging_amd.engine import Plan
    plan = Plan(list(net.decoder.children()))
    convs = [s for s in plan.steps if s.kind == "conv"]

    def pick(cin, cout, b, side):
        st = next(s for s in convs if (s.conv.in_channels, s.conv.out_channels) == (cin, cout))
        return plan._wino_choice(st, torch.empty(b, side, side, cin, device="meta"))
    assert pick(512, 512, 5, 16) == (True, 3)        # the bundled shape (5 x 128^2): 5 tiles x 16 channel blocks = 80 items -> 240
    assert pick(512, 256, 5, 16) == (True, 6)        # 40 items -> 240, parts of 6 or 5 chunks
    assert pick(256, 256, 5, 32) == (True, 3)        # 16 chunks as 6 + 5 + 5
    assert pick(256, 128, 5, 32) == (True, 6)
    assert pick(128, 128, 5, 64) == (True, 1)        # 160 items of 8 chunks: a second round would cost more than it saves
    assert pick(256, 256, 64, 64) == (True, 1)       # the benchmark's batch: never split
    assert pick(512, 512, 1, 32) == (True, 8)        # one 256^2 hologram: 32 items x 8 = one full round
    assert pick(256, 256, 1, 16) == (False, 1)       # 8 items: even 8 parts stay below half a wave -> direct kernel
    assert pick(512, 512, 3, 16) == pick(512, 512, 4, 16) == pick(512, 512, 5, 16)      # one bucket, one choice
    assert Plan._ksplit(80, 16, 5.2e6) == 3 and Plan._ksplit(512, 16, 1e6) == 1 and Plan._ksplit(64, 4, 1e6) in (1, 2)
    plan.conv_algo = "winograd"
    assert pick(512, 512, 5, 16) == (True, 1)        # forced Winograd is the un-split kernel


def test_backward_entry_points_validate_before_any_launch():
    """ABI 5 (csrc/backward.hip): argument and workspace checks of the training-path entry points run on the host"""
    from style_transfer_based_holographic_imaging_amd._lib import lib
    fake = ctypes.c_void_p(4096)
    assert lib.holo_pix_gemm_workspace_bytes(2, 8, 32, 64, 64, 9) > 0 and lib.holo_pix_gemm_workspace_bytes(2, 8, 32, 64, 64, 3) == 0
    assert lib.holo_pix_gemm_f32(fake, 2, 0, 0, fake, fake, 2, 8, 32, 64, 48, 9, fake, 1 << 30, None) == 1          # channels % 32
    assert b"channels % 32" in lib.holo_last_error()
    assert lib.holo_pix_gemm_f32(fake, 2, 8, 0, fake, fake, 2, 8, 32, 64, 64, 9, fake, 1 << 30, None) == 1          # pitch < h + border
    assert b"p_pitch_rows" in lib.holo_last_error()
    assert lib.holo_pix_gemm_f32(fake, 2, 0, 0, fake, fake, 2, 8, 32, 64, 64, 9, fake, 16, None) == 3               # workspace
    assert lib.holo_relu_pool_bwd_pad_f32(fake, fake, None, fake, 2, 8, 8, 64, 1, 2, 13, 0, None, None, 0, None) == 1   # pitch > h + 2*border
    assert lib.holo_relu_pool_bwd_pad_f32(fake, fake, None, fake, 2, 8, 8, 64, 1, 0, 0, 2, None, None, 0, None) == 1    # side by side needs a border
    assert lib.holo_relu_pool_bwd_pad_f32(fake, fake, None, fake, 2, 8, 8, 64, 1, 2, 10, 0, fake, None, 0, None) == 3   # bias sums need a workspace
    assert lib.holo_relu_pool_bwd_pad_workspace_bytes(64) > 0
    assert lib.holo_fold_reflect_f32(fake, fake, 1, 1, 8, 64, 1, 0, 0, None) == 1                                   # reflection needs h >= 2
    assert lib.holo_conv1x1_f32(fake, fake, fake, fake, 1, 4, 4, 24, 64, 0, None) == 2                              # Cin % 16
    assert lib.holo_channel_mean_std_wide_workspace_bytes(2, 5000, 64) == 2 * 5 * 2 * 64 * 8
    assert lib.holo_channel_mean_std_wide_f32(fake, fake, fake, 2, 5000, 64, 1e-5, fake, 8, None) == 3
    assert lib.holo_mask_scale_f32(fake, None, 2.0, fake, 10, None) == 1
    assert lib.holo_adain_fused_f32(fake, fake, fake, fake, fake, 0, 1.5, fake, 1, 16, 64, 1e-5, None) == 1          # alpha outside [0,1]


def test_mosaic_of_embedded_gradients():
    """ops.mosaic_shape / autograd._mosaic_per_row: the layout of a batch's embedded gradients (host arithmetic only)"""
    from style_transfer_based_holographic_imaging_amd import autograd as ag
    from style_transfer_based_holographic_imaging_amd import ops
    assert ops.mosaic_shape(16, 32, 32, 2, False) == (36, 36, 36)                    # separate blocks
    assert ops.mosaic_shape(16, 32, 32, 2, True) == (34, 15 * 34 + 36, 36)           # one column: neighbours share two zero rows
    assert ops.mosaic_shape(16, 32, 32, 2, True, 7) == (34, 2 * 34 + 36, 6 * 34 + 36)   # 3 rows of 7 (the last one ragged)
    assert ops.mosaic_shape(16, 32, 32, 2, True, 2) == (34, 7 * 34 + 36, 34 + 36)
    assert ag._mosaic_per_row(1, 32, 32, 512) == 0                                   # a single image: no mosaic
    assert ag._mosaic_per_row(16, 32, 32, 512) == 2 and ag._mosaic_per_row(16, 256, 256, 64) == 2    # 8 rows of 2: 45 tiles (96 apart, 54 in a column)
    assert ag._mosaic_per_row(64, 1024, 1024, 64) == 0                               # past 2^30 elements per image: separate blocks
    for b, h, w in ((16, 32, 32), (16, 64, 64), (5, 40, 56), (3, 128, 128), (16, 256, 256)):
        gx = ag._mosaic_per_row(b, h, w, 64)
        _, rows, width = ops.mosaic_shape(b, h, w, 2, True, gx)
        area = lambda r, c: -(-r // 32) * 32 * (-(-c // 16) * 16)
        assert all(area(rows, width) <= area(*ops.mosaic_shape(b, h, w, 2, True, g)[1:]) for g in range(1, min(b, 8) + 1))
