"""ctypes binding of ``include/holo_hip.h`` (the C-ABI of ``libholo_hip.so``).

The library is built in-tree by ``build.py`` (``hipcc --offload-arch=gfx950 -shared``).  There is no
fallback: if the shared object is missing or does not export every declared symbol, importing
this module raises — the product path never silently runs on something else.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: torch bundles its own libamdhip64.so.7 / libhsa-runtime64; loading ours before it
#                      would bind the process to /opt/rocm's runtime and split streams and memory across two runtimes

_HERE = os.path.dirname(os.path.abspath(__file__))
# HOLO_HIP_LIB selects another build of the same ABI (A/B experiments); default = the in-tree library
LIB_PATH = os.environ.get("HOLO_HIP_LIB") or os.path.join(_HERE, "libholo_hip.so")

_p = C.c_void_p
_i = C.c_int
_l = C.c_long
_f = C.c_float
_d = C.c_double
_z = C.c_size_t

# name -> (restype, argtypes); mirrors include/holo_hip.h one to one
SIGNATURES = {
    "holo_abi_version": (_i, []),
    "holo_last_error": (C.c_char_p, []),
    "holo_target_arch": (C.c_char_p, []),
    "holo_source_hash": (C.c_char_p, []),
    "holo_debug_set_conv_tile": (_i, [_i]),
    "holo_pack_conv3x3_weight_f32": (_i, [_p, _p, _i, _i, _p]),
    "holo_conv3x3_packed_elems": (_z, [_i, _i]),
    "holo_pack_convt2x2_weight_f32": (_i, [_p, _p, _i, _i, _p]),
    "holo_convt2x2_packed_elems": (_z, [_i, _i]),
    "holo_conv_first_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_conv_first_sqrt_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_conv_first_sqrt_bf16": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_conv_first_sqrt_f32s": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_conv3x3_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "holo_pack_conv3x3_wino_weight_f32": (_i, [_p, _p, _i, _i, _p]),
    "holo_conv3x3_wino_packed_elems": (_z, [_i, _i]),
    "holo_conv3x3_wino_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "holo_conv3x3_wino_splitk_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "holo_conv3x3_wino_splitk_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "holo_convt2x2_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "holo_conv_last_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_channel_mean_std_f32": (_i, [_p, _p, _p, _i, _i, _i, _f, _p]),
    "holo_adain_apply_f32": (_i, [_p, _p, _p, _p, _p, _l, _f, _p, _i, _i, _i, _p]),
    "holo_adain_fused_f32": (_i, [_p, _p, _p, _p, _p, _l, _f, _p, _i, _i, _i, _f, _p]),
    "holo_adain_fused_bf16": (_i, [_p, _p, _p, _p, _p, _l, _f, _p, _i, _i, _i, _f, _p]),
    "holo_adain_fused_f32s": (_i, [_p, _p, _p, _p, _p, _l, _f, _p, _i, _i, _i, _f, _p]),
    "holo_linear_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_rownorm_relu_f32": (_i, [_p, _p, _i, _i, _f, _p]),
    "holo_linear1_sigmoid_f32": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "holo_distance_head_workspace_bytes": (_z, [_i, _i]),
    "holo_distance_head_f32": (_i, [_p] * 11 + [_i, _i, _f, _p, _z, _p]),
    "holo_asm_phase_constants_f32": (_i, [_p, _l, _i, _f, _f, _p, _i, _p]),
    "holo_asm_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "holo_asm_propagate": (_i, [_p, _p, _i, _l, _l, _p, _p, _i, _l, _l, _p, _i, _i, _i, _i, _d, _d, _f, _p, _z, _p]),
    "holo_unwrap2d_workspace_bytes": (_z, [_i, _i, _i]),
    "holo_unwrap2d_f32": (_i, [_p, _l, _p, _l, _i, _i, _i, _p, _z, _p]),
    "holo_pack_conv3x3_weight_bf16": (_i, [_p, _p, _i, _i, _p]),
    "holo_pack_convt2x2_weight_bf16": (_i, [_p, _p, _i, _i, _p]),
    "holo_conv_first_bf16": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_conv3x3_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "holo_convt2x2_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "holo_conv_last_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_channel_mean_std_bf16": (_i, [_p, _p, _p, _i, _i, _i, _f, _p]),
    "holo_adain_apply_bf16": (_i, [_p, _p, _p, _p, _p, _l, _f, _p, _i, _i, _i, _p]),
    "holo_cast_f32_to_f32s": (_i, [_p, _p, _l, _p]),
    "holo_cast_f32s_to_f32": (_i, [_p, _p, _l, _p]),
    "holo_pack_conv3x3_weight_f32s": (_i, [_p, _p, _i, _i, _p]),
    "holo_pack_convt2x2_weight_f32s": (_i, [_p, _p, _i, _i, _p]),
    "holo_conv_first_f32s": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_conv3x3_f32s": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "holo_convt2x2_f32s": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "holo_conv_last_f32s": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_channel_mean_std_f32s": (_i, [_p, _p, _p, _i, _i, _i, _f, _p]),
    "holo_adain_apply_f32s": (_i, [_p, _p, _p, _p, _p, _l, _f, _p, _i, _i, _i, _p]),
    "holo_cast_f32_to_bf16": (_i, [_p, _p, _l, _p]),
    "holo_cast_bf16_to_f32": (_i, [_p, _p, _l, _p]),
    "holo_nhwc_to_nchw_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "holo_nchw_to_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    # backward pass (ABI 5)
    "holo_maxpool2x2_f32": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "holo_mask_scale_f32": (_i, [_p, _p, _f, _p, _l, _p]),
    "holo_conv1x1_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "holo_relu_pool_bwd_pad_workspace_bytes": (_z, [_i]),
    "holo_relu_pool_bwd_pad_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _z, _p]),
    "holo_fold_reflect_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "holo_space_to_depth2_relu_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "holo_channel_sum_workspace_bytes": (_z, [_l, _i]),
    "holo_channel_sum_f32": (_i, [_p, _p, _l, _i, _p, _z, _p]),
    "holo_pix_gemm_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "holo_pix_gemm_f32": (_i, [_p, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "holo_conv_last_bwd_workspace_bytes": (_z, [_i, _i, _i]),
    "holo_conv_last_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "holo_conv_first_dgrad_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "holo_channel_mean_std_wide_workspace_bytes": (_z, [_i, _i, _i]),
    "holo_channel_mean_std_wide_f32": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _z, _p]),
    "holo_mean_std_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_linear_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "holo_rownorm_relu_bwd_f32": (_i, [_p, _p, _p, _i, _i, _f, _p]),
    "holo_sigmoid_bwd_f32": (_i, [_p, _p, _p, _i, _p]),
}

ASM_IN_AMP_PHASE, ASM_IN_COMPLEX = 0, 1
ASM_OUT_AMP_PHASE, ASM_OUT_INTENSITY, ASM_OUT_COMPLEX = 0, 1, 2
ABI_VERSION = 5


class HoloLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.isfile(LIB_PATH):
        raise HoloLibraryError(
            f"{LIB_PATH} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise HoloLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.holo_abi_version() != ABI_VERSION:
        raise HoloLibraryError(f"ABI version {lib.holo_abi_version()} != expected {ABI_VERSION}")
    # a library built from other sources (same symbols, e.g. another weight-packing layout) must not load silently —
    # checked on every rank of a multi-process job.  HOLO_HIP_LIB (an explicitly chosen A/B build) opts out; an install
    # that ships the library without csrc/ has nothing to compare with and says so.  The ROCm version is not part of the
    # comparison (build.needs_build() uses it to decide on a REbuild).
    if not os.environ.get("HOLO_HIP_LIB"):
        from . import build as _build
        if not _build.sources_present():
            import warnings
            warnings.warn(f"{LIB_PATH}: csrc/ is not present, the library's source hash cannot be checked", RuntimeWarning)
        else:
            built, here = lib.holo_source_hash().decode(), _build.source_hash()
            if built != here:
                raise HoloLibraryError(
                    f"{LIB_PATH} was built from other sources (library {built[:12]}, tree {here[:12]}): rebuild it "
                    "(python -c 'import __graft_entry__ as g; g.build()')")
    return lib


lib = _load()


def check(rc: int, what: str):
    if rc != 0:
        msg = lib.holo_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
