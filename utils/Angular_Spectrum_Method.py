"""Drop-in for ``utils/Angular_Spectrum_Method.py``: the angular-spectrum propagator on HIP kernels.

``ASM(O, lamb, d, px, requires_grad=True, zero_padding=False)`` keeps the reference signature
(Angular_Spectrum_Method.py:7): complex ``(B,C,S,S)`` in, complex128 ``(B,C,S,S)`` out; ``d`` is a
tensor broadcastable to ``(B,1,1,1)`` in metres or a Python float.  The three fused kernels
(row FFT, column FFT x transfer function x column IFFT, row IFFT) run in complex64 with a float64
phase argument; ``requires_grad`` is accepted and ignored (inference path).
"""
import torch

from style_transfer_based_holographic_imaging_amd import _lib, ops


def _per_field_constant(d, batch, channels, device):
    if isinstance(d, torch.Tensor) and d.numel() > 1:
        if d.numel() != batch:
            raise RuntimeError(f"distance tensor of {d.numel()} entries does not match batch {batch}")
        d = d.reshape(batch, 1).expand(batch, channels).reshape(-1)
    return ops.phase_constant(d, batch * channels, device)


def ASM(O, lamb, d, px, requires_grad=True, zero_padding=False):
    batch, c, sh, sw = O.shape
    if sh != sw:
        raise RuntimeError("ASM works on square fields (the reference's pad/crop arithmetic mixes H and W)")
    if not O.is_complex():
        O = O.to(torch.complex64)
    phase_c = _per_field_constant(d, batch, c, O.device)
    out, _ = ops.asm_propagate(O, None, phase_c, sh, zero_padding, lamb, px, 1.0, _lib.ASM_OUT_COMPLEX)
    return out.to(torch.complex128)


def center_crop(H, size):
    batch, channel, Nh, Nw = H.size()
    return H[:, :, (Nh - size) // 2:(Nh + size) // 2, (Nw - size) // 2:(Nw + size) // 2]
