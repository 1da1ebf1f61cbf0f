"""Drop-in for ``utils/Forward_model.py``: ``Holo_Generator`` on the fused HIP propagator.

``Holo_Generator(args)`` reads ``wavelength, pixel_size, distance_normalize,
distance_normalize_constant, phase_normalize`` from ``args`` (Forward_model.py:9-13);
``forward(amplitude, phase, d, return_field=False, complex_number=False, unwrap=False)`` returns
``(amp, phase)`` float32, the complex128 field, or the float32 intensity (Forward_model.py:16-39).
The field packing ``amplitude*exp(1j*phase)`` and the abs/angle/|.|^2 epilogues are fused into the
first and last propagation kernels, and strided views such as ``g_t[:, :1]`` are consumed in place.
"""
import torch
from torch import nn

from style_transfer_based_holographic_imaging_amd import _lib, ops
from utils.functions import unwrap as uw


class Holo_Generator(nn.Module):
    def __init__(self, args):
        super(Holo_Generator, self).__init__()
        self.wavelength = args.wavelength
        self.pixel_size = args.pixel_size
        self.distance_normalize = args.distance_normalize
        self.distance_normalize_constant = args.distance_normalize_constant
        self.phase_normalize = args.phase_normalize

    def _distance_m(self, d):
        # Forward_model.py:18, float32 tensor arithmetic when d is a tensor
        return ((d + self.distance_normalize_constant) * self.distance_normalize) * 1e-3

    def forward(self, amplitude, phase, d, return_field=False, complex_number=False, unwrap=False):
        if phase.shape != amplitude.shape:
            # the reference multiplies amplitude * exp(1j * phase): torch broadcasting applies (utils/Data_loader.py:59-62
            # passes one amplitude plane with a phase batch); expanded views cost nothing, the kernels take a zero stride
            try:
                amplitude, phase = torch.broadcast_tensors(amplitude, phase)
            except RuntimeError as e:
                raise RuntimeError(f"amplitude {tuple(amplitude.shape)} / phase {tuple(phase.shape)} do not broadcast") from e
        b, c, s, s2 = amplitude.shape
        if s != s2:
            raise RuntimeError(f"amplitude {tuple(amplitude.shape)} / phase {tuple(phase.shape)} must be square")
        if isinstance(d, torch.Tensor) and d.is_cuda and d.dtype == torch.float32 and c == 1 and d.numel() in (1, b):
            # the usual call: per-sample float32 distances on the device -> one launch with the reference's rounding chain
            phase_c = ops.phase_constant_from_distance(d, self.distance_normalize_constant, self.distance_normalize, b, b)
        else:
            d = self._distance_m(d)
            if isinstance(d, torch.Tensor) and d.numel() > 1:
                if d.numel() != b:
                    raise RuntimeError(f"distance tensor of {d.numel()} entries does not match batch {b}")
                d = d.reshape(b, 1).expand(b, c).reshape(-1)
            phase_c = ops.phase_constant(d, b * c, amplitude.device)
        mode = (_lib.ASM_OUT_AMP_PHASE if return_field else
                _lib.ASM_OUT_COMPLEX if complex_number else _lib.ASM_OUT_INTENSITY)
        out0, out1 = ops.asm_propagate(amplitude, phase, phase_c, s, True, self.wavelength, self.pixel_size,
                                       self.phase_normalize, mode)
        if return_field:
            return (out0, uw(out1)) if unwrap else (out0, out1)
        if complex_number:
            return out0.to(torch.complex128)
        return out0

    def sweep(self, amplitude, phase, distances, return_field=True):
        """Propagate every field to all ``distances`` (1-D tensor / list, same units as ``d``): the
        row-FFT pass runs once per field and only the column/inverse passes repeat.  Equivalent to the
        reference idiom ``model_forward(a.repeat(n,1,1,1), p.repeat(n,1,1,1), distances)`` for one field
        (utils/Data_loader.py:59-62).  Returns (B, n, S, S) tensors."""
        b, c, s, _ = amplitude.shape
        if c != 1:
            raise RuntimeError("sweep expects single-channel fields")
        dist = torch.as_tensor(distances, dtype=torch.float32, device=amplitude.device).reshape(-1)
        n = dist.numel()
        phase_c = ops.phase_constant_from_distance(dist, self.distance_normalize_constant, self.distance_normalize, b * n, n)
        mode = _lib.ASM_OUT_AMP_PHASE if return_field else _lib.ASM_OUT_INTENSITY
        out0, out1 = ops.asm_propagate(amplitude, phase, phase_c, s, True, self.wavelength, self.pixel_size,
                                       self.phase_normalize, mode, n_dist=n)
        return (out0, out1) if return_field else out0


class Back_prop(nn.Module):
    """utils/Forward_model.py:42-65 of the reference (never instantiated by its scripts): sqrt(hologram) propagated
    on the UN-padded grid, returned as [abs, angle] ('amp_pha') or [re, im] stacked on the channel axis."""

    _FIELDS = ("amplitude_normalize", "wavelength", "pixel_size", "distance_normalize", "distance_normalize_constant")

    def __init__(self, args):
        super().__init__()
        for name in self._FIELDS:
            setattr(self, name, getattr(args, name))
        self.input_type = args.Holo_G_input

    def forward(self, holo, d):
        from utils.Angular_Spectrum_Method import ASM
        metres = (d + self.distance_normalize_constant) * self.distance_normalize * 0.001
        field = ASM(torch.sqrt(holo), self.wavelength, metres, self.pixel_size) * self.amplitude_normalize
        parts = (field.abs(), field.angle()) if self.input_type == 'amp_pha' else (field.real, field.imag)
        return torch.cat(parts, dim=1)
