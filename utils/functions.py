"""Drop-in for the hot-path helpers of ``utils/functions.py``: ``field_retrieval`` (functions.py:22-29)
and ``unwrap`` (functions.py:44-59).  The plotting / loss helpers of the reference are not on the
inference path."""
import torch


def field_retrieval(network, content, style_vector, alpha=1.0, unkonwn_distance=False, intensity=False):
    """(amplitude, phase[, distance (B,1,1,1)]) — the reference's keyword is misspelt and kept as is.
    ``intensity`` (additive, default off): ``content`` is the hologram intensity; its square root
    (test_field_retrieval_mnist.py:119) is taken by the first layer's loader."""
    assert (0.0 <= alpha <= 1.0)
    extra = {"intensity": True} if intensity else {}
    if not unkonwn_distance:
        return network.field_retrieval(content, style_vector, alpha, **extra)
    amplitude, phase, dist = network.field_retrieval(content, style_vector, alpha, unkonwn_distance, **extra)
    return amplitude, phase, dist.view(-1, 1, 1, 1)


def unwrap(x):
    """Per-image 2-D phase unwrapping, (B,1,H,W) in and out.

    Device tensors: ``holo_unwrap2d_f32`` — the reliability-ordered spanning tree of the published algorithm
    behind scikit-image's ``unwrap_phase``, built on the GPU for the whole batch (csrc/unwrap.hip); the result
    stays on the device in float32, where the reference returns a host tensor.  Host tensors take the
    reference's own route through scikit-image and raise when it is not installed."""
    if x.is_cuda:
        from style_transfer_based_holographic_imaging_amd import ops
        return ops.unwrap2d(x.detach())
    try:
        from skimage.restoration import unwrap_phase
    except ImportError as e:
        raise ImportError("utils.functions.unwrap on a host tensor needs scikit-image "
                          "(skimage.restoration.unwrap_phase), which is not installed") from e
    x = x.detach().numpy()
    out = []
    for i in range(x.shape[0]):
        ph = x[i]
        if ph.ndim > 2:
            ph = ph.squeeze()
        out.append(torch.from_numpy(unwrap_phase(ph)).unsqueeze(0).unsqueeze(0))
    return torch.cat(out, dim=0)
