"""Phase maps shared by the CPU (oracle) and GPU (kernel) unwrapping tests."""
import numpy as np


def smooth_surface(h, w, peak=18.0, seed=0):
    """A surface whose neighbour differences stay below pi: its wrapping has exactly one unwrapping
    up to a global multiple of 2 pi (the analytic known answer)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    y, x = y / h, x / w
    cx, cy = 0.3 + 0.4 * rng.random(), 0.3 + 0.4 * rng.random()
    true = peak * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / 0.08) + 6.0 * x - 3.0 * y
    step = max(np.abs(np.diff(true, axis=0)).max(), np.abs(np.diff(true, axis=1)).max())
    assert step < 2.5, step
    return true


def wrap(true):
    return np.angle(np.exp(1j * true)).astype(np.float32)


def noisy_wrapped(h, w, sigma=0.9, seed=0):
    """Wrapped surface + strong noise: has residues, so the result depends on the tree, not only on the data."""
    rng = np.random.default_rng(seed)
    return wrap(smooth_surface(h, w, 14.0, seed) + sigma * rng.standard_normal((h, w)))


def vortex(h, w):
    """A phase singularity (two residues of opposite sign far apart are impossible to avoid here)."""
    y, x = np.mgrid[0:h, 0:w]
    return np.arctan2(y - h / 2 + 0.5, x - w / 2 + 0.5).astype(np.float32)


def global_shift(unwrapped, true):
    """Integer m minimising |unwrapped - true - 2 pi m|."""
    return np.round(np.mean(unwrapped.astype(np.float64) - true) / (2 * np.pi))
