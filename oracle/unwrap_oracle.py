"""CPU oracle for 2-D phase unwrapping.  TEST INFRASTRUCTURE ONLY (see holo_oracle.py for the rules).

The reference unwraps the focused phase on the host, one image at a time, with
``skimage.restoration.unwrap_phase`` (utils/functions.py:2,44-59; called from
utils/Forward_model.py:30-31 and test_field_retrieval_mnist.py:126).  scikit-image is a third-party
dependency the reference neither vendors nor pins (no requirements file), and it is not installed in
this image, so this file restates the PUBLISHED algorithm that function implements:

    M. A. Herraez, D. R. Burton, M. J. Lalor, M. A. Gdeisat, "Fast two-dimensional phase-unwrapping
    algorithm based on sorting by reliability following a noncontinuous path", Appl. Opt. 41(35), 2002

as scikit-image's 2-D unwrapper applies it with ``wrap_around=False`` and no mask:

  1. reliability of an interior pixel = H^2 + V^2 + D1^2 + D2^2, the squared wrapped second differences
     along the row, the column and the two diagonals (float64 arithmetic on the float32 phase);
  2. every horizontal and vertical neighbour pair is an edge with weight rel(p1) + rel(p2) and jump
     -1/0/+1 according to whether value(p1) - value(p2) is > pi, in [-pi, pi], or < -pi;
  3. edges are visited in ascending weight; an edge whose pixels lie in different groups joins the
     groups, and the SMALLER group (the lone pixel; the first pixel's group when sizes are equal and
     both have grown) has the integer offsets of all its pixels shifted so the edge's jump is honoured;
  4. output = value + 2 pi * offset.

Parity status: UNPINNED against scikit-image (absent; version unknown).  Two places where that library
is itself not deterministic are fixed here and in the HIP kernel the same way:
  * border pixels: the library gives them reliability 9999999 + rand(); here it is the constant 9999999;
  * ties: the library's quicksort is unstable; here edges are ordered by (float32(weight), edge id), edge
    id = 2*pixel + (0 right | 1 down).
What IS pinned: analytic known answers (a wrapped smooth surface has exactly one unwrapping up to a
global multiple of 2 pi) in tests/test_oracle_golden.py.

Step 3 is Kruskal's algorithm, so the set of joining edges is the minimum spanning tree of the pixel grid
under that total order, and every pixel's offset relative to any other is the sum of jumps along the tree
path.  Only the GLOBAL multiple of 2 pi depends on the merge history.  ``unwrap_phase_2d`` follows the
published history rule; ``mode_convention`` re-bases a result so the most frequent offset is zero, which
is the convention of the HIP kernel (holo_unwrap2d_f32).
"""
from __future__ import annotations

import numpy as np

PI = np.pi
TWO_PI = 2.0 * np.pi
BORDER_RELIABILITY = 9999999.0


def _wrap(d):
    return np.where(d > PI, d - TWO_PI, np.where(d < -PI, d + TWO_PI, d))


def reliability(phase):
    """Step 1.  phase (H,W) float32 -> (H,W) float64; the one-pixel frame holds BORDER_RELIABILITY."""
    v = np.asarray(phase, dtype=np.float32).astype(np.float64)
    h, w = v.shape
    rel = np.full((h, w), BORDER_RELIABILITY, dtype=np.float64)
    if h >= 3 and w >= 3:
        c = v[1:-1, 1:-1]
        hh = _wrap(v[1:-1, :-2] - c) - _wrap(c - v[1:-1, 2:])
        vv = _wrap(v[:-2, 1:-1] - c) - _wrap(c - v[2:, 1:-1])
        d1 = _wrap(v[:-2, :-2] - c) - _wrap(c - v[2:, 2:])
        d2 = _wrap(v[:-2, 2:] - c) - _wrap(c - v[2:, :-2])
        rel[1:-1, 1:-1] = ((hh * hh + vv * vv) + d1 * d1) + d2 * d2
    return rel


def edges(phase):
    """Step 2 + the ordering of step 3.  Returns (p1, p2, jump, weight32, edge_id) sorted by visit order."""
    v = np.asarray(phase, dtype=np.float32).astype(np.float64)
    h, w = v.shape
    rel = reliability(phase)
    idx = np.arange(h * w).reshape(h, w)
    p1 = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()])
    p2 = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
    eid = np.concatenate([2 * idx[:, :-1].ravel(), 2 * idx[:-1, :].ravel() + 1])
    fv, fr = v.ravel(), rel.ravel()
    weight = (fr[p1] + fr[p2]).astype(np.float32)
    diff = fv[p1] - fv[p2]
    jump = np.where(diff > PI, -1, np.where(diff < -PI, 1, 0)).astype(np.int64)
    order = np.lexsort((eid, weight))
    return p1[order], p2[order], jump[order], weight[order], eid[order]


def unwrap_offsets(phase):
    """Steps 2-3 with the published merge-history rule.  Returns the integer offsets (H,W)."""
    h, w = np.asarray(phase).shape
    n = h * w
    p1s, p2s, jumps, _, _ = edges(phase)
    inc = [0] * n
    head = list(range(n))
    members = [[i] for i in range(n)]
    for a, b, j in zip(p1s.tolist(), p2s.tolist(), jumps.tolist()):
        ga, gb = head[a], head[b]
        if ga == gb:
            continue
        ma, mb = members[ga], members[gb]
        if len(mb) == 1:                       # second pixel alone: it joins the first pixel's group
            inc[b] = inc[a] - j
            head[b] = ga
            ma.append(b)
            members[gb] = None
        elif len(ma) == 1:                     # first pixel alone
            inc[a] = inc[b] + j
            head[a] = gb
            mb.append(a)
            members[ga] = None
        elif len(ma) > len(mb):                # the smaller group moves
            shift = inc[a] - j - inc[b]
            for p in mb:
                inc[p] += shift
                head[p] = ga
            ma.extend(mb)
            members[gb] = None
        else:
            shift = inc[b] + j - inc[a]
            for p in ma:
                inc[p] += shift
                head[p] = gb
            mb.extend(ma)
            members[ga] = None
    return np.asarray(inc, dtype=np.int64).reshape(h, w)


def apply_offsets(phase, offsets):
    """Step 4: float64 sum, returned in the input's float32."""
    return (np.asarray(phase, dtype=np.float32).astype(np.float64) + TWO_PI * offsets).astype(np.float32)


def unwrap_phase_2d(phase):
    """The whole published algorithm on one (H,W) float32 image."""
    return apply_offsets(phase, unwrap_offsets(phase))


def mode_convention(offsets):
    """Re-base integer offsets so the most frequent one is 0 (ties: the lowest offset wins)."""
    vals, counts = np.unique(offsets, return_counts=True)
    return offsets - vals[np.argmax(counts)]


def unwrap_batch(x, convention="mode"):
    """utils/functions.py:44-59 for a (B,1,H,W) / (B,H,W) float32 array: per-image unwrapping."""
    x = np.asarray(x, dtype=np.float32)
    imgs = x.reshape((-1,) + x.shape[-2:])
    out = np.empty_like(imgs)
    for i, img in enumerate(imgs):
        off = unwrap_offsets(img)
        if convention == "mode":
            off = mode_convention(off)
        out[i] = apply_offsets(img, off)
    return out.reshape(x.shape[0], 1, *x.shape[-2:])
