"""Known answers for oracle/unwrap_oracle.py (the restatement of the published algorithm behind
skimage.restoration.unwrap_phase, which the reference calls at utils/functions.py:54)."""
import os
import sys

import numpy as np
import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "oracle"))
import unwrap_oracle as uo  # noqa: E402
import unwrap_cases as uc  # noqa: E402


@pytest.mark.parametrize("h,w,seed", [(32, 32, 0), (48, 80, 1), (96, 64, 2), (128, 128, 3)])
def test_smooth_surface_is_recovered_up_to_a_global_cycle(h, w, seed):
    true = uc.smooth_surface(h, w, seed=seed)
    wrapped = uc.wrap(true)
    assert np.abs(wrapped - true).max() > 6.0           # it really is wrapped
    un = uo.unwrap_phase_2d(wrapped)
    m = uc.global_shift(un, true)
    assert np.abs(un - true - 2 * np.pi * m).max() < 5e-6
    k = uo.mode_convention(uo.unwrap_offsets(wrapped))
    vals, counts = np.unique(k, return_counts=True)
    assert vals[np.argmax(counts)] == 0


def test_wrap_free_input_is_returned_unchanged():
    rng = np.random.default_rng(4)
    img = (rng.random((40, 56)) - 0.5).astype(np.float32)
    assert np.array_equal(uo.unwrap_phase_2d(img), img)
    assert np.array_equal(uo.unwrap_batch(img[None, None])[0, 0], img)


def test_tree_offsets_honour_every_tree_edge_and_are_independent_of_history():
    """With residues the answer depends on the spanning tree, never on the merge history: re-deriving the offsets
    from the tree edges alone (breadth-first from pixel 0) gives the same image up to the global cycle."""
    img = uc.noisy_wrapped(40, 36, seed=5)
    k = uo.unwrap_offsets(img)
    p1, p2, jump, weight, eid = uo.edges(img)
    assert np.all(np.diff(weight.astype(np.float64)) >= 0)
    parent = list(range(img.size))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    adj = [[] for _ in range(img.size)]
    for a, b, j in zip(p1.tolist(), p2.tolist(), jump.tolist()):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
            adj[a].append((b, -j))        # k[b] = k[a] - j
            adj[b].append((a, j))
    kk = {0: 0}
    todo = [0]
    while todo:
        a = todo.pop()
        for b, dj in adj[a]:
            if b not in kk:
                kk[b] = kk[a] + dj
                todo.append(b)
    tree = np.array([kk[i] for i in range(img.size)]).reshape(img.shape)
    assert np.array_equal(tree - tree.flat[0], k - k.flat[0])
    assert len(np.unique(k)) > 2                              # noise really produced several cycles


def test_degenerate_shapes():
    for shape in [(1, 1), (1, 9), (7, 1), (2, 2), (3, 3)]:
        img = uc.wrap(np.linspace(0, 9, shape[0] * shape[1]).reshape(shape))
        un = uo.unwrap_phase_2d(img)
        d = np.diff(un.ravel())
        assert np.all(np.abs(d) <= np.pi + 1e-5) or min(shape) > 1
