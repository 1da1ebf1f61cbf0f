"""CPU: the scheduling protocol of the opt-in dataflow propagator (asm_fused_kernel) terminates and is complete for any
block placement — checked on a discrete-event model of it (tests/dataflow_model.py)."""
import random

import pytest

from dataflow_model import simulate


@pytest.mark.parametrize("seed", range(24))
def test_protocol_completes_under_random_placement(seed):
    rng = random.Random(1000 + seed)
    fields = rng.choice([1, 2, 3, 5, 8, 9, 20])
    n_dist = rng.choice([1, 1, 2])
    n1, n2 = rng.choice([1, 2, 4]), rng.choice([1, 2, 4])
    blocks = [rng.choice([0, 1, 2, 5]) for _ in range(8)]          # XCDs without a block must not matter
    if sum(blocks) == 0:
        blocks[3] = 1
    for nslot in (3, 4):                                            # the kernel's ring is 4 deep; 3 is the pipeline's depth
        verdict, ran = simulate(fields, n_dist, n1, n2, n1, blocks, nslot, seed)
        assert verdict == "ok", (verdict, dict(fields=fields, n_dist=n_dist, n1=n1, n2=n2, blocks=blocks, nslot=nslot))
        assert len(ran) == fields * (n1 + n_dist * (n2 + n1))


def test_a_ring_shallower_than_the_pipeline_deadlocks_in_the_model():
    """the model is sharp enough to see the hazard the kernel's NSLOT >= 3 rule avoids"""
    verdicts = {simulate(20, 1, 4, 1, 4, [5, 2, 0, 5, 1, 5, 1, 1], 1, s)[0] for s in range(3)}
    assert "LIVELOCK/DEADLOCK" in verdicts
