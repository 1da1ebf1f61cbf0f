"""RCCL on the one-GPU box: the collective path of shard.py / bench.py with ONE rank (HOLO_FORCE_DIST=1).  RCCL refuses two ranks on
one device, so this is what a single card can execute of it: backend start-up, device binding, the asynchronous gather into the root's
buffer, its wait, the barrier and the MAX all-reduce of the timing.  The multi-rank control flow (ragged blocks, padding, trimming,
pipelining) is covered under gloo in tests/test_shard_gloo.py; each test runs in a child process (a process group is global state)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ)
    env.update({"HOLO_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONPATH": REPO})
    env.pop("HOLO_DIST_BACKEND", None)
    return env


GATHER = r"""
import torch, torch.distributed as dist
from style_transfer_based_holographic_imaging_amd import shard
rank, world, local = shard.init_from_env("nccl")
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
g = torch.Generator().manual_seed(5)
amp, ph, af, pf = (torch.randn(6, 1, 16, 16, generator=g).to(dev) for _ in range(4))
d = torch.rand(6, 1, 1, 1, generator=g).to(dev)
packed = shard.pack_outputs(amp, ph, af, pf, d)
got = shard.gather_to_root(packed)                               # blocking form
assert got.shape == (6, shard.packed_width(16)) and torch.equal(got, packed)
handles = [shard.gather_to_root(packed * k, async_op=True) for k in (1.0, 2.0, 3.0)]      # three gathers in flight
for k, h in zip((1.0, 2.0, 3.0), handles):
    assert torch.equal(h.wait(), packed * k)
out = shard.unpack_outputs(got, 16)
assert all(torch.equal(a, b) for a, b in zip(out, (amp, ph, af, pf, d)))
rows = []
res = shard.sharded_pass(10, 4, lambda m0, m1, block: (rows.append((m0, m1)), block.fill_(float(m0)))[0], 3, dev)
assert rows == [(0, 4), (4, 7), (7, 10)] and res.shape == (10, 3) and float(res[5, 0]) == 4.0
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t) == 1.5
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_gather_through_rccl_one_rank():
    r = subprocess.run([sys.executable, "-c", GATHER], env=_env(29611), cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("extra", [[], ["--total", "10"]])
def test_bench_step_through_rccl_one_rank(extra):
    """bench.py's weak step (one asynchronous gather per step, waited for one step later) and its fixed-total pass, with the gather,
    the barrier and the timing all-reduce on RCCL."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--batch", "4", "--size", "64", "--steps", "3", "--warmup", "1",
           "--cpu-sample", "0"] + extra
    r = subprocess.run(cmd, env=_env(29612), cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert "RCCL gather" in line["config"]["parallelism"]
    assert line["scaling"] == ("strong" if extra else "weak")


def test_two_ranks_on_one_card_gather_the_one_rank_block(tmp_path):
    """bench.py --total 96 --batch 16 with TWO ranks sharing the card (gloo + HOLO_SHARE_GPU=1: RCCL refuses two ranks on one
    device): the block gathered on rank 0 equals the block of a one-rank run bit for bit — the shards, the micro-batches, the
    padded send buffers and the root's receive buffer put every hologram's outputs where the unsharded pass puts them — and the
    line carries the gather's own time.  A ragged total (95) goes through the padded path."""
    import torch
    blocks = {}
    for total in (96, 95):
        for world in (1, 2):
            path = str(tmp_path / f"block_{total}_{world}.pt")
            args = ["bench.py", "--gpus", str(world), "--total", str(total), "--batch", "16", "--size", "64", "--steps", "2", "--warmup", "1",
                    "--cpu-sample", "0", "--no-events", "--dump-block", path]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=REPO)
            for k in ("HOLO_FORCE_DIST", "RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            if world == 1:
                cmd = [sys.executable] + args
            else:
                env.update(HOLO_DIST_BACKEND="gloo", HOLO_SHARE_GPU="1")
                cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                       "--master-port", str(29620 + (total % 2))] + args
            r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
            line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            assert line["n_gpus"] == world and line["scaling"] == "strong" and line["config"]["total_holograms"] == total
            if world == 2:
                assert line["gather_ms"] > 0 and "gloo gather" in line["config"]["parallelism"]
                assert line["roofline"]["traffic"] is None if "roofline" in line else True
            blocks[(total, world)] = torch.load(path)
        one, two = blocks[(total, 1)], blocks[(total, 2)]
        assert one.shape == two.shape == (total, 4 * 64 * 64 + 1)
        assert torch.equal(one, two), total
        assert not torch.equal(one[0], one[total - 1])
