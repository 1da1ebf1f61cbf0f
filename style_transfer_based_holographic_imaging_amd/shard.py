"""Batch sharding of holograms over the GPUs of one node and the single gather at the end.

Holograms are independent (no cross-sample op on the path), so rank r of G takes the contiguous
block [r*N/G, (r+1)*N/G) with replicated weights and there is no data-path collective; the only
exchange is one gather of the packed outputs to rank 0 (RCCL over xGMI on the GPU box: every peer
writes over its own link into the root).  The same code runs under ``gloo`` on CPU tensors for the
world_size-2 tests.  One process per GPU, launched by ``torch.distributed.run``.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous block of ``n`` items owned by ``rank``: sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def force_collectives() -> bool:
    """HOLO_FORCE_DIST=1: initialise the process group and run the gather even with ONE rank, so that the RCCL path (backend
    start-up, device binding, the asynchronous gather and its wait) executes on a one-GPU box (tests/test_hip_rccl.py)."""
    return os.environ.get("HOLO_FORCE_DIST") == "1"


def init_from_env(backend: str | None = None):
    """(rank, world, local_rank); initialises the process group when WORLD_SIZE > 1 (or when HOLO_FORCE_DIST=1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(min(local, torch.cuda.device_count() - 1) if os.environ.get("HOLO_SHARE_GPU") == "1" else local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def pack_outputs(amp, ph, amp_foc, ph_foc, distance, out=None):
    """(B,1,S,S) x4 + (B,1,1,1) -> one (B, 4*S*S + 1) float32 buffer: a single message per rank.
    ``out``: rows of a preallocated block to fill instead (the strong-scaling pass packs micro-batch by micro-batch)."""
    b = amp.shape[0]
    parts = [amp.reshape(b, -1), ph.reshape(b, -1), amp_foc.reshape(b, -1), ph_foc.reshape(b, -1), distance.reshape(b, 1)]
    if out is None:
        return torch.cat(parts, dim=1).contiguous()
    return torch.cat(parts, dim=1, out=out)


def packed_width(s: int) -> int:
    return 4 * s * s + 1


def micro_batches(lo: int, hi: int, micro_batch: int):
    """Split the block [lo, hi) into ceil(n / micro_batch) near-equal pieces (sizes differ by at most one): no short tail,
    so every piece of a block of >= 16 holograms falls in one batch bucket of the convolution-algorithm choice
    (engine.Plan.BATCH_BUCKETS) and a hologram's arithmetic does not depend on where the block boundaries fall."""
    n = hi - lo
    if n <= 0:
        return []
    pieces = -(-n // max(1, micro_batch))
    return [tuple(lo + v for v in shard_bounds(n, pieces, i)) for i in range(pieces)]


def block_rows(n_total: int, world: int) -> int:
    """Rows of the largest rank block of a fixed-total job: the size every rank's send buffer is padded to."""
    return -(-n_total // world)


def sharded_pass(n_total: int, micro_batch: int, process, width: int, device, dtype=torch.float32, async_op: bool = False,
                 block=None, recv=None):
    """One pass over a FIXED total of ``n_total`` holograms (strong scaling; BASELINE configs[3]: 2048 holograms over
    1/2/4/8 GPUs): this rank's contiguous block ``shard_bounds(n_total, world, rank)`` is processed in micro-batches —
    ``process(m0, m1, out_rows)`` handles GLOBAL holograms [m0, m1) and writes their packed outputs into ``out_rows``
    ((m1 - m0, width)) — and the block goes to rank 0 with ONE gather (ragged blocks allowed).  Returns what
    ``gather_to_root`` returns.  No data-path collective besides that gather.

    ``block``: a send buffer kept across passes, of ``block_rows(n_total, world)`` rows (the padded size: a ragged job then
    gathers it as it stands, nothing is copied or allocated per pass) or of exactly this rank's rows; ``recv``: the root's
    (world, block_rows, width) receive buffer, kept across passes likewise."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(n_total, world, rank)
    if block is None:
        block = torch.zeros((block_rows(n_total, world), width), dtype=dtype, device=device)
    assert block.shape[1] == width and block.shape[0] in (hi - lo, block_rows(n_total, world)), block.shape
    for m0, m1 in micro_batches(lo, hi, micro_batch):
        process(m0, m1, block[m0 - lo:m1 - lo])
    counts = [shard_bounds(n_total, world, q)[1] - shard_bounds(n_total, world, q)[0] for q in range(world)]
    return gather_to_root(block, counts, async_op=async_op, recv=recv)


def unpack_outputs(packed, s):
    b = packed.shape[0]
    n = s * s
    planes = [packed[:, k * n:(k + 1) * n].reshape(b, 1, s, s) for k in range(4)]
    return (*planes, packed[:, 4 * n:].reshape(b, 1, 1, 1))


class PendingGather:
    """A gather in flight.  ``wait()`` makes the current stream wait for it and returns what ``gather_to_root``
    returns.  Holding the handle keeps the send/receive buffers alive, so the next step's kernels can be enqueued
    before the collective has finished: the exchange rides under the following step's compute."""

    def __init__(self, work, out, counts, is_root, keep):
        self._work, self._out, self._counts, self._is_root, self._keep = work, out, counts, is_root, keep

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._keep = None
        if not self._is_root or self._out is None:
            return None
        return self._out if self._counts is None else _trim(self._out, self._counts)


def gather_to_root(packed: torch.Tensor, counts: list[int] | None = None, dst: int = 0, async_op: bool = False, recv=None):
    """One gather of every rank's packed block to ``dst``.  Returns the concatenated (N_total, F)
    tensor on ``dst`` and None elsewhere (``async_op``: a ``PendingGather`` whose ``wait()`` returns that).
    ``counts`` (rows per rank) allows ragged shards: every rank sends max(counts) rows and the root trims.  A ``packed``
    that already has max(counts) rows (its own rows first; ``sharded_pass`` keeps such a buffer) is sent as it stands; a
    shorter one is copied into a padded buffer first.  ``recv``: a (world, max(counts), F) receive buffer to reuse on
    ``dst`` (allocated per call otherwise)."""
    own = packed.shape[0] if counts is None else None
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collectives()):
        if counts is not None and packed.shape[0] > counts[0]:
            packed = packed[:counts[0]]
        return PendingGather(None, packed, None, True, None) if async_op else packed
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = counts or [own] * world
    rows = max(counts)
    if packed.shape[0] < rows:
        padded = torch.zeros((rows, packed.shape[1]), dtype=packed.dtype, device=packed.device)
        padded[:packed.shape[0]] = packed
        packed = padded
    assert packed.shape[0] == rows, (packed.shape, rows)
    shape = (world, rows, packed.shape[1])

    def _recv():
        if recv is not None:
            assert tuple(recv.shape) == shape and recv.dtype == packed.dtype and recv.device == packed.device, (recv.shape, shape)
            return recv
        return torch.empty(shape, dtype=packed.dtype, device=packed.device)

    global _GATHER_UNSUPPORTED
    work, out = None, None
    if not _GATHER_UNSUPPORTED:
        try:
            if rank == dst:
                out = _recv()
                work = dist.gather(packed, list(out.unbind(0)), dst=dst, async_op=True)
            else:
                work = dist.gather(packed, None, dst=dst, async_op=True)
        except (RuntimeError, NotImplementedError) as e:   # a backend without gather: every rank sees the same error
            _GATHER_UNSUPPORTED = True
            work, out = None, None
            if rank == dst:
                print(f"[shard] dist.gather unavailable ({type(e).__name__}); using all_gather_into_tensor", flush=True)
    if work is None:
        out = _recv() if rank == dst else torch.empty(shape, dtype=packed.dtype, device=packed.device)
        work = dist.all_gather_into_tensor(out, packed, async_op=True)
    pending = PendingGather(work, out, counts, rank == dst, packed)
    return pending if async_op else pending.wait()


_GATHER_UNSUPPORTED = False


def _trim(out, counts):
    if all(c == out.shape[1] for c in counts):
        return out.reshape(-1, out.shape[2])
    return torch.cat([out[r, :counts[r]] for r in range(out.shape[0])], 0)
