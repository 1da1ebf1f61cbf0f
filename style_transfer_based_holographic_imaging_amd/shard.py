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


def init_from_env(backend: str | None = None):
    """(rank, world, local_rank); initialises the process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(min(local, torch.cuda.device_count() - 1) if os.environ.get("HOLO_SHARE_GPU") == "1" else local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def pack_outputs(amp, ph, amp_foc, ph_foc, distance):
    """(B,1,S,S) x4 + (B,1,1,1) -> one (B, 4*S*S + 1) float32 buffer: a single message per rank."""
    b = amp.shape[0]
    return torch.cat([amp.reshape(b, -1), ph.reshape(b, -1), amp_foc.reshape(b, -1), ph_foc.reshape(b, -1),
                      distance.reshape(b, 1)], dim=1).contiguous()


def unpack_outputs(packed, s):
    b = packed.shape[0]
    n = s * s
    planes = [packed[:, k * n:(k + 1) * n].reshape(b, 1, s, s) for k in range(4)]
    return (*planes, packed[:, 4 * n:].reshape(b, 1, 1, 1))


def gather_to_root(packed: torch.Tensor, counts: list[int] | None = None, dst: int = 0):
    """One gather of every rank's packed block to ``dst``.  Returns the concatenated (N_total, F)
    tensor on ``dst`` and None elsewhere.  ``counts`` (rows per rank) allows ragged shards: blocks
    are padded to the largest and trimmed on the root."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return packed
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = counts or [packed.shape[0]] * world
    rows = max(counts)
    if packed.shape[0] < rows:
        pad = torch.zeros((rows - packed.shape[0], packed.shape[1]), dtype=packed.dtype, device=packed.device)
        packed = torch.cat([packed, pad], 0)
    global _GATHER_UNSUPPORTED
    if not _GATHER_UNSUPPORTED:
        try:
            if rank == dst:
                out = torch.empty((world, rows, packed.shape[1]), dtype=packed.dtype, device=packed.device)
                dist.gather(packed, list(out.unbind(0)), dst=dst)
                return out[0] if world == 1 else _trim(out, counts)
            dist.gather(packed, None, dst=dst)
            return None
        except (RuntimeError, NotImplementedError) as e:   # a backend without gather: every rank sees the same error
            _GATHER_UNSUPPORTED = True
            if rank == dst:
                print(f"[shard] dist.gather unavailable ({type(e).__name__}); using all_gather_into_tensor", flush=True)
    out = torch.empty((world, rows, packed.shape[1]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed)
    return _trim(out, counts) if rank == dst else None


_GATHER_UNSUPPORTED = False


def _trim(out, counts):
    if all(c == out.shape[1] for c in counts):
        return out.reshape(-1, out.shape[2])
    return torch.cat([out[r, :counts[r]] for r in range(out.shape[0])], 0)
