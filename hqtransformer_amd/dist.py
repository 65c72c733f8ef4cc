"""Sample-sharded multi-GPU sampling.

Every image is an independent AR chain + decode, so a global batch is cut into contiguous per-rank slices,
weights are replicated, one process drives one GPU, and the only collective is the gather of finished
results to rank 0 (RCCL over xGMI on GPUs; gloo in the CPU tests).  Class ids and Philox noise are keyed by the
GLOBAL sample index (``sample_offset``), so a sharded run reproduces the one-GPU run of the same global batch.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of rank `rank`; the first `global_batch % world` ranks get one extra sample."""
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_to_rank0(t: torch.Tensor, sizes: Sequence[int], group=None) -> Optional[torch.Tensor]:
    """Gathers per-rank tensors of (possibly different) leading size to rank 0 and concatenates them."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return t
    m = max(sizes)
    pad = t
    if t.shape[0] < m:
        pad = torch.cat([t, t.new_zeros((m - t.shape[0],) + tuple(t.shape[1:]))], dim=0)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad.contiguous(), bufs, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def sample_and_decode_sharded(sample_fn: Callable, decode_fn: Callable, global_batch: int, cond, seed: int,
                              gather: str = 'pixels', group=None):
    """Runs ``sample_fn(batch, cond_slice, seed, sample_offset) -> (codes_top, codes_bot)`` and
    ``decode_fn(codes_top, codes_bot) -> pixels`` on this rank's slice and gathers to rank 0.

    ``cond``: None, an int, or a tensor whose first dimension is the global batch.  Returns on rank 0
    ``(codes_top, codes_bot, pixels)`` of the global batch (pixels None when gather == 'codes'), None elsewhere.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(global_batch, world, rank)
    sizes = [shard_bounds(global_batch, world, r)[1] - shard_bounds(global_batch, world, r)[0] for r in range(world)]
    cond_slice = cond[lo:hi] if torch.is_tensor(cond) else cond
    ct, cb = sample_fn(hi - lo, cond_slice, seed, lo)
    px = decode_fn(ct, cb) if gather == 'pixels' else None
    if world == 1:
        return ct, cb, px
    ct_all = gather_to_rank0(ct, sizes, group)
    cb_all = gather_to_rank0(cb, sizes, group)
    px_all = gather_to_rank0(px, sizes, group) if px is not None else None
    return (ct_all, cb_all, px_all) if rank == 0 else None
