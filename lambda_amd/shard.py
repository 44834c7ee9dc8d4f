"""Query sharding across the GPUs of one node and the final top-hit gather.

Extensions are independent and the reference already hands each OpenMP thread a contiguous chunk of queries
(/root/reference/src/search.cpp:384-385).  Here rank r of W owns queries [r*Q/W, (r+1)*Q/W), runs both passes locally and
the only collective is one gather of fixed-size hit records at the end: `all_gather` of per-rank counts followed by a
padded `gather` of the records to the rank that writes the output (the reference has one output file) or, with
dst=None, a padded `all_gather` -- what bench.py uses: over point-to-point xGMI both cost about the same, ~2 ms for
8 x 90 MB (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Query ranges are disjoint, so concatenating in rank
order is already the final order; no merge is needed.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_queries: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced query range of `rank` (first `n % world` ranks get one extra query)."""
    base, extra = divmod(n_queries, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_hits(local: torch.Tensor, group=None, dst: int | None = 0) -> torch.Tensor:
    """Concatenates the [n_r, F] hit-record tensors of all ranks in rank order on rank `dst` (the other ranks get an
    empty [0, F] tensor); dst=None: on every rank (all_gather)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts) if counts else 0
    padded = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    if dst is None:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(bufs, padded, group=group)
    else:
        bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
        dist.gather(padded, gather_list=bufs, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return local[:0]
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
