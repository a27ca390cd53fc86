"""Multi-GPU plumbing: one process per GPU, blocks sharded across ranks, one all-gather of per-block
results (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).

The scoring path has no exchange step inside a block: every (block, sample) unit is independent
given the block constants, and seeds depend only on the sample index, so any sharding gives the
same numbers.  The only collective is the final gather of
`maxima[blocks][sampleN]` or, for callers that only need p-values, of the fit results `[blocks][4]`
(evd_rc, mu, lambda, samples above the best native score: 16 B per block)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of rank's items (first n % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_blocks(blocks: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_range(len(blocks), rank, world)
    return list(blocks[lo:hi])


def all_gather_rows(local, counts: Sequence[int]):
    """Gather per-rank tensors [n_r, width] (one row per block) into one [sum n_r, width] tensor on
    every rank, in rank order.  Equal shards use a single all_gather_into_tensor (one ring pass);
    ragged shards are padded to the largest shard first."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    n_max = max(counts)
    if local.shape[0] != n_max:
        pad = torch.full((n_max - local.shape[0], local.shape[1]), -1.0, dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * n_max, local.shape[1]), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(out, local.contiguous())
    except (RuntimeError, NotImplementedError):   # backends without the fused form
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous())
        out = torch.cat(parts, dim=0)
    if all(c == n_max for c in counts):
        return out
    return torch.cat([out[r * n_max:r * n_max + counts[r]] for r in range(world)], dim=0)


all_gather_maxima = all_gather_rows   # [n_r, sampleN] per-sample maxima
all_gather_fits = all_gather_rows     # [n_r, 4] rows of rc_batch_fit_all


def max_over_ranks(seconds: float, device=None) -> float:
    """Wall time of the slowest rank (bench contract: MAX over ranks)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
