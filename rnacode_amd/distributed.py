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


def all_gather_rows(local, counts: Sequence[int], group=None):
    """Gather per-rank tensors [n_r, width] (one row per block) into one [sum n_r, width] tensor on
    every rank, in rank order.  Equal shards use a single all_gather_into_tensor (one ring pass);
    ragged shards are padded to the largest shard first.  group: another process group than the default one
    (bench.py's host fallback: a gloo group beside the RCCL one)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_max = max(counts)
    if local.shape[0] != n_max:
        pad = torch.full((n_max - local.shape[0], local.shape[1]), -1.0, dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * n_max, local.shape[1]), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):   # backends without the fused form
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous(), group=group)
        out = torch.cat(parts, dim=0)
    if all(c == n_max for c in counts):
        return out
    return torch.cat([out[r * n_max:r * n_max + counts[r]] for r in range(world)], dim=0)


all_gather_maxima = all_gather_rows   # [n_r, sampleN] per-sample maxima


class OverlappedGather:
    """The path's one exchange with its wire time hidden: step i's all-gather runs while step i + 1 is being scored.

    Two source tensors [rows, width] alternate.  `source(i)` is what step i's kernels write (the caller binds it with
    rc_batch_bind_maxima / rc_stream_submit_bound); `launch(i)` starts the all-gather of that tensor without waiting for it
    (RCCL runs it on its own stream over xGMI); `source(i + 2)` first waits, on the host, until gather i has read the buffer
    -- the library's kernels run on streams RCCL knows nothing about, so the reuse rule is enforced here.  `finish()` waits
    for everything and returns the last gathered tensor [world * rows, width].  Equal shards only (one fused
    all_gather_into_tensor per step, no padding copies): ragged shards use all_gather_rows."""

    def __init__(self, rows: int, width: int, device, dtype=None, stage_on_cpu: bool = False):
        import torch
        import torch.distributed as dist

        self._dist = dist
        self._torch = torch
        dtype = dtype or torch.float32
        world = dist.get_world_size()
        self.src = [torch.empty((rows, width), dtype=dtype, device=device) for _ in range(2)]
        # stage_on_cpu (tests on a box with fewer GPUs than ranks): the kernels still write device tensors, the collective runs on
        # host copies over a CPU backend
        self.stage = stage_on_cpu and self.src[0].is_cuda
        out_dev = "cpu" if self.stage else device
        self.out = [torch.empty((world * rows, width), dtype=dtype, device=out_dev) for _ in range(2)]
        self.work = [None, None]
        self.last = None

    def _wait(self, k: int):
        w = self.work[k]
        if w is not None:
            w.wait()                                           # the current torch stream waits for the collective ...
            if self.out[0].is_cuda:
                self._torch.cuda.current_stream().synchronize()    # ... and the host for the stream (CPU backends: wait() blocks)
            self.work[k] = None

    def source(self, i: int):
        self._wait(i & 1)
        return self.src[i & 1]

    def launch(self, i: int):
        k = i & 1
        src = self.src[k].cpu() if self.stage else self.src[k]
        self.work[k] = self._dist.all_gather_into_tensor(self.out[k], src, async_op=True)
        self.last = k

    def finish(self):
        self._wait(0)
        self._wait(1)
        return None if self.last is None else self.out[self.last]


all_gather_fits = all_gather_rows     # [n_r, 4] rows of rc_batch_fit_all


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    """Wall time of the slowest rank (bench contract: MAX over ranks)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


# ---------------------------------------------------------------------------------------------------
# few blocks, many samples (BASELINE configs[0]/[1]): split every block's SAMPLE range across the ranks
# ---------------------------------------------------------------------------------------------------

def sample_range(n_samples: int, rank: int, world: int, granule: int = 64) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank's samples, cut at multiples of `granule` (a wavefront scores 64 samples of one
    block) so that only the last rank has a ragged group."""
    groups = -(-n_samples // granule)
    glo, ghi = shard_range(groups, rank, world)
    return min(n_samples, glo * granule), min(n_samples, ghi * granule)


def score_sample_sharded(score_range, fit, pvalue, gather, n_samples: int, rank: int, world: int, stop_early: bool = False,
                         cutoff: float = 1.0):
    """getExtremeValuePars (score.c:976-1064) with the sample loop :1004-1048 split across ranks and the fit :1050 run
    on the gathered maxima.  Sample s of every block is seeded with seed_base + s whichever rank simulates it, so the
    gathered row equals the single-process one bit for bit.

      score_range(lo, hi) -> (maxima [n_blocks][hi - lo] float32, native)   this rank's samples of every block and the
                              native HSS lists (every rank scores the native blocks: 1/1000 of the work)
      gather(local)       -> [n_blocks][n_samples]                         the path's one all-gather
      fit(row)            -> (rc, mu, lambda)                              EVDMaxLikelyFit of one block's maxima
      pvalue(score, mu, lambda)

    Returns a list of (evd_rc, mu, lambda, hss list with p-values) per block.  --stop-early's verdict only needs the
    gathered row (the count of samples above the best native score only grows along the loop, score.c:1036-1042)."""
    import numpy as np

    lo, hi = sample_range(n_samples, rank, world)
    local, native = score_range(lo, hi)
    full = np.asarray(gather(local))
    assert full.shape[1] == n_samples
    out = []
    for blk, hss in enumerate(native):
        if hss is None:                       # block the driver skips
            out.append(None)
            continue
        row = full[blk]
        best = max([h["score"] for h in hss], default=-1.0)
        rc, mu, lam = -1, 0.0, 0.0
        stopped = stop_early and int((row > np.float32(best)).sum()) > int(np.float32(cutoff) * np.float32(n_samples))
        if not stopped:
            ok, mu, lam = fit(row.astype(np.float64))
            rc = 1 if ok == 1 else -1
        mu32, lam32 = float(np.float32(mu)), float(np.float32(lam))    # *parMu = mu, score.c:1051-1052
        scored = [dict(h, pvalue=(pvalue(h["score"], mu32, lam32) if rc == 1 else 99.0)) for h in hss]
        out.append((rc, mu32, lam32, scored))
    return out


def all_gather_columns(local, widths: Sequence[int]):
    """All-gather of per-rank [n_blocks, w_r] sample slices into [n_blocks, sum w_r] (samples are the ragged
    dimension here): one all_gather_into_tensor on the transposed slices."""
    t = all_gather_rows(local.t().contiguous(), widths)
    return t.t().contiguous()


def hip_sample_sharded(ctx, blocks, rank: int, world: int, gather, **params):
    """score_sample_sharded on the HIP path: this rank's sample range as one batch with seed_base + lo."""
    import numpy as np

    from . import api

    n = params.get("sampleN", 100)
    base = params.get("seed_base", 42)

    def score_range(lo, hi):
        kw = dict(params, sampleN=max(1, hi - lo), seed_base=base + lo, stopEarly=0)
        batch = api.Batch(ctx, blocks, api.default_params(**kw)).run()
        mx = batch.maxScores_all()[:, :hi - lo]
        native = [batch.scoreAln(i) if batch.status(i) == api.RC_OK else None for i in range(batch.n)]
        batch.close()
        return np.ascontiguousarray(mx), native

    return score_sample_sharded(score_range, ctx.evd_fit, api.pvalue, gather, n, rank, world,
                                stop_early=bool(params.get("stopEarly", 0)), cutoff=params.get("cutoff", 1.0))
