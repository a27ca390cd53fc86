"""world_size-2 gloo test of the multi-GPU plumbing on CPU: block sharding, the single
all-gather (of the per-sample maxima, bench.py's default exchange, and of the per-block fit
results), max-over-ranks timing.  The per-rank maxima come from the CPU
oracle (the HIP path cannot run here); what is under test is that sharding + gather reproduces
the single-process result bit for bit, i.e. the property bench.py --gpus N relies on."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nblocks, n, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from oracle import binding as ob
    from rnacode_amd import distributed as rd
    from rnacode_amd.synth import synth_blocks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = [b.upper() for b in synth_blocks(nblocks, 4, 36, seed=9)]
    mine = rd.shard_blocks(blocks, rank, world)
    counts = [rd.shard_range(nblocks, r, world)[1] - rd.shard_range(nblocks, r, world)[0] for r in range(world)]
    rows, fits = [], []
    for b in mine:
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length,
                           b.tree, b.kappa, ob.default_params(n), 42)
        rows.append(np.float32(res.maxScores))
        fits.append(np.float32([res.evd_rc, res.mu, res.lam, 0.0]))
    local = torch.from_numpy(np.stack(rows)) if rows else torch.empty((0, n), dtype=torch.float32)
    gathered = rd.all_gather_maxima(local, counts)
    gfits = rd.all_gather_fits(torch.from_numpy(np.stack(fits)) if fits else torch.empty((0, 4), dtype=torch.float32), counts)
    slowest = rd.max_over_ranks(1.0 + rank)
    if rank == 0:
        np.savez(out_path, maxima=gathered.numpy(), fits=gfits.numpy())
        assert slowest == float(world)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nblocks", [4, 5])   # equal and ragged shards
def test_sharded_gather_equals_single_process(tmp_path, nblocks):
    import torch.multiprocessing as mp

    from oracle import binding as ob
    from rnacode_amd.synth import synth_blocks

    n, world = 24, 2
    out = str(tmp_path / "g.npz")
    mp.spawn(_worker, args=(world, _free_port(), nblocks, n, out), nprocs=world, join=True)
    got = np.load(out)
    blocks = [b.upper() for b in synth_blocks(nblocks, 4, 36, seed=9)]
    res = [ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree,
                        b.kappa, ob.default_params(n), 42) for b in blocks]
    np.testing.assert_array_equal(got["maxima"], np.stack([np.float32(r.maxScores) for r in res]))
    np.testing.assert_array_equal(got["fits"], np.stack([np.float32([r.evd_rc, r.mu, r.lam, 0.0]) for r in res]))


def test_shard_range_is_a_partition():
    from rnacode_amd import distributed as rd
    for n in (0, 1, 7, 10000):
        for world in (1, 2, 3, 8):
            ranges = [rd.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1


def _sample_worker(rank, world, port, n, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from oracle import binding as ob
    from rnacode_amd import distributed as rd
    from rnacode_amd.synth import synth_blocks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = [b.upper() for b in synth_blocks(3, 4, 36, seed=9)]
    widths = [rd.sample_range(n, r, world)[1] - rd.sample_range(n, r, world)[0] for r in range(world)]

    def score_range(lo, hi):
        rows, native = [], []
        for b in blocks:
            res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length,
                               b.tree, b.kappa, ob.default_params(max(1, hi - lo)), 42 + lo)
            rows.append(np.float32(res.maxScores)[:hi - lo])
            native.append(res.hss)
        return np.stack(rows), native

    def gather(local):
        return rd.all_gather_columns(torch.from_numpy(np.ascontiguousarray(local)), widths).numpy()

    res = rd.score_sample_sharded(score_range, ob.evd_fit, ob.pvalue, gather, n, rank, world)
    if rank == 0:
        import pickle
        with open(out_path, "wb") as fh:
            pickle.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [200, 64])   # 200: ranks get 128 + 72 samples; 64: the second rank has none
def test_sample_range_sharding_equals_single_process(tmp_path, n):
    """SURVEY section 8e's second partitioning (few blocks, many samples): each rank simulates its own sample range
    of every block, one all-gather, the fit runs on the gathered maxima."""
    import pickle

    import torch.multiprocessing as mp

    from oracle import binding as ob
    from rnacode_amd.synth import synth_blocks

    world = 2
    out = str(tmp_path / "s.pkl")
    mp.spawn(_sample_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    with open(out, "rb") as fh:
        got = pickle.load(fh)
    blocks = [b.upper() for b in synth_blocks(3, 4, 36, seed=9)]
    for g, b in zip(got, blocks):
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree,
                           b.kappa, ob.default_params(n), 42)
        rc, mu, lam, hss = g
        assert rc == res.evd_rc
        assert np.float32(mu) == np.float32(res.mu) and np.float32(lam) == np.float32(res.lam)
        assert [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"]), np.float32(h["pvalue"])) for h in hss] == \
               [(h["strand"], h["frame"], h["start"], h["end"], np.float32(h["score"]), np.float32(h["pvalue"])) for h in res.hss]


def test_sample_range_is_a_partition_in_wavefront_groups():
    from rnacode_amd import distributed as rd
    for n in (1, 63, 64, 65, 100, 1000, 10000):
        for world in (1, 2, 3, 8):
            r = [rd.sample_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert all(lo % 64 == 0 for lo, _ in r if lo < n)


def _overlap_worker(rank, world, port, steps, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from rnacode_amd import distributed as rd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, width = 3, 5
    og = rd.OverlappedGather(rows, width, "cpu")
    seen = []
    for i in range(steps):
        src = og.source(i)                       # waits until gather i - 2 has read this buffer
        src.copy_(torch.full((rows, width), float(100 * i + rank)))   # "step i's kernels write their maxima"
        og.launch(i)
        if i >= 1:                               # gather i - 1 is complete once source(i + 1) has been handed out; check it a step late
            og._wait((i - 1) & 1)
            seen.append(og.out[(i - 1) & 1].clone())
    last = og.finish()
    seen.append(last.clone())
    if rank == 0:
        np.save(out_path, torch.stack(seen).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_alternates_two_buffers(tmp_path):
    """rnacode_amd.distributed.OverlappedGather (bench.py --gpus N: step i's all-gather runs while step i + 1 is scored): every
    step's gathered tensor holds that step's rows of every rank, in rank order, although only two source buffers exist."""
    import torch.multiprocessing as mp

    world, steps = 2, 5
    out = str(tmp_path / "og.npy")
    mp.spawn(_overlap_worker, args=(world, _free_port(), steps, out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape == (steps, world * 3, 5)
    for i in range(steps):
        for r in range(world):
            assert (got[i, 3 * r:3 * r + 3] == 100 * i + r).all()
