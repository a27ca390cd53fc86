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
