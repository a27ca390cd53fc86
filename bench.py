#!/usr/bin/env python3
"""bench.py -- alignment-columns x samples scored per second on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (MT19937 streams -> native scoring -> null sampling ->
EVD fit) over one batch of synthetic alignment blocks already resident in HBM.  Workload at
N=1: BASELINE.json configs[2], the configuration the north_star target is quoted on
("10k-block MAF stream at num-samples=1000": 10 000 blocks x 120 cols x 6 seqs, n=1000) --
configs[1] (genomic-preprocessed.maf, 33 scored blocks) is a parity case (tests/), it cannot
fill one GPU.  With --gpus N every rank scores its own 10 000-block shard (weak scaling, no
data-path collective) and the per-sample maxima are exchanged by one RCCL all-gather.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
VALU_PEAK_GOPS = 78643.2       # 256 CU x 4 SIMD x 64 lanes / 2 cycles x 2.4 GHz (v_add/v_max wave64 = 2 cyc/SIMD)


def cells(L: int) -> int:
    """#{(b,i): 1<=b, i=b+2+3t<=L}."""
    return sum(m // 3 + 1 for m in range(0, L - 2))


def algorithmic_work(blocks, sampleN):
    """Per launch of the null kernel: bytes B = N*cols + 4 and flops F = 2*cells(L)*(6(N-1)+5)
    per sample-alignment, summed over the batch."""
    by = fl = 0
    for b in blocks:
        by += (b.n * b.cols + 4) * sampleN
        fl += 2 * cells(b.ref_len) * (6 * (b.n - 1) + 5) * sampleN
    return by, fl


def measured_traffic():
    """HBM-side bytes per k_null launch from the committed rocprofv3 PMC passes of this same
    command (profiles/collect.sh -> profiles/r01/k_null_pmc.json): FETCH_SIZE and WRITE_SIZE are in
    KiB, collected in separate passes; on gfx950 FETCH_SIZE under-reports streaming reads by 2x
    (MI355X_MICROARCH.md, HBM), so the read side is doubled.  None if no profile is committed."""
    path = os.path.join(ROOT, "profiles", "r01", "k_null_pmc.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


def cpu_baseline(blocks, sampleN, seed, target_s=12.0):
    """The unmodified reference objects (oracle/_ref/ref_harness, kind "reference") on a bounded
    sample of the same workload, single thread (the reference has no threading).  The sample is sized
    from a two-block probe so that it takes about target_s seconds on whatever host this runs on.
    Falls back to the oracle restatement (kind "port") only if the harness binary did not travel."""
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    from rnacode_amd.alnio import write_harness_blocks

    def run_reference(sub):
        with tempfile.NamedTemporaryFile("w", suffix=".blk", delete=False) as fh:
            path = fh.name
        try:
            write_harness_blocks(sub, path)
            t0 = time.perf_counter()
            subprocess.check_output([harness, path, "--samples", str(sampleN), "--seed", str(seed), "--no-backtrack"])
            return time.perf_counter() - t0
        finally:
            os.unlink(path)

    def run_port(sub):
        from oracle import binding as ob
        t0 = time.perf_counter()
        for b in sub:
            ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length,
                         b.tree, b.kappa, ob.default_params(sampleN), seed)
        return time.perf_counter() - t0

    kind, run = ("reference", run_reference) if os.path.exists(harness) else ("port", run_port)
    probe = run(blocks[:2]) / 2.0                                    # seconds per block on this host
    nblk = max(2, min(len(blocks), int(target_s / max(probe, 1e-3))))
    sub = blocks[:nblk]
    dt = run(sub)
    units = sum(b.cols for b in sub) * sampleN
    return dict(value=units / dt, unit="alignment-columns*samples/s", cores=1, kind=kind,
                sample=f"first {nblk} blocks of the workload at n={sampleN} (tree+kappa given), {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=10000)
    ap.add_argument("--cols", type=int, default=120)
    ap.add_argument("--seqs", type=int, default=6)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--seed-base", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", choices=("maxima", "fits"), default="maxima",
                    help="what the ranks all-gather per step: the per-sample maxima (north_star's exchange; 4 B x samples per "
                         "block, written by the kernel straight into the gathered tensor) or only the per-block fit results "
                         "(16 B per block: enough for p-values, since every rank fits its own blocks on the device)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --blocks per GPU (default); strong: --blocks in total, split across ranks (configs[3])")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    coll_dev = f"cuda:{local_rank}"
    # test hook: RC_BENCH_FORCE_DIST=1 takes the multi-rank code path with a world of one, so that a box with a
    # single GPU still runs init_process_group("nccl"), the all-gather and the reductions through RCCL
    multi = world > 1 or bool(os.environ.get("RC_BENCH_FORCE_DIST"))
    if multi:
        import torch
        import torch.distributed as dist
        if os.environ.get("RC_BENCH_TEST_BACKEND") == "gloo":
            # test hook: exercises this file's multi-rank logic on a box with fewer GPUs than ranks
            # (collectives on CPU tensors over gloo, ranks share the GPUs that exist); never used for numbers
            local_rank %= max(1, torch.cuda.device_count())
            coll_dev = "cpu"
            dist.init_process_group("gloo")
        else:
            ndev = torch.cuda.device_count()
            if 0 < ndev <= local_rank:   # a launcher that masks the GPUs per rank (each process sees its own as device 0)
                local_rank %= ndev
                coll_dev = f"cuda:{local_rank}"
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from rnacode_amd import api
    from rnacode_amd import distributed as rd
    from rnacode_amd.synth import synth_blocks

    t0 = time.perf_counter()
    if args.scaling == "weak":   # this rank's own shard: generator seed 1 + rank (generator seed 1)
        blocks = [b.upper() for b in synth_blocks(args.blocks, args.seqs, args.cols, seed=1 + rank)]
        counts = [args.blocks] * world
    else:                        # one stream of --blocks blocks, contiguous shards
        lo, hi = rd.shard_range(args.blocks, rank, world)
        blocks = [b.upper() for b in synth_blocks(args.blocks, args.seqs, args.cols, seed=1)[lo:hi]]
        counts = [rd.shard_range(args.blocks, r, world)[1] - rd.shard_range(args.blocks, r, world)[0] for r in range(world)]
    t_gen = time.perf_counter() - t0

    ctx = api.Context(local_rank)
    api.lib().rc_set_stream_cache(0)   # the MT19937 streams are part of every step
    params = api.default_params(sampleN=args.samples, seed_base=args.seed_base)
    t0 = time.perf_counter()
    batch = api.Batch(ctx, blocks, params)   # host prep (getModels etc.) + upload: inputs now resident in HBM
    t_prep = time.perf_counter() - t0

    gathered = None
    mine = None
    if multi and args.exchange == "maxima":
        mine = torch.empty((len(blocks), args.samples), dtype=torch.float32, device=f"cuda:{local_rank}")
        batch.bind_maxima(mine.data_ptr())

    def step():
        batch.run()                      # synchronous: kernels done, small results on the host
        if multi:                        # the path's one exchange (RCCL over xGMI): results of all blocks on every rank
            nonlocal gathered
            if args.exchange == "maxima":
                gathered = rd.all_gather_maxima(mine if coll_dev != "cpu" else mine.cpu(), counts)
            else:
                fits = torch.from_numpy(batch.fits()).to(coll_dev, non_blocking=True)
                gathered = rd.all_gather_fits(fits, counts)

    def sync():
        if multi:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    tk = {"total": 0.0, "mt_stream": 0.0, "null": 0.0, "native": 0.0, "evd_fit": 0.0}
    nlaunch = {}
    for _ in range(args.steps):
        step()
        t, nl = batch.timing()
        for k in tk:
            tk[k] += t[k]
        nlaunch = nl
    sync()
    elapsed = time.perf_counter() - t0
    if multi:
        elapsed = rd.max_over_ranks(elapsed, device=coll_dev)

    sa, cs = batch.work()
    if multi:   # units of all ranks (shards may differ by one block under strong scaling)
        tot = torch.tensor([float(cs), float(sa)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tot)
        cs, sa = int(tot[0].item()), int(tot[1].item())
    total_units = cs * args.steps
    value = total_units / elapsed
    out = None
    if rank == 0:
        by, fl = algorithmic_work(blocks, args.samples)
        null_ms = tk["null"] / args.steps / max(1, nlaunch.get("null", 1))   # average duration of one k_null launch
        ach = by / (null_ms * 1e-3) / 1e9
        out = {
            "metric": "alignment-columns x samples scored /sec",
            "value": value,
            "unit": "alignment-columns*samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"synthetic MAF {args.blocks} blocks x {args.cols} cols x {args.seqs} seqs, "
                                   f"--num-samples {args.samples}, " + ("per GPU" if args.scaling == "weak" else "in total, split across the GPUs")
                                   + " (BASELINE.json configs[2]/[3])",
                       "blocks_per_gpu": len(blocks), "cols": args.cols, "seqs": args.seqs, "samples": args.samples,
                       "seed_base": args.seed_base, "sharding": f"blocks x{world}",
                       "exchange": ("all-gather of " + args.exchange) if multi else "none"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": measured_traffic() if (args.blocks, args.cols, args.seqs, args.samples) == (10000, 120, 6, 1000) else None, "kernel": "k_null<%d>" % (args.seqs - 1), "kernel_ms": null_ms,
                         "algorithmic_bytes_per_launch": by,
                         "valu": {"achieved_gops": fl / (null_ms * 1e-3) / 1e9, "peak_gops": VALU_PEAK_GOPS,
                                  "frac": fl / (null_ms * 1e-3) / 1e9 / VALU_PEAK_GOPS,
                                  "algorithmic_flops_per_launch": fl}},
            "device_ms_per_step": {k: v / args.steps for k, v in tk.items()},
            "device_ms_note": "HIP-event intervals; the native-block kernels run on a second stream beside k_null, so "
                              "'native' is the interval until they are done (they fill k_null's tail), not their busy time "
                              "(1.8 ms when run alone, profiles/r01)",
            "host_prep_s": {"generate": t_gen, "models_and_upload": t_prep},
            "sample_alignments_per_s": sa * args.steps / elapsed,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(blocks, args.samples, args.seed_base)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    batch.close()
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
