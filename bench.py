#!/usr/bin/env python3
"""bench.py -- alignment-columns x samples scored per second on MI355X (BASELINE.json metric).

Workload: BASELINE.json configs[2]/[3], the configuration the north_star target is quoted on: a stream of
10 000 synthetic MAF blocks x 120 cols x 6 seqs at --num-samples 1000.  (configs[1],
genomic-preprocessed.maf with 33 scored blocks, is a parity case in tests/: it cannot fill one GPU.)
`--config c5` switches to a slice of configs[4] (12 seqs x 300 cols, n = 10 000: the long-block stress).

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment this file starts its own N ranks (a fresh
`python -m torch.distributed.run` child, before anything touches a GPU) and relays rank 0's line; under the
driver's torchrun it just runs as one rank.  The 10 000 blocks are split across the ranks (strong scaling =
configs[3]; `--scaling weak` gives every rank its own 10 000), no data-path collective, and every step ends
with the path's one exchange: an RCCL all-gather of the per-sample maxima.

With one GPU and no flags the same JSON line also carries four further legs, each with a timed region of about three seconds
(`--no-extra` leaves them out): `end_to_end` -- the native driver rnacode_amd/rnacode_hip as a fresh process on a 10 000-block
MAF file, wall time from exec to exit, trees fitted on the GPU (no sidecar): SURVEY.md section 8d's metric with I/O included
(RNAcode.c:115-228); `shard` -- the 1250 blocks ONE of eight ranks holds under configs[3], resident and streamed with an eighth
of the host's CPUs; `c5` -- the long-block slice of configs[4] with a roofline object of its own; `mixed` -- a stream of
10 000 blocks of 3..12 rows and 60..200 columns (ten row-count classes).

A step is one pass of the hot path over this rank's blocks.  Two measurements per run, K steps each:
  * resident (`value`, the contract's definition: inputs in HBM when the timed region starts): MT19937
    streams -> native scoring -> null sampling -> EVD fit on a resident batch, then the all-gather;
  * stream (`stream.value`): from host rc_block arrays to fitted results on the host -- block tables prepared
    by host threads, one pinned copy per sub-batch, the expected-score tables made on the device, kernels,
    results back -- as a pipeline of sub-batches (rc_stream_*), then the all-gather.  This is SURVEY.md
    section 8d's "native scoring and fit included" rate; file parsing is not part of it.
Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before torch starts the HIP runtime: see rnacode_amd/api.py

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
VALU_PEAK_GOPS = 78643.2       # 256 CU x 4 SIMD x 64 lanes / 2 cycles x 2.4 GHz (v_add/v_max wave64 = 2 cyc/SIMD)
PROFILE_ROUND = "r03"

CONFIGS = {
    # name: (blocks, cols, seqs, samples, description)
    "c3": (10000, 120, 6, 1000, "BASELINE.json configs[2]/[3]"),
    "c5": (1000, 300, 12, 10000, "a 1000-block slice of BASELINE.json configs[4] (100k blocks over 8 GPUs = 12 500 per GPU)"),
}


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


def measured_traffic(config, kernel):
    """HBM-side bytes per k_null launch from the committed rocprofv3 PMC passes of this same command
    (profiles/collect.sh -> profiles/<round>/k_null_<config>_pmc.json): FETCH_SIZE and WRITE_SIZE are in KiB,
    collected in separate passes; on gfx950 FETCH_SIZE under-reports streaming reads by 2x
    (MI355X_MICROARCH.md, HBM), so the read side is doubled.  `kernel` is the instantiation this run just timed
    (rc_batch_null_kernel): a profile of another kernel is refused.  (None, reason) if no usable profile is committed."""
    why = "no committed profile"
    for rnd in (PROFILE_ROUND, "r02", "r01"):
        for name in (f"k_null_{config}_pmc.json", "k_null_pmc.json" if config == "c3" else None):
            if not name:
                continue
            try:
                with open(os.path.join(ROOT, "profiles", rnd, name)) as fh:
                    d = json.load(fh)
                prof_kernel = (d.get("kernel") or "").split("(")[0].replace("void ", "").strip()
                if prof_kernel != kernel:
                    why = f"profiles/{rnd}/{name} is a profile of {prof_kernel!r}, this run timed {kernel!r}: refused"
                    continue
                return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0, f"profiles/{rnd}/{name}"
            except Exception:
                continue
    return None, why


def roofline_object(blocks, sampleN, null_ms, kernel, config, with_traffic):
    """The contract's roofline object for one k_null launch of `null_ms` ms over `blocks` (algorithmic work: SURVEY.md section 8d)."""
    by, fl = algorithmic_work(blocks, sampleN)
    ach = by / (null_ms * 1e-3) / 1e9
    traffic, src = measured_traffic(config, kernel) if with_traffic else (None, "not the profiled configuration")
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "kernel": kernel, "kernel_ms": null_ms, "algorithmic_bytes_per_launch": by,
            "valu": {"achieved_gops": fl / (null_ms * 1e-3) / 1e9, "peak_gops": VALU_PEAK_GOPS,
                     "frac": fl / (null_ms * 1e-3) / 1e9 / VALU_PEAK_GOPS, "algorithmic_flops_per_launch": fl}}


def end_to_end_leg(blocks, sampleN, seed, runs=5):
    """SURVEY.md section 8d's metric as the reference's driver would be timed (RNAcode.c:115-228: read -> tree -> models -> score ->
    sample -> print): the native stand-alone driver as a fresh process on a MAF file, trees and kappa fitted on the GPU (no sidecar),
    the tabular listing read from its standard output.  Wall time from exec to the LAST OUTPUT BYTE (end of file on the pipe: the
    driver closes its output when the listing is complete; the kernel then takes another 0.1 s to release the process's GPU
    resources, reported as wall_s_to_exit).  Runs before this process touches a GPU; a short pause separates the runs."""
    from rnacode_amd.synth import to_maf
    exe = os.path.join(ROOT, "rnacode_amd", "rnacode_hip")
    if not os.path.exists(exe):
        return {"error": "rnacode_amd/rnacode_hip is not built"}
    d = tempfile.mkdtemp(prefix="rc_e2e_")
    maf = os.path.join(d, "in.maf")
    try:
        with open(maf, "w") as fh:
            fh.write(to_maf(blocks))
        units = sum(b.cols for b in blocks) * sampleN
        walls, exits, stages, hss_lines = [], [], [], 0
        for _ in range(runs):
            time.sleep(0.3)
            t0 = time.perf_counter()
            p = subprocess.Popen([exe, maf, "-n", str(sampleN), "-t", "--seed-base", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                 env=dict(os.environ, RC_CLI_TIMES="1"))
            out = p.stdout.read()                      # until the driver closes its output
            walls.append(time.perf_counter() - t0)
            err = p.stderr.read().decode(errors="replace")
            rc = p.wait()
            exits.append(time.perf_counter() - t0)
            if rc != 0:
                return {"error": f"rnacode_hip exited with {rc}: {err.strip()[-300:]}"}
            hss_lines = out.count(b"\n")
            line = [ln for ln in err.splitlines() if ln.startswith("[rnacode_hip]")]
            stages.append(line[-1][len("[rnacode_hip] "):] if line else "")
        order = sorted(range(runs), key=lambda i: walls[i])
        med = order[runs // 2]
        return {"value": units / walls[med], "unit": "alignment-columns*samples/s", "wall_s_median": walls[med], "wall_s_best": walls[order[0]],
                "wall_s_all": walls, "wall_s_to_exit_all": exits, "stages_of_median_run": stages[med], "stages_of_best_run": stages[order[0]],
                "hss_lines": hss_lines, "maf_bytes": os.path.getsize(maf), "blocks": len(blocks), "samples": sampleN, "runs": runs,
                "includes": "process start, library load, HIP context, MAF parsing, tree + kappa fits on the GPU (k_tree_fit, no sidecar), block "
                            "preparation, scoring, sampling, fits, p-values, the tabular listing written to standard output and read by this process",
                "note": "fresh process per run, 0.3 s apart; wall = exec to end of the listing on the pipe; value = units / median wall"}
    finally:
        if os.path.exists(maf):
            os.unlink(maf)
        os.rmdir(d)


LAST_STEP_MS = []   # per-step times of the last timed_region (diagnostics: a mean hides a stalled step)


def timed_region(step, min_s, min_steps=3):
    """Run `step` until at least min_s seconds and min_steps steps have passed; (seconds, steps)."""
    n = 0
    del LAST_STEP_MS[:]
    gc.collect()
    gc.disable()   # this process holds millions of Python objects (the synthetic blocks): a full collection in the middle of a step takes 0.3-2 s
    try:
        t0 = time.perf_counter()
        prev = t0
        while True:
            step()
            n += 1
            now = time.perf_counter()
            LAST_STEP_MS.append((now - prev) * 1e3)
            prev = now
            dt = now - t0
            if dt >= min_s and n >= min_steps:
                return dt, n
    finally:
        gc.enable()


def spread():
    """min / median / max of the last timed region's steps, ms"""
    x = sorted(LAST_STEP_MS)
    return {"min": x[0], "median": x[len(x) // 2], "max": x[-1]} if x else None


def resident_leg(api, ctx, blocks, params, min_s, warmup=2):
    """Resident steps of `blocks` for about min_s seconds: (ms per step, average k_null launch ms, kernel name, units per step, batch timing)."""
    m = api.Marshalled(blocks)
    m.set_trees()
    batch = api.Batch(ctx, m, params)
    for _ in range(warmup):
        batch.run()
    acc = {"null": 0.0, "n": 0, "launches": 1}

    def step():
        batch.run()
        t, nl = batch.timing()
        acc["null"] += t["null"]; acc["n"] += 1; acc["launches"] = max(1, nl.get("null", 1))

    dt, n = timed_region(step, min_s)
    kernel = batch.null_kernel()
    _, cs = batch.work()
    batch.close()
    resident_leg.spread = spread()
    resident_leg.spread["mean"] = dt / n * 1e3
    # the MEDIAN step (a stalled step -- before the collector was switched off around the timed regions, single steps took 0.3-2 s
    # now and then -- would shift a three-second mean by tens of per cent; min / median / max / mean are reported)
    return resident_leg.spread["median"], acc["null"] / acc["n"] / acc["launches"], kernel, cs, n, m


def stream_leg(api, ctx, m, params, sub_blocks, min_s, depth=3, warmup=2):
    """Streamed passes over the Marshalled blocks `m` for about min_s seconds: (ms per pass, passes).  sub_blocks: a list of sizes."""
    st = api.Stream(ctx, params, depth)

    def step():
        for sb in api.score_stream(ctx, m, params, sub_blocks, stream=st):
            sb.fits()
            sb.close()

    for _ in range(warmup):
        step()
    dt, n = timed_region(step, min_s)
    st.close()
    stream_leg.spread = spread()
    stream_leg.spread["mean"] = dt / n * 1e3
    return stream_leg.spread["median"], n   # the median pass, see resident_leg


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
    probe_n = min(sampleN, 1000)
    t0 = time.perf_counter()
    if kind == "reference":   # probe at a bounded sample count (C5: n = 10 000 would take a minute per block)
        with tempfile.NamedTemporaryFile("w", suffix=".blk", delete=False) as fh:
            path = fh.name
        try:
            write_harness_blocks(blocks[:2], path)
            subprocess.check_output([harness, path, "--samples", str(probe_n), "--seed", str(seed), "--no-backtrack"])
        finally:
            os.unlink(path)
        probe = (time.perf_counter() - t0) / 2.0 * (sampleN / probe_n)
    else:
        probe = run(blocks[:2]) / 2.0
    nblk = max(1, min(len(blocks), int(target_s / max(probe, 1e-3))))
    sub = blocks[:nblk]
    dt = run(sub)
    units = sum(b.cols for b in sub) * sampleN
    return dict(value=units / dt, unit="alignment-columns*samples/s", cores=1, kind=kind,
                sample=f"first {nblk} blocks of the workload at n={sampleN}, {dt:.1f} s; tree and kappa given to the harness, "
                       f"so PhyML's share (1-2 % of the reference's run time, SURVEY.md section 6) is not in it")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--blocks", type=int)
    ap.add_argument("--cols", type=int)
    ap.add_argument("--seqs", type=int)
    ap.add_argument("--samples", type=int)
    ap.add_argument("--seed-base", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the host-inclusive stream measurement")
    ap.add_argument("--sub-batches", type=int, default=0,
                    help="stream mode: equal sub-batches per rank and pass (0: the library's schedule, rc_stream_plan)")
    ap.add_argument("--depth", type=int, default=3, help="stream mode: batches in flight")
    ap.add_argument("--host-threads", type=int, default=0, help="block preparation threads per rank (0: this rank's share of the usable CPUs)")
    ap.add_argument("--exchange", choices=("maxima", "fits"), default="maxima",
                    help="what the ranks all-gather per step: the per-sample maxima (north_star's exchange; 4 B x samples per "
                         "block, written by the kernel straight into the gathered tensor) or only the per-block fit results "
                         "(16 B per block: enough for p-values, since every rank fits its own blocks on the device)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: wait for every step's all-gather before the next step starts (default: step i's gather runs over "
                         "xGMI while step i + 1 is scored; all of them are complete when the timed region ends)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default): --blocks in total, split across the ranks (BASELINE configs[3]); weak: --blocks per GPU")
    ap.add_argument("--no-weak", action="store_true", help="N > 1, strong scaling: skip the extra weak-scaling measurement")
    ap.add_argument("--dump-gathered", help="rank 0 writes the gathered per-sample maxima of the last resident step here (.npy; tests)")
    ap.add_argument("--no-extra", action="store_true", help="one GPU: skip the end_to_end / shard / c5 / mixed legs")
    ap.add_argument("--extra-seconds", type=float, default=3.0, help="length of the timed region of each extra leg")
    a = ap.parse_args()
    blocks, cols, seqs, samples, _ = CONFIGS[a.config]
    a.blocks = a.blocks or blocks
    a.cols = a.cols or cols
    a.seqs = a.seqs or seqs
    a.samples = a.samples or samples
    return a


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # start the ranks as a fresh child before this process touches a GPU (never re-exec a process that has)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without torchrun)")
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

    import numpy as np

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

    # one GPU, the default workload: the legs that start child processes run first, before this process touches a GPU
    single = world == 1 and not multi
    default_workload = (args.blocks, args.cols, args.seqs, args.samples) == CONFIGS[args.config][:4]
    extras = single and not args.no_extra and default_workload and args.config == "c3"
    cpu = None
    if single and not args.no_cpu_baseline:
        cpu = cpu_baseline(blocks, args.samples, args.seed_base)
    legs = {}
    if extras:
        legs["end_to_end"] = end_to_end_leg(blocks, args.samples, args.seed_base)

    ctx = api.Context(local_rank)
    cpus = api.lib().rc_host_cpus()
    threads = args.host_threads or max(1, min(32, cpus // max(1, local_world)))
    ctx.set_host_threads(threads)
    api.lib().rc_set_stream_cache(0)   # the MT19937 streams are part of every step
    params = api.default_params(sampleN=args.samples, seed_base=args.seed_base)
    marshalled = api.Marshalled(blocks)   # the rc_block arrays a C caller would hold: the stream measurement starts from these
    marshalled.set_trees()
    t0 = time.perf_counter()
    batch = api.Batch(ctx, marshalled, params)   # host prep + upload + table kernels: inputs now resident in HBM
    t_prep = time.perf_counter() - t0

    gathered = None
    mine = None
    # The exchange overlapped with the next step (equal shards, maxima, RCCL): two source tensors alternate, step i's all-gather is
    # started without waiting for it, and the tensor is written again (step i + 2) only after that gather has read it.
    overlap = None
    step_no = 0
    if multi and args.exchange == "maxima":
        if not args.no_overlap and coll_dev != "cpu" and len(set(counts)) == 1:
            overlap = rd.OverlappedGather(len(blocks), args.samples, f"cuda:{local_rank}")
            mine = overlap.src[0]
        else:
            mine = torch.empty((len(blocks), args.samples), dtype=torch.float32, device=f"cuda:{local_rank}")
        batch.bind_maxima(mine.data_ptr())

    def exchange(fit_rows):
        """The path's one exchange (RCCL over xGMI): results of all blocks on every rank.  Synchronised before it
        returns: the next step overwrites the gathered buffer's source on the library's own streams."""
        nonlocal gathered
        if args.exchange == "maxima":
            gathered = rd.all_gather_maxima(mine if coll_dev != "cpu" else mine.cpu(), counts)
        else:
            fits = torch.from_numpy(fit_rows).to(coll_dev, non_blocking=True)
            gathered = rd.all_gather_fits(fits, counts)
        if coll_dev != "cpu":
            torch.cuda.current_stream().synchronize()

    def resident_step():
        nonlocal step_no
        if overlap is not None:
            batch.bind_maxima(overlap.source(step_no).data_ptr())   # (waits for the gather of two steps ago, long done)
            batch.run()                  # synchronous: this step's maxima are complete in the tensor
            overlap.launch(step_no)      # their all-gather runs while the next step is scored
            step_no += 1
            return
        batch.run()                      # synchronous: kernels done, small results on the host
        if multi:
            exchange(batch.fits() if args.exchange == "fits" else None)

    # stream mode: the library's schedule (rc_stream_plan: small first sub-batch, then doubling, whole rounds of the chip) unless told otherwise
    if args.sub_batches:
        sb = max(1, -(-len(blocks) // args.sub_batches))
        sub_sizes = [min(sb, len(blocks) - lo) for lo in range(0, len(blocks), sb)]
    else:
        sub_sizes = api.stream_plan(ctx, params, len(blocks), 1)
    stream = None if args.no_stream else api.Stream(ctx, params, args.depth)

    def stream_step():
        nonlocal step_no
        rows = []
        src = overlap.source(step_no) if overlap is not None else mine
        ptr = src.data_ptr() if (multi and args.exchange == "maxima") else 0
        for sb in api.score_stream(ctx, marshalled, params, sub_sizes, stream=stream, maxima_ptr=ptr):
            rows.append(sb.fits())       # (evd_rc, mu, lambda, better) per block: what a caller needs for p-values
            sb.close()                   # buffers back to the stream
        if overlap is not None:
            overlap.launch(step_no)
            step_no += 1
        elif multi:
            exchange(np.concatenate(rows) if args.exchange == "fits" else None)
        return rows

    def sync():
        nonlocal gathered
        if multi:
            if overlap is not None:          # every all-gather started so far is complete before the clock is read
                g = overlap.finish()
                if g is not None:
                    gathered = g
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    if multi:   # RCCL's first collective builds its rings: keep that out of the timed steps even with --warmup 0
        probe = torch.zeros((1, 4), dtype=torch.float32, device=coll_dev)
        rd.all_gather_rows(probe, [1] * world)
        sync()

    def timed(step, collect_timing):
        for _ in range(args.warmup):
            step()
        sync()
        gc.collect()
        gc.disable()   # see timed_region
        t0 = time.perf_counter()
        tk = {"total": 0.0, "mt_stream": 0.0, "null": 0.0, "native": 0.0, "evd_fit": 0.0}
        nlaunch = {}
        for _ in range(args.steps):
            step()
            if collect_timing:
                t, nl = batch.timing()
                for k in tk:
                    tk[k] += t[k]
                nlaunch = nl
        sync()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if multi:
            elapsed = rd.max_over_ranks(elapsed, device=coll_dev)
        return elapsed, tk, nlaunch

    elapsed, tk, nlaunch = timed(resident_step, True)
    if args.dump_gathered and rank == 0:
        np.save(args.dump_gathered, gathered.cpu().numpy() if multi else batch.maxScores_all())
    prep_host_ms, prep_kernels_ms, up_bytes = batch.prep_timing()
    s_elapsed = None
    c_elapsed = None
    if stream is not None:
        s_elapsed, _, _ = timed(stream_step, False)
        if not multi:
            # the same sub-batches as ONE stream over all the passes: the pipeline is filled once, not once per pass (what a long input sees)
            big = max(sub_sizes)   # after the fill every sub-batch can be a large one
            slices = [(lo, min(len(blocks), lo + big)) for lo in range(0, len(blocks), big)] * args.steps
            slices = [(0, sub_sizes[0])] + ([(sub_sizes[0], slices[0][1])] if slices[0][1] > sub_sizes[0] else []) + slices[1:]
            sent = 0
            gc.collect()
            gc.disable()
            t0 = time.perf_counter()
            while sent < len(slices) or stream.pending:
                while sent < len(slices) and stream.pending < stream.depth:
                    stream.submit(marshalled, slices[sent][0], slices[sent][1])
                    sent += 1
                sb = stream.next()
                sb.fits()
                sb.close()
            c_elapsed = time.perf_counter() - t0
            gc.enable()

    # N > 1 under strong scaling: also the weak-scaling figure (every rank its own --blocks blocks, generator seed 1 + rank),
    # as an extra key: the per-GPU work then equals the N = 1 run's, which is what separates kernel scaling from shard size
    weak = None
    if world > 1 and args.scaling == "strong" and not args.no_weak:
        wblocks = [b.upper() for b in synth_blocks(args.blocks, args.seqs, args.cols, seed=1 + rank)]
        wm = api.Marshalled(wblocks)
        wm.set_trees()
        wbatch = api.Batch(ctx, wm, params)
        wmine = None
        if args.exchange == "maxima":
            wmine = torch.empty((len(wblocks), args.samples), dtype=torch.float32, device=f"cuda:{local_rank}")
            wbatch.bind_maxima(wmine.data_ptr())
        wcounts = [args.blocks] * world

        def weak_step():
            wbatch.run()
            if args.exchange == "maxima":
                rd.all_gather_maxima(wmine if coll_dev != "cpu" else wmine.cpu(), wcounts)
            else:
                rd.all_gather_fits(torch.from_numpy(wbatch.fits()).to(coll_dev), wcounts)
            if coll_dev != "cpu":
                torch.cuda.current_stream().synchronize()

        w_elapsed, _, _ = timed(weak_step, False)
        _, wcs = wbatch.work()
        weak = {"value": wcs * world * args.steps / w_elapsed, "unit": "alignment-columns*samples/s", "ms_per_step": w_elapsed / args.steps * 1e3,
                "blocks_per_gpu": len(wblocks), "scaling": "weak"}
        wbatch.close()
        del wmine

    sa, cs = batch.work()
    if multi:   # units of all ranks (shards may differ by one block under strong scaling)
        tot = torch.tensor([float(cs), float(sa)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tot)
        cs, sa = int(tot[0].item()), int(tot[1].item())
    total_units = cs * args.steps
    value = total_units / elapsed
    kernel_name = batch.null_kernel()
    if extras:
        main_ms = elapsed / args.steps * 1e3
        # ---- shard: what ONE of eight ranks holds under configs[3] (strong scaling), with an eighth of the host's CPUs
        lo, hi = rd.shard_range(len(blocks), 0, 8)
        t8 = max(1, cpus // 8)
        ctx.set_host_threads(t8)
        sh_ms, sh_null, sh_kernel, sh_units, sh_steps, sh_m = resident_leg(api, ctx, blocks[lo:hi], params, args.extra_seconds)
        sh_sub = api.stream_plan(ctx, params, hi - lo, 1)
        sh_res_spread = resident_leg.spread
        sh_stream_ms, sh_passes = stream_leg(api, ctx, sh_m, params, sh_sub, args.extra_seconds)
        legs["shard"] = {
            "blocks": hi - lo, "of": len(blocks), "ranks": 8, "host_threads": t8, "usable_cpus": cpus,
            "ms_per_step": sh_ms, "ms_per_step_spread": sh_res_spread, "steps": sh_steps, "kernel_ms": sh_null, "ideal_ms": main_ms / 8,
            "efficiency": (main_ms / 8) / sh_ms, "value": sh_units / (sh_ms * 1e-3),
            "stream": {"ms_per_step": sh_stream_ms, "ms_per_step_spread": stream_leg.spread, "passes": sh_passes, "frac_of_resident": sh_ms / sh_stream_ms, "sub_batches": sh_sub,
                       "efficiency": (main_ms / 8) / sh_stream_ms},
            "note": "ms_per_step of the extra legs is the MEDIAN step of the timed region (min / median / max / mean in ms_per_step_spread); "
                    "efficiency = (ms per step of the whole 10 000-block workload on this GPU / 8) / ms per step of the shard: what strong "
                    "scaling to eight GPUs can reach before the all-gather; the stream prepares its blocks on usable_cpus // 8 host threads"}
        ctx.set_host_threads(threads)
        del sh_m
        # ---- mixed: ten row-count classes, ragged lengths, as a stream from host blocks
        import numpy as _np
        from rnacode_amd.synth import synth_block
        rng = _np.random.RandomState(5)
        t0 = time.perf_counter()
        mixed = [synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True).upper() for i in range(10000)]
        genm = time.perf_counter() - t0
        mx_ms, mx_null, mx_kernel, mx_units, mx_steps, mx_m = resident_leg(api, ctx, mixed, params, args.extra_seconds, warmup=2)
        mx_res_spread = resident_leg.spread
        classes = len({b.n for b in mixed})
        mx_sub = api.stream_plan(ctx, params, len(mixed), classes)
        # (three warm-up passes: the stream's three batch objects each grow their multi-GB staging areas to the largest sub-batch once)
        mx_stream_ms, mx_passes = stream_leg(api, ctx, mx_m, params, mx_sub, args.extra_seconds, warmup=3)
        legs["mixed"] = {"workload": "synthetic MAF 10000 blocks of 3..12 rows x 60..200 cols, --num-samples %d" % args.samples,
                         "row_count_classes": classes, "value": mx_units / (mx_stream_ms * 1e-3), "unit": "alignment-columns*samples/s",
                         "ms_per_step": mx_stream_ms, "passes": mx_passes, "blocks_per_sub_batch": mx_sub, "host_threads": threads,
                         "ms_per_step_spread": stream_leg.spread,
                         "resident": {"ms_per_step": mx_ms, "steps": mx_steps, "value": mx_units / (mx_ms * 1e-3), "ms_per_step_spread": mx_res_spread},
                         "generate_s": genm, "note": "value = the streamed rate (host rc_block arrays -> fitted results on the host), from the median pass"}
        del mx_m, mixed

        # ---- c5: the long-block slice of configs[4] (last: 0.6 s launches at the HBM roof; the legs after it ran a few per cent slower)
        nb5, cols5, seqs5, n5, desc5 = CONFIGS["c5"]
        t0 = time.perf_counter()
        blocks5 = [b.upper() for b in synth_blocks(nb5, seqs5, cols5, seed=1)]
        gen5 = time.perf_counter() - t0
        params5 = api.default_params(sampleN=n5, seed_base=args.seed_base)
        c5_ms, c5_null, c5_kernel, c5_units, c5_steps, c5_m = resident_leg(api, ctx, blocks5, params5, args.extra_seconds, warmup=1)
        legs["c5"] = {"workload": f"synthetic MAF {nb5} blocks x {cols5} cols x {seqs5} seqs, --num-samples {n5} ({desc5})",
                      "value": c5_units / (c5_ms * 1e-3), "unit": "alignment-columns*samples/s", "ms_per_step": c5_ms, "ms_per_step_spread": resident_leg.spread,
                      "steps": c5_steps,
                      "generate_s": gen5, "roofline": roofline_object(blocks5, n5, c5_null, c5_kernel, "c5", True)}
        del c5_m, blocks5
    if rank == 0:
        null_ms = tk["null"] / args.steps / max(1, nlaunch.get("null", 1))   # average duration of one k_null launch
        desc = CONFIGS[args.config][4]
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
                                   + f" ({desc})",
                       "blocks_per_gpu": len(blocks), "cols": args.cols, "seqs": args.seqs, "samples": args.samples,
                       "seed_base": args.seed_base, "sharding": f"blocks x{world}",
                       "exchange": (("all-gather of " + args.exchange + (", step i's gather overlapped with the scoring of step i + 1 (all complete "
                                     "inside the timed region)" if overlap is not None else "")) if multi else "none")},
            "value_definition": "resident: inputs in HBM when the timed region starts (bench contract); the host-inclusive pipeline is stream.value",
            "roofline": roofline_object(blocks, args.samples, null_ms, kernel_name, args.config, default_workload and world == 1),
            "device_ms_per_step": {k: v / args.steps for k, v in tk.items()},
            "device_ms_note": "HIP-event intervals; the native-block kernels run on a second stream beside k_null, so "
                              "'native' is the interval until they are done (they fill k_null's tail), not their busy time",
            "host_prep_s": {"generate": t_gen, "prepare_upload_tables": t_prep, "prepare_host_part": prep_host_ms * 1e-3,
                            "table_kernels": prep_kernels_ms * 1e-3, "uploaded_bytes": up_bytes, "host_threads": threads,
                            "usable_cpus": cpus},
            "sample_alignments_per_s": sa * args.steps / elapsed,
            # draws beyond the last cumulative frequency (float frequencies need not sum to 1): the reference reads out of bounds there
            # (seqgen/evolve.c:173), this path clamps to state 3 and counts -- the one documented divergence
            "reference_ub_draws_per_step": {"count": int(batch.clamped()), "draws": int(sum(int(b.cols) * (2 * int(b.n) - 2) for b in blocks)) * int(params.sampleN),
                                            "note": "sample s of every block reads the one MT19937 stream seeded seed_base + s, so blocks of equal shape compare the same draws with their own thresholds"},
        }
        out["value_resident"] = value
        if s_elapsed is not None:
            out["value_stream"] = total_units / s_elapsed
            out["stream"] = {
                "value": total_units / s_elapsed, "unit": "alignment-columns*samples/s", "ms_per_step": s_elapsed / args.steps * 1e3,
                "frac_of_resident": elapsed / s_elapsed, "sub_batches": len(sub_sizes), "blocks_per_sub_batch": sub_sizes,
                "depth": args.depth, "host_threads": threads, "usable_cpus": cpus,
                "continuous": None if c_elapsed is None else {
                    "ms_per_step": c_elapsed / args.steps * 1e3, "frac_of_resident": elapsed / c_elapsed,
                    "note": "all the passes as one stream: the pipeline fills once instead of once per pass"},
                "includes": "block preparation on host threads (getModels inputs, gap tables, tree thresholds), one pinned H2D copy per "
                            "sub-batch, expected-score tables on the device, MT19937 streams, native scoring, null sampling, EVD fit, "
                            "fit results back on the host" + (", all-gather" if multi else ""),
                "excludes": "file parsing and tree fitting (sidecar trees, as for the cpu_baseline)"}
        if weak is not None:
            out["weak_scaling"] = weak
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        for k, v in legs.items():
            out[k] = v
        print(json.dumps(out))
    if stream is not None:
        stream.close()
    batch.close()
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
