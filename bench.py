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

With one GPU and no flags the same JSON line also carries further legs, each with a timed region of about three seconds
(`--no-extra` leaves them out): `end_to_end` -- the native driver rnacode_amd/rnacode_hip as a fresh process on a 10 000-block
MAF file, wall time from exec to the last output byte, trees fitted on the GPU (no sidecar): SURVEY.md section 8d's metric with I/O
included (RNAcode.c:115-228) -- and `end_to_end_100k`, the same on a 100 000-block file (what the fixed costs amortise to);
`c1`, `c2` -- BASELINE.json configs[0] and [1], the two inputs the reference ships (tests/golden/*.json.gz hold them with the
reference's own trees): latency per file as one batch and block by block in the shim's call sequence; `shard` -- the 1250 blocks ONE
of eight ranks holds under configs[3], resident, streamed pass by pass and as one continuous stream, with an eighth of the host's
CPUs; `c5` -- the long-block slice of configs[4] with a roofline object of its own; `mixed` -- a stream of 10 000 blocks of 3..12
rows and 60..200 columns (ten row-count classes).  `cpu_baseline` (the unmodified reference objects, one core) has a sibling for
every leg, and `cpu_baseline_all_cores` runs one reference process per usable CPU over a split block list.

With N > 1 the line proves where the ranks sat (`n_ranks_seen`, `devices`: the UUIDs gathered from every rank), records RCCL's channel
cap (`config.rccl_env`), and says which order of the exchange ran the timed region (`config.exchange_ab`: overlapped with the next
step or waited for -- both are timed during the warm-up unless --overlap / --no-overlap forces one); `native_driver_multi_gpu` is the
native driver's own `--gpus N` mode on the whole input file, run by rank 0 while the others wait on the host.

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
PROFILE_ROUND = "r06"

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
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02", "r01"):
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
    """The contract's roofline object for one k_null launch of `null_ms` ms over `blocks` (algorithmic work: SURVEY.md section 8d).
    Two roofs are evaluated and `bound` names the one that binds: "valu" -- the FP32 add/max issue peak against the algorithmic
    flops -- or "hbm" -- the counter-measured HBM-side traffic against 8 TB/s.  The path has no dense contraction (no MFMA) and by the
    algorithmic bytes it is far from HBM (230-590 flop/B); it only becomes HBM-bound where a kernel re-reads its sigma codes from
    memory (the long-block shape).  The top-level achieved/peak/unit/frac describe the binding roof; the other one is the sub-object."""
    by, fl = algorithmic_work(blocks, sampleN)
    sec = null_ms * 1e-3
    ach_gbs = by / sec / 1e9
    traffic, src = measured_traffic(config, kernel) if with_traffic else (None, "not the profiled configuration")
    hbm = {"achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": by,
           "traffic": traffic, "traffic_source": src,
           "traffic_gbs": None if traffic is None else traffic / sec / 1e9,
           "traffic_frac_of_peak": None if traffic is None else traffic / sec / 1e9 / HBM_PEAK_GBS,
           "traffic_over_algorithmic": None if traffic is None else traffic / by}
    ach_tops = fl / sec / 1e12
    valu = {"achieved": ach_tops, "peak": VALU_PEAK_GOPS / 1e3, "unit": "TFLOP/s", "frac": ach_tops / (VALU_PEAK_GOPS / 1e3),
            "algorithmic_flops_per_launch": fl,
            "peak_definition": "FP32 add/max wave-instruction issue: 256 CU x 4 SIMD x 64 lanes / 2 cycles x 2.4 GHz (no FMA, no packed math in max-plus chains)"}
    hbm_binds = hbm["traffic_frac_of_peak"] is not None and hbm["traffic_frac_of_peak"] > valu["frac"]
    top = hbm if hbm_binds else valu
    return {"bound": "hbm" if hbm_binds else "valu", "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
            "traffic": traffic, "traffic_source": src, "kernel": kernel, "kernel_ms": null_ms,
            "binding_rule": "hbm if measured traffic / time / 8 TB/s exceeds the FP32 issue fraction, else valu",
            "hbm": hbm, "valu": valu}


def end_to_end_leg(blocks, sampleN, seed, runs=5, repeat=1, extra_args=()):
    """SURVEY.md section 8d's metric as the reference's driver would be timed (RNAcode.c:115-228: read -> tree -> models -> score ->
    sample -> print): the native stand-alone driver as a fresh process on a MAF file, trees and kappa fitted on the GPU (no sidecar),
    the tabular listing read from its standard output.  Wall time from exec to the LAST OUTPUT BYTE (end of file on the pipe: the
    driver closes its output when the listing is complete; the kernel then takes another 0.1 s to release the process's GPU
    resources, reported as wall_s_to_exit).  Runs before this process touches a GPU; a short pause separates the runs.
    repeat > 1: the file holds the blocks `repeat` times over (a 100 000-block file from the 10 000 generated blocks: what the fixed
    costs -- process start, HIP context, first tree chunk -- amount to on an input of configs[4]'s block count)."""
    from rnacode_amd.synth import to_maf
    exe = os.path.join(ROOT, "rnacode_amd", "rnacode_hip")
    if not os.path.exists(exe):
        return {"error": "rnacode_amd/rnacode_hip is not built"}
    d = tempfile.mkdtemp(prefix="rc_e2e_")
    maf = os.path.join(d, "in.maf")
    try:
        text = to_maf(blocks)
        with open(maf, "w") as fh:
            fh.write(text)
            body = text[text.index("\n") + 1:]      # without the "##maf" header line
            for _ in range(repeat - 1):
                fh.write(body)
        del text
        units = sum(b.cols for b in blocks) * sampleN * repeat
        walls, exits, stages, hss_lines = [], [], [], 0
        for _ in range(runs):
            time.sleep(0.3)
            t0 = time.perf_counter()
            p = subprocess.Popen([exe, maf, "-n", str(sampleN), "-t", "--seed-base", str(seed), *extra_args], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
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
                "hss_lines": hss_lines, "maf_bytes": os.path.getsize(maf), "blocks": len(blocks) * repeat, "samples": sampleN, "runs": runs,
                "includes": "process start, library load, HIP context, MAF parsing, tree + kappa fits on the GPU (k_tree_fit, no sidecar), block "
                            "preparation, scoring, sampling, fits, p-values, the tabular listing written to standard output and read by this process",
                "note": "fresh process per run, 0.3 s apart; wall = exec to end of the listing on the pipe; value = units / median wall"}
    finally:
        if os.path.exists(maf):
            os.unlink(maf)
        os.rmdir(d)


FILE_LEGS = {
    # BASELINE.json configs[0] and [1]: the two inputs the reference ships, scored with the trees and kappa its own PhyML step
    # gave (the sidecar fields of the committed goldens -- data, the reference itself is not read at run time)
    "c1": ("coding_aln_n100", "examples/coding.aln (ClustalW, 8 rows x 137 cols) --num-samples 100 (BASELINE.json configs[0])"),
    "c2": ("genomic_preprocessed_n1000", "examples/genomic-preprocessed.maf (34 blocks, 33 scored) --num-samples 1000 (BASELINE.json configs[1])"),
}


def golden_blocks(name):
    """(blocks, samples, seed_base, params) of a committed fixture under tests/golden/."""
    import gzip
    from rnacode_amd.alnio import AlnBlock, AlnRow
    with gzip.open(os.path.join(ROOT, "tests", "golden", name + ".json.gz"), "rb") as fh:
        doc = json.loads(fh.read().decode())
    blocks = []
    for e in doc["blocks"]:
        inp, ref = e["input"], e["ref"]
        rows = [AlnRow(r["name"], r["seq"], r["start"], r["length"], r["strand"], r["full_length"]) for r in inp["rows"]]
        blocks.append(AlnBlock(rows, inp["id"], ref.get("tree", inp.get("tree")), ref.get("kappa", inp.get("kappa"))).upper())
    return blocks, int(doc["samples"]), int(doc["seed_base"]), dict(doc.get("params") or {})


def file_leg(api, ctx, key, min_s=1.0):
    """Latency of one shipped input file, trees given: (a) as ONE batch through rc_batch_* (create = host preparation + upload + table
    kernels, run, HSS lists with p-values), (b) block by block in the order the reference's main loop makes its calls through the shim
    of integration/score_hip_shim.c (RNAcode.c:164-216: getModels -> scoreAln -> getExtremeValuePars per block, each block a batch of
    one).  The footer the reference prints (RNAcode.c:225-228) times the same span plus file parsing and PhyML."""
    name, desc = FILE_LEGS[key]
    blocks, n, seed, pk = golden_blocks(name)
    params = api.default_params(sampleN=n, seed_base=seed, **pk)
    scored = [b for b in blocks if b.tree is not None and b.n >= 3]
    units = sum(b.cols for b in scored) * n

    def as_batch():
        batch = api.Batch(ctx, blocks, params).run()
        for i in range(batch.n):
            if batch.status(i) == api.RC_OK:
                batch.scoreAln(i)
        batch.close()

    stages = {"create": 0.0, "run": 0.0, "results": 0.0, "close": 0.0, "device_null": 0.0, "device_native": 0.0, "device_fit": 0.0, "files": 0}

    def per_block():
        # (the stage sums: host wall time of create = prepare + upload + table kernels, run = every launch until the results are on the
        # host, results = HSS list + fit, close; and inside run the device's own intervals of the sampling launch, the native block's
        # kernels -- which run beside it -- and the fit)
        pc = time.perf_counter
        for b in scored:
            t0 = pc()
            batch = api.Batch(ctx, [b], params)      # make_batch (getModels, first call)
            batch.getModels(0)
            t1 = pc()
            batch.run()                              # scoreAln
            t2 = pc()
            batch.scoreAln(0)
            batch.getExtremeValuePars(0)             # getExtremeValuePars
            t3 = pc()
            tt, _ = batch.timing()
            batch.close()
            t4 = pc()
            stages["create"] += t1 - t0; stages["run"] += t2 - t1; stages["results"] += t3 - t2; stages["close"] += t4 - t3
            stages["device_null"] += tt["null"] * 1e-3; stages["device_native"] += tt["native"] * 1e-3; stages["device_fit"] += tt["evd_fit"] * 1e-3
        stages["files"] += 1

    out = {"workload": desc, "blocks": len(blocks), "scored_blocks": len(scored), "samples": n, "units": units, "unit": "alignment-columns*samples/s"}
    for label, fn in (("batch", as_batch), ("per_block_shim_sequence", per_block)):
        fn()
        dt, k = timed_region(fn, min_s, min_steps=200 if label == "batch" else 3)
        sp = spread()
        out[label] = {"ms_per_file": sp["median"], "ms_per_file_spread": sp, "files": k, "value": units / (sp["median"] * 1e-3)}
        if label == "per_block_shim_sequence" and stages["files"]:
            out[label]["stages_ms_per_file"] = {kk: vv / stages["files"] * 1e3 for kk, vv in stages.items() if kk != "files"}
    out["value"] = out["batch"]["value"]
    out["ms_per_file"] = out["batch"]["ms_per_file"]
    out["note"] = ("latency per file with a warm context, trees and kappa from the fixture (the reference's own PhyML results); value = "
                   "units / median latency of the one-batch form; these inputs cannot fill one GPU (33 blocks x 16 sample groups = 528 "
                   "work items on 4096 wavefront slots)")
    return out, scored, n, seed


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
    """min / median / p99 / max of the last timed region's steps (ms), and how many took more than ten times the median: a stalled
    step (round 4: 1.6 s among 5 ms ones, a hardware-queue oversubscription) must be visible, not averaged away"""
    x = sorted(LAST_STEP_MS)
    if not x:
        return None
    med = x[len(x) // 2]
    return {"min": x[0], "median": med, "p99": x[min(len(x) - 1, int(0.99 * len(x)))], "max": x[-1], "steps": len(x),
            "steps_over_10x_median": sum(1 for v in x if v > 10.0 * med)}


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
    resident_leg.null_interval_ms = acc["null"] / acc["n"]   # the whole sampling interval of a step (several class launches side by side)
    # the MEDIAN step (a stalled step -- before the collector was switched off around the timed regions, single steps took 0.3-2 s
    # now and then -- would shift a three-second mean by tens of per cent; min / median / max / mean are reported)
    return resident_leg.spread["median"], acc["null"] / acc["n"] / acc["launches"], kernel, cs, n, m


ROWS_LEG = (4, 8, 13, 17, 24, 30, 46, 64, 100)


def rows_leg(api, ctx, sampleN, seed, cols=150, target_ms=40.0):
    """Uniform resident batches of N rows x `cols` columns for N in ROWS_LEG, blocks enough for ~target_ms of sampling each: units/s, the
    sampling kernel that ran and its FP32 issue fraction by the algorithmic flops."""
    from rnacode_amd.synth import synth_blocks
    shapes = []
    p = api.default_params(sampleN=sampleN, seed_base=seed)
    for n in ROWS_LEG:
        base = [b.upper() for b in synth_blocks(32, n, cols, seed=7 + n)]
        _, fl = algorithmic_work(base, sampleN)
        want = max(256, min(8000, int(target_ms * 1e-3 * 0.3 * VALU_PEAK_GOPS * 1e9 / (fl / len(base)))))
        blocks = [base[i % len(base)] for i in range(want)]
        batch = api.Batch(ctx, blocks, p)
        batch.run()
        ts, nulls = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            batch.run()
            ts.append((time.perf_counter() - t0) * 1e3)
            nulls.append(batch.timing()[0]["null"])
        _, fl = algorithmic_work(blocks, sampleN)
        ms, null_ms = sorted(ts)[1], sorted(nulls)[1]
        shapes.append({"rows": n, "cols": cols, "blocks": len(blocks), "ms_per_step": ms, "sampling_ms": null_ms, "value": len(blocks) * cols * sampleN / (ms * 1e-3),
                       "issue_frac": fl / (null_ms * 1e-3) / (VALU_PEAK_GOPS * 1e9), "kernel": batch.null_kernel()})
        batch.close()
    return {"shapes": shapes, "unit": "alignment-columns*samples/s", "samples": sampleN,
            "note": "issue_frac = algorithmic flops of the batch / the sampling launches' interval / 78.6 T lane-ops/s; the full sweep (3..100 rows, 150 and "
                    "300 columns, registers and occupancy of every kernel) is tools/rows_sweep.py -> profiles/r06/"}


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


def continuous_leg(api, ctx, m, n_blocks, params, first_sizes, steady, min_s, depth=3):
    """The blocks of `m` submitted over and over as ONE stream for about min_s seconds: the first pass in the ramped sub-batches
    `first_sizes` (a cold pipeline), every later pass in sub-batches of `steady` blocks (the previous pass's last sub-batch is still in
    flight when the next pass starts, so nothing has to start early).  (ms per pass over the whole region, fill and drain included; passes)"""
    st = api.Stream(ctx, params, depth)
    first = []
    lo = 0
    for sz in first_sizes:
        first.append((lo, lo + sz)); lo += sz
    later = [(lo, min(n_blocks, lo + steady)) for lo in range(0, n_blocks, steady)]

    def run(seconds):
        sent_passes = 0
        queue = list(first)
        gc.collect()
        gc.disable()
        try:
            t0 = time.perf_counter()
            while True:
                if not queue:
                    if time.perf_counter() - t0 >= seconds and sent_passes >= 3:
                        break
                    queue = list(later)
                while queue and st.pending < st.depth:
                    a, b = queue.pop(0)
                    st.submit(m, a, b)
                    if not queue:
                        sent_passes += 1
                sb = st.next()
                sb.fits()
                sb.close()
            while st.pending:
                sb = st.next()
                sb.fits()
                sb.close()
            return time.perf_counter() - t0, sent_passes
        finally:
            gc.enable()

    run(0.2)                      # buffers grown, kernels loaded
    dt, passes = run(min_s)
    st.close()
    return dt / passes * 1e3, passes


def _harness_path():
    return os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def _run_reference(sub, n, seed):
    """Wall seconds of the unmodified reference objects (oracle/_ref/ref_harness) on the blocks `sub` at n samples, one thread."""
    from rnacode_amd.alnio import write_harness_blocks
    with tempfile.NamedTemporaryFile("w", suffix=".blk", delete=False) as fh:
        path = fh.name
    try:
        write_harness_blocks(sub, path)
        t0 = time.perf_counter()
        subprocess.check_output([_harness_path(), path, "--samples", str(n), "--seed", str(seed), "--no-backtrack", "--no-digest"])
        return time.perf_counter() - t0
    finally:
        os.unlink(path)


def _run_port(sub, n, seed):
    from oracle import binding as ob
    t0 = time.perf_counter()
    for b in sub:
        ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length,
                     b.tree, b.kappa, ob.default_params(n), seed)
    return time.perf_counter() - t0


def cpu_baseline(blocks, sampleN, seed, target_s=12.0, max_n=None, what="the workload"):
    """The unmodified reference objects (oracle/_ref/ref_harness, kind "reference") on a bounded
    sample of the same workload, single thread (the reference has no threading).  The sample is sized
    from a two-block probe so that it takes about target_s seconds on whatever host this runs on.
    max_n bounds the sample count of the run itself (C5's n = 10 000 takes a minute per block on one core): the rate
    is per (column x sample) and the null loop is linear in n (score.c:1004-1048), which `sample` then states.
    Falls back to the oracle restatement (kind "port") only if the harness binary did not travel."""
    kind, run = ("reference", _run_reference) if os.path.exists(_harness_path()) else ("port", _run_port)
    n_run = min(sampleN, max_n) if max_n else sampleN
    probe_n = min(n_run, 1000 if kind == "reference" else n_run)
    probe = run(blocks[:2], probe_n, seed) / 2.0 * (n_run / probe_n)
    nblk = max(1, min(len(blocks), int(target_s / max(probe, 1e-3))))
    sub = blocks[:nblk]
    dt = run(sub, n_run, seed)
    units = sum(b.cols for b in sub) * n_run
    cut = "" if n_run == sampleN else f" (of the workload's {sampleN}: the sampling loop is linear in n)"
    return dict(value=units / dt, unit="alignment-columns*samples/s", cores=1, kind=kind, blocks=nblk, seconds=dt,
                sample=f"first {nblk} blocks of {what} at n={n_run}{cut}, {dt:.1f} s; tree and kappa given to the harness, "
                       f"so PhyML's share (1-2 % of the reference's run time, SURVEY.md section 6) is not in it")


def cpu_baseline_all_cores(blocks, sampleN, seed, per_core_blocks, cores):
    """BASELINE.md section 3.3 / SURVEY.md section 8d(ii): the reference has no threads, but blocks are independent, so the fair
    many-core figure is one reference process per usable core over a split block list.  Core c gets blocks
    [c * per_core_blocks, (c + 1) * per_core_blocks) of the workload; wall time = start of the first process to the exit of the last."""
    from rnacode_amd.alnio import write_harness_blocks
    if not os.path.exists(_harness_path()):
        return {"error": "oracle/_ref/ref_harness did not travel: no all-cores reference baseline"}
    cores = max(1, min(cores, len(blocks) // max(1, per_core_blocks)))
    d = tempfile.mkdtemp(prefix="rc_cpu_")
    paths = []
    try:
        for c in range(cores):
            paths.append(os.path.join(d, f"part{c}.blk"))
            write_harness_blocks(blocks[c * per_core_blocks:(c + 1) * per_core_blocks], paths[-1])
        t0 = time.perf_counter()
        procs = [subprocess.Popen([_harness_path(), p, "--samples", str(sampleN), "--seed", str(seed), "--no-backtrack", "--no-digest"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for p in paths]
        rcs = [p.wait() for p in procs]
        dt = time.perf_counter() - t0
        if any(rcs):
            return {"error": f"a reference process exited with {max(rcs)}"}
        sub = blocks[:cores * per_core_blocks]
        units = sum(b.cols for b in sub) * sampleN
        return dict(value=units / dt, unit="alignment-columns*samples/s", cores=cores, kind="reference", blocks=len(sub), seconds=dt,
                    sample=f"{cores} reference processes side by side (one per usable CPU of this host), {per_core_blocks} blocks each = the first "
                           f"{len(sub)} blocks of the workload at n={sampleN}, {dt:.1f} s of wall time; tree and kappa given")
    finally:
        for p in paths:
            if os.path.exists(p):
                os.unlink(p)
        os.rmdir(d)


def _sig(x, digits=4):
    """floats to `digits` significant digits (the compact line), recursively"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(out):
    """The one line bench.py prints: at most ~1900 characters.  Keys of the bench contract unchanged; `legs` holds the other
    measurements as numbers only (DESIGN.md section 7 says what each is; the full record is in bench_full.json):
      stream / cont: ms per pass from host blocks to fitted results, pass by pass / as one continuous stream
      e2e, e2e100k, e2emix, e2emix100k, e2e_c5: the native driver as a fresh process, file in -> listing out (units/s; 10 000 blocks, 100 000, 10 000 mixed,
        100 000 mixed, 1000 blocks of 12 x 300 at n = 10 000)
      rows: uniform batches of N rows x 150 columns [N, fraction of the issue roof] (tools/rows_sweep.py is the whole axis)
      c1, c2: the shipped inputs as one batch [median ms, p99 ms, max ms, steps, steps over 10 x median]; c1pb, c2pb: block by block through the shim's call order
      shard: one of eight ranks' 1250 blocks [resident ms, efficiency vs a perfect eighth, streamed ms, continuous ms, valu frac]
      mixed: ten row-count classes [streamed ms, continuous ms, resident ms, resident units/s, valu frac]
      c5: 1000 blocks x 300 x 12 at n = 10 000 [ms, valu frac, traffic TB per launch]
      cpu16: the reference on every usable core [units/s, cores]"""
    def g(d, *path, default=None):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d

    def sp(leg, form):
        x = g(out, leg, form, "ms_per_file_spread") or {}
        return [x.get("median"), x.get("p99"), x.get("max"), x.get("steps"), x.get("steps_over_10x_median")]

    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out[k] for k in keep if k in out}
    cfg = out.get("config", {})
    line["config"] = {"workload": "synthetic MAF %s blocks x %s cols x %s seqs, n=%s (BASELINE configs[2]/[3])" % (
        g(out, "config", "blocks_per_gpu"), cfg.get("cols"), cfg.get("seqs"), cfg.get("samples")), "exchange": cfg.get("exchange", "none")[:40]}
    for k in ("rccl_env", "exchange_ab", "fit_exp_mode"):
        if k in cfg:
            line["config"][k] = cfg[k]
    r = out.get("roofline") or {}
    line["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms")}
    line["roofline"]["hbm_frac"] = g(r, "hbm", "frac")
    line["roofline"]["traffic_over_algorithmic"] = g(r, "hbm", "traffic_over_algorithmic")
    c = out.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = {"value": c.get("value"), "unit": "units/s", "cores": c.get("cores"), "kind": c.get("kind"),
                                "sample": "first %s blocks of the workload, %.0f s" % (c.get("blocks"), c.get("seconds") or 0)}
    legs = {}
    if "stream" in out:
        legs["stream"] = g(out, "stream", "ms_per_step"); legs["cont"] = g(out, "stream", "continuous", "ms_per_step")
    for short, key in (("e2e", "end_to_end"), ("e2e100k", "end_to_end_100k"), ("e2emix", "end_to_end_mixed"), ("e2emix100k", "end_to_end_mixed_100k"),
                       ("e2e_c5", "end_to_end_c5")):
        if key in out:
            legs[short] = g(out, key, "value") if "error" not in out[key] else "error"
    for key in ("c1", "c2"):
        if key in out:
            legs[key] = sp(key, "batch"); legs[key + "pb"] = sp(key, "per_block_shim_sequence")[:3]
    if "shard" in out:
        legs["shard"] = [g(out, "shard", "ms_per_step"), g(out, "shard", "efficiency"), g(out, "shard", "stream", "ms_per_step"),
                         g(out, "shard", "stream", "continuous", "ms_per_step"), g(out, "shard", "roofline", "valu", "frac")]
    if "mixed" in out:
        legs["mixed"] = [g(out, "mixed", "ms_per_step"), g(out, "mixed", "continuous", "ms_per_step"), g(out, "mixed", "resident", "ms_per_step"),
                         g(out, "mixed", "resident", "value"), g(out, "mixed", "resident", "roofline", "valu", "frac")]
    if "rows" in out:   # uniform batches by row count at 150 columns: [rows, fraction of the issue roof]
        legs["rows"] = [[r["rows"], r["issue_frac"]] for r in g(out, "rows", "shapes", default=[])]
    if "c5" in out:
        t = g(out, "c5", "roofline", "traffic")
        legs["c5"] = [g(out, "c5", "ms_per_step"), g(out, "c5", "roofline", "valu", "frac"), None if t is None else t / 1e12]
    if "cpu_baseline_all_cores" in out:
        legs["cpu16"] = [g(out, "cpu_baseline_all_cores", "value"), g(out, "cpu_baseline_all_cores", "cores")]
    for k in ("weak_scaling", "native_driver_multi_gpu"):
        if k in out:
            v = out[k]
            legs[k] = {kk: (g(v, kk, "value") if isinstance(v.get(kk), dict) else v.get(kk)) for kk in v if kk != "note"} if isinstance(v, dict) else v
    if "n_ranks_seen" in out:
        line["n_ranks_seen"] = out["n_ranks_seen"]
        line["distinct_devices"] = g(out, "devices", "distinct_devices")
    line["legs"] = legs
    line = _sig(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > 1950:   # never let the tail cut the head: drop the least important entries first
        for k in ("cpu16", "c1pb", "c2pb", "cont", "native_driver_multi_gpu", "weak_scaling", "e2e"):
            line["legs"].pop(k, None)
            if len(json.dumps(line, separators=(",", ":"))) <= 1950:
                break
    return line


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-json", default=None, help="where the full record goes (default gpurun_out/bench_full.json); the printed line is the compact one")
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
    ap.add_argument("--exchange", choices=("auto", "maxima", "fits"), default="auto",
                    help="what the ranks all-gather per step: the per-sample maxima (north_star's exchange; 4 B x samples per "
                         "block, written by the kernel straight into the gathered tensor) or only the per-block fit results "
                         "(16 B per block: enough for p-values, since every rank fits its own blocks on the device).  auto (default): the maxima, "
                         "unless a rank's share of them exceeds 64 MB per step (configs[4] whole: 12 500 blocks x 10 000 samples = 500 MB per rank) -- "
                         "then the fits (SURVEY.md section 8e)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: wait for every step's all-gather before the next step starts")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: step i's gather runs over xGMI while step i + 1 is scored; all of them are complete when the timed region "
                         "ends.  Default (neither flag): both orders are timed during the warm-up and the faster one runs the timed region "
                         "(RCCL's all-gather is a kernel of its own: beside a k_null that holds every register of a SIMD it may cost more "
                         "than it hides; config.exchange reports both timings and the choice)")
    ap.add_argument("--ab-steps", type=int, default=0, help="steps per order in that warm-up comparison (0: max(3, --warmup))")
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
    init_error = None
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
            # The path's one collective moves 5 MB per rank (configs[3]).  RCCL's all-gather is a CU kernel with one workgroup per
            # channel; its default channel count (tuned for GB-sized tensors) would put tens of workgroups beside k_null, whose
            # two-row instantiation holds every VGPR of the SIMDs it runs on.  Two channels keep the gather on one or two CUs; the
            # ring is per-link bound at this size anyway.  An explicit setting in the environment wins.
            os.environ.setdefault("NCCL_MAX_NCHANNELS", "2")
            os.environ.setdefault("NCCL_MIN_NCHANNELS", "1")
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            except Exception as e:   # noqa: BLE001 -- the line must still be printed: every exchange through host memory over gloo instead
                init_error = "init_process_group(nccl): %s: %s" % (type(e).__name__, str(e)[:300])
                coll_dev = "cpu"
                dist.init_process_group("gloo")

    import numpy as np

    from rnacode_amd import api
    from rnacode_amd import distributed as rd
    from rnacode_amd.synth import synth_blocks

    # If anything on the RCCL path raises -- the warm-up comparison of the two exchange orders, a gather, a reduction -- the run goes on
    # with every collective through host memory over a gloo group made now (`fb`), says so in config.exchange_ab, and still prints its
    # line: a first run on an 8-GPU node must not end without a number.  (A failure on ONE rank only can still hang the others inside
    # the collective until RCCL's own timeout; what this covers is the symmetric kind: unsupported call, allocation, configuration.)
    fb = {"on": False, "error": init_error, "group": None, "fail_test": os.environ.get("RC_BENCH_TEST_FAIL_EXCHANGE"), "calls": 0}
    if multi and coll_dev != "cpu":
        try:
            fb["group"] = dist.new_group(backend="gloo")
        except Exception:   # noqa: BLE001 -- no gloo in this build: no fallback either
            fb["group"] = None

    def cdev():
        return "cpu" if fb["on"] else coll_dev

    def cgroup():
        return fb["group"] if fb["on"] else None

    def fall_back(e, where):
        if fb["on"] or fb["group"] is None:
            raise e
        fb["on"] = True
        fb["error"] = "%s: %s: %s" % (where, type(e).__name__, str(e)[:300])

    def inject(where):   # test hook
        if fb["fail_test"] == where and not fb["on"]:
            fb["calls"] += 1
            if fb["calls"] >= 2:
                raise RuntimeError("injected failure of the %s path (RC_BENCH_TEST_FAIL_EXCHANGE)" % where)

    t0 = time.perf_counter()
    if args.scaling == "weak":   # this rank's own shard: generator seed 1 + rank (generator seed 1)
        blocks = [b.upper() for b in synth_blocks(args.blocks, args.seqs, args.cols, seed=1 + rank)]
        counts = [args.blocks] * world
    else:                        # one stream of --blocks blocks, contiguous shards
        lo, hi = rd.shard_range(args.blocks, rank, world)
        blocks = [b.upper() for b in synth_blocks(args.blocks, args.seqs, args.cols, seed=1)[lo:hi]]
        counts = [rd.shard_range(args.blocks, r, world)[1] - rd.shard_range(args.blocks, r, world)[0] for r in range(world)]
    t_gen = time.perf_counter() - t0
    exchange_auto = args.exchange == "auto"
    if exchange_auto:   # every rank decides alike: the largest shard's payload
        limit = float(os.environ.get("RC_BENCH_EXCHANGE_AUTO_MB", "64")) * (1 << 20)   # (the variable: a test hook)
        args.exchange = "fits" if max(counts) * args.samples * 4 > limit else "maxima"

    # one GPU, the default workload: the legs that start child processes run first, before this process touches a GPU
    single = world == 1 and not multi
    default_workload = (args.blocks, args.cols, args.seqs, args.samples) == CONFIGS[args.config][:4]
    extras = single and not args.no_extra and default_workload and args.config == "c3"
    cpu = None
    cpu_all = None
    legs = {}
    mixed = blocks5 = None
    genm = gen5 = 0.0
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if extras:
        # the other workloads of the extra legs, generated now: their CPU baselines run before this process creates its HIP context
        import numpy as _np
        from rnacode_amd.synth import synth_block
        rng = _np.random.RandomState(5)
        t0 = time.perf_counter()
        mixed = [synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True).upper() for i in range(10000)]
        genm = time.perf_counter() - t0
        nb5, cols5, seqs5, n5, desc5 = CONFIGS["c5"]
        t0 = time.perf_counter()
        blocks5 = [b.upper() for b in synth_blocks(nb5, seqs5, cols5, seed=1)]
        gen5 = time.perf_counter() - t0
    if single and not args.no_cpu_baseline:
        cpu = cpu_baseline(blocks, args.samples, args.seed_base)
        if extras and "blocks" in cpu:
            try:   # the cgroup quota, where there is one (what rc_host_cpus reports as well)
                with open("/sys/fs/cgroup/cpu.max") as fh:
                    q, per = fh.read().split()[:2]
                    if q != "max":
                        usable = max(1, min(usable, -(-int(q) // int(per))))
            except Exception:   # noqa: BLE001
                pass
            cpu_all = cpu_baseline_all_cores(blocks, args.samples, args.seed_base, cpu["blocks"], usable)
            legs["cpu_baselines"] = {
                "mixed": cpu_baseline(mixed, args.samples, args.seed_base, target_s=8.0, what="the mixed workload"),
                "c5": cpu_baseline(blocks5, CONFIGS["c5"][3], args.seed_base, target_s=8.0, max_n=1000, what="the c5 workload"),
            }
            for key, (gname, _) in FILE_LEGS.items():
                gb, gn, gseed, _pk = golden_blocks(gname)
                gb = [b for b in gb if b.tree is not None and b.n >= 3]
                legs["cpu_baselines"][key] = cpu_baseline(gb, gn, gseed, target_s=8.0, what=f"tests/golden/{gname}")
    if extras:
        legs["end_to_end"] = end_to_end_leg(blocks, args.samples, args.seed_base)
        legs["end_to_end_100k"] = end_to_end_leg(blocks, args.samples, args.seed_base, runs=3, repeat=10)
        legs["end_to_end_mixed"] = end_to_end_leg(mixed, args.samples, args.seed_base, runs=5)   # file in -> listing out on the ten-class workload
        legs["end_to_end_mixed_100k"] = end_to_end_leg(mixed, args.samples, args.seed_base, runs=3, repeat=10)   # ... at steady state: the same ten times over
        legs["end_to_end_c5"] = end_to_end_leg(blocks5, CONFIGS["c5"][3], args.seed_base, runs=3)   # a 1000-block 12 x 300 file at n = 10 000 (configs[4]'s shape)

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
    # The exchange, two orders (equal shards, maxima, RCCL):
    #   overlapped -- two source tensors alternate, step i's all-gather is started without waiting for it and runs while step i + 1
    #                 is scored; a tensor is written again (step i + 2) only after the gather that read it has finished;
    #   synchronous -- every step waits for its gather.
    # Which one is faster depends on what RCCL's gather kernel costs beside k_null (see --overlap): unless forced, both are timed during
    # the warm-up and the faster one runs the timed region (`mode`).
    overlap = None
    mode = {"overlap": False}
    step_no = 0
    can_overlap = multi and args.exchange == "maxima" and coll_dev != "cpu" and len(set(counts)) == 1
    if os.environ.get("RC_BENCH_TEST_OVERLAP_ON_CPU") and multi and args.exchange == "maxima" and len(set(counts)) == 1:
        can_overlap = True   # test hook: the overlapped order over gloo (CPU tensors), so that a box without N GPUs covers both branches
    if multi and args.exchange == "maxima":
        dev = f"cuda:{local_rank}"
        if can_overlap and not args.no_overlap:
            overlap = rd.OverlappedGather(len(blocks), args.samples, dev, stage_on_cpu=(coll_dev == "cpu"))
            mine = overlap.src[0]
            mode["overlap"] = True
        else:
            mine = torch.empty((len(blocks), args.samples), dtype=torch.float32, device=dev)
        batch.bind_maxima(mine.data_ptr())

    def exchange(fit_rows, src=None):
        """The path's one exchange (RCCL over xGMI): results of all blocks on every rank.  Synchronised before it
        returns: the next step overwrites the gathered buffer's source on the library's own streams."""
        nonlocal gathered
        for attempt in (0, 1):
            try:
                if attempt == 0:
                    inject("gather")
                if args.exchange == "maxima":
                    src = mine if src is None else src
                    gathered = rd.all_gather_maxima(src if cdev() != "cpu" else src.cpu(), counts, group=cgroup())
                else:
                    fits = torch.from_numpy(fit_rows).to(cdev(), non_blocking=True)
                    gathered = rd.all_gather_fits(fits, counts, group=cgroup())
                if cdev() != "cpu":
                    torch.cuda.current_stream().synchronize()
                return
            except Exception as e:   # noqa: BLE001
                fall_back(e, "all-gather")

    def resident_step():
        nonlocal step_no
        if overlap is not None:
            src = overlap.source(step_no)    # (waits for the gather of two steps ago, long done)
            batch.bind_maxima(src.data_ptr())
            batch.run()                      # synchronous: this step's maxima are complete in the tensor
            if mode["overlap"] and not fb["on"]:
                try:
                    inject("overlap")
                    overlap.launch(step_no)      # their all-gather runs while the next step is scored
                except Exception as e:   # noqa: BLE001
                    fall_back(e, "overlapped all-gather")
                    mode["overlap"] = False
                    exchange(None, src)
            else:
                exchange(None, src)
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
            if mode["overlap"] and not fb["on"]:
                try:
                    overlap.launch(step_no)
                except Exception as e:   # noqa: BLE001
                    fall_back(e, "overlapped all-gather (stream)")
                    mode["overlap"] = False
                    exchange(None, src)
            else:
                exchange(None, src)
            step_no += 1
        elif multi:
            exchange(np.concatenate(rows) if args.exchange == "fits" else None)
        return rows

    def sync():
        nonlocal gathered
        if multi:
            if overlap is not None and not fb["on"]:          # every all-gather started so far is complete before the clock is read
                try:
                    g = overlap.finish()
                    if g is not None and mode["overlap"]:
                        gathered = g
                except Exception as e:   # noqa: BLE001
                    fall_back(e, "waiting for the overlapped all-gathers")
                    mode["overlap"] = False
            if coll_dev != "cpu":
                torch.cuda.synchronize()
            try:
                dist.barrier(group=cgroup())
            except Exception as e:   # noqa: BLE001
                fall_back(e, "barrier")
                dist.barrier(group=cgroup())
            if coll_dev != "cpu":
                torch.cuda.synchronize()

    if multi:   # RCCL's first collective builds its rings: keep that out of the timed steps even with --warmup 0
        try:
            probe = torch.zeros((1, 4), dtype=torch.float32, device=coll_dev)
            rd.all_gather_rows(probe, [1] * world)
        except Exception as e:   # noqa: BLE001
            fall_back(e, "first collective")
        sync()

    # ---- which order of the exchange?  k steps of each between barriers, the slowest rank's time decides (every rank takes the same branch)
    exchange_ab = None
    if overlap is not None and not args.overlap and not fb["on"]:
        k = args.ab_steps or max(3, args.warmup)
        t_ab = {}
        for name, flag in (("synchronous", False), ("overlapped", True), ("synchronous_again", False)):
            if fb["on"]:
                break
            mode["overlap"] = flag
            resident_step()              # one untimed step per order: its first gather may still build something
            sync()
            t0 = time.perf_counter()
            for _ in range(k):
                resident_step()
            sync()
            t_ab[name] = rd.max_over_ranks((time.perf_counter() - t0) / k * 1e3, device=cdev(), group=cgroup())
        if fb["on"]:
            mode["overlap"] = False
            exchange_ab = {"all_timings_ms": t_ab}
        else:
            t_sync = min(t_ab["synchronous"], t_ab["synchronous_again"])   # the second run of it guards against a warm-up effect in the first
            mode["overlap"] = bool(t_ab["overlapped"] < t_sync)
            exchange_ab = {"ms_per_step_synchronous": t_sync, "ms_per_step_overlapped": t_ab["overlapped"], "steps_each": k,
                           "all_timings_ms": t_ab, "chosen": "overlapped" if mode["overlap"] else "synchronous"}
    elif overlap is not None:
        exchange_ab = {"chosen": "overlapped", "forced": "--overlap"}
    elif multi and args.exchange == "maxima":
        exchange_ab = {"chosen": "synchronous", "forced": "--no-overlap" if args.no_overlap else "ragged shards or a CPU backend"}

    # ---- the devices the ranks sit on: the SCALE record must show N distinct GPUs
    devices_seen = None
    if multi:
        me = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname()}
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            me["name"] = pr.name
            me["uuid"] = str(getattr(pr, "uuid", "")) or None
            me["pci"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        except Exception as e:   # noqa: BLE001
            me["error"] = repr(e)
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, me)
        ids = [(d.get("host"), d.get("uuid") or d.get("pci") or d.get("local_rank")) for d in seen]
        devices_seen = {"n_ranks_seen": dist.get_world_size(), "distinct_devices": len(set(ids)), "ranks": seen}

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
            elapsed = rd.max_over_ranks(elapsed, device=cdev(), group=cgroup())
        return elapsed, tk, nlaunch

    elapsed, tk, nlaunch = timed(resident_step, True)
    if args.dump_gathered and rank == 0:
        np.save(args.dump_gathered, gathered.cpu().numpy() if multi else batch.maxScores_all())
    prep_host_ms, prep_kernels_ms, up_bytes = batch.prep_timing()
    s_elapsed = None
    c_elapsed = None
    if stream is not None:
        s_elapsed, _, _ = timed(stream_step, False)
        if multi:
            # N > 1: the shard's passes as ONE stream per rank as well (round 5: a shard streamed pass by pass took 8.0 ms against 5.9 as a continuous
            # stream -- a strong-scaled run that restarts its pipeline every pass caps an 8-GPU curve near 5.9 x before the exchange).  Every pass still
            # ends with the path's one exchange, of THAT pass's results: the maxima of pass p go to buffer p mod (depth + 1), so that the sub-batches of
            # the passes queued behind it (at most `depth`) never write what a gather is still reading.
            nbuf = args.depth + 1
            srcs = [torch.empty((len(blocks), args.samples), dtype=torch.float32, device=f"cuda:{local_rank}") for _ in range(nbuf)] if args.exchange == "maxima" else None
            big = max(sub_sizes)
            one = [(lo, min(len(blocks), lo + big)) for lo in range(0, len(blocks), big)]
            firstp = [(0, sub_sizes[0])] + ([(sub_sizes[0], one[0][1])] if one[0][1] > sub_sizes[0] else []) + one[1:]
            slices = [(0, lo, hi, k == len(firstp) - 1) for k, (lo, hi) in enumerate(firstp)]
            for ps in range(1, args.steps):
                slices += [(ps, lo, hi, k == len(one) - 1) for k, (lo, hi) in enumerate(one)]
            sync()
            gc.collect()
            gc.disable()
            sent = done = 0
            rows = []
            t0 = time.perf_counter()
            while done < len(slices):
                while sent < len(slices) and stream.pending < stream.depth:
                    ps, lo, hi, _ = slices[sent]
                    stream.submit(marshalled, lo, hi, (srcs[ps % nbuf].data_ptr() + 4 * lo * args.samples) if srcs else 0)
                    sent += 1
                sb = stream.next()
                rows.append(sb.fits())
                sb.close()
                ps, _, _, endp = slices[done]
                done += 1
                if endp:
                    exchange(np.concatenate(rows) if args.exchange == "fits" else None, srcs[ps % nbuf] if srcs else None)
                    rows = []
            sync()
            c_elapsed = rd.max_over_ranks(time.perf_counter() - t0, device=cdev(), group=cgroup())
            gc.enable()
            del srcs
        else:
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
            for attempt in (0, 1):
                try:
                    if args.exchange == "maxima":
                        rd.all_gather_maxima(wmine if cdev() != "cpu" else wmine.cpu(), wcounts, group=cgroup())
                    else:
                        rd.all_gather_fits(torch.from_numpy(wbatch.fits()).to(cdev()), wcounts, group=cgroup())
                    break
                except Exception as e:   # noqa: BLE001
                    fall_back(e, "all-gather (weak scaling)")
            if cdev() != "cpu":
                torch.cuda.current_stream().synchronize()

        w_elapsed, _, _ = timed(weak_step, False)
        _, wcs = wbatch.work()
        weak = {"value": wcs * world * args.steps / w_elapsed, "unit": "alignment-columns*samples/s", "ms_per_step": w_elapsed / args.steps * 1e3,
                "blocks_per_gpu": len(wblocks), "scaling": "weak"}
        wbatch.close()
        del wmine

    # ---- N > 1: the native driver's own multi-GPU mode (rnacode_hip --gpus N: N contexts in ONE process, no collective library) on the
    # whole input as a file, measured by rank 0 while the other ranks wait on the host (a store key, not a collective: nothing of
    # theirs runs on the GPUs meanwhile).  A fresh process per run, wall time to the last output byte, as the one-GPU end_to_end leg.
    native_multi = None
    native_hook = bool(os.environ.get("RC_BENCH_TEST_NATIVE_MULTI"))   # test hook: this leg on a box with one GPU (every context on device 0), any workload
    if world > 1 and args.scaling == "strong" and not args.no_extra and ((coll_dev != "cpu" and default_workload) or native_hook):
        # the other ranks wait at a barrier of a gloo group: on the host, with nothing of theirs spinning on a GPU meanwhile
        try:
            host_group = dist.new_group(backend="gloo")
        except Exception:   # noqa: BLE001 -- no gloo here: wait in the default group instead (its barrier kernel occupies a CU or two)
            host_group = None
        if rank == 0:
            try:
                if torch.cuda.device_count() < world and not native_hook:
                    native_multi = {"skipped": f"this process sees {torch.cuda.device_count()} device(s), not {world}: the launcher masks the GPUs per rank"}
                else:
                    full = [b.upper() for b in synth_blocks(args.blocks, args.seqs, args.cols, seed=1)]
                    rep = 1 if native_hook else 10   # 100 000 blocks: on 10 000 the fixed costs (process start, HIP contexts) are most of the run
                    native_multi = {"gpus_1": end_to_end_leg(full, args.samples, args.seed_base, runs=3, repeat=rep),
                                    f"gpus_{world}": end_to_end_leg(full, args.samples, args.seed_base, runs=3, repeat=rep,
                                                                    extra_args=["--gpus", str(world)] + (["--devices", ",".join(["0"] * world)] if native_hook else [])),
                                    "note": "rnacode_amd/rnacode_hip on a 100 000-block MAF (the workload's blocks ten times over), trees fitted on the GPUs, "
                                            "fresh process per run; gpus_1 beside it on device 0 for the ratio; reading the file (one thread ... eight "
                                            "parser threads) and writing the listing (one thread, input order) do not shard; the other ranks of this job "
                                            "hold their contexts but run nothing"}
                    del full
            except Exception as e:   # noqa: BLE001 -- this leg must never cost the run its line
                native_multi = {"error": repr(e)}
        dist.barrier(group=host_group)

    sa, cs = batch.work()
    if multi:   # units of all ranks (shards may differ by one block under strong scaling)
        tot = torch.tensor([float(cs), float(sa)], dtype=torch.float64, device=cdev())
        dist.all_reduce(tot, group=cgroup())
        cs, sa = int(tot[0].item()), int(tot[1].item())
    total_units = cs * args.steps
    value = total_units / elapsed
    kernel_name = batch.null_kernel()
    if extras:
        main_ms = elapsed / args.steps * 1e3
        # ---- c1, c2: the two inputs the reference ships (BASELINE.json configs[0], [1]), latency per file
        for key in FILE_LEGS:
            legs[key], _, _, _ = file_leg(api, ctx, key)
        # ---- shard: what ONE of eight ranks holds under configs[3] (strong scaling), with an eighth of the host's CPUs
        lo, hi = rd.shard_range(len(blocks), 0, 8)
        t8 = max(1, cpus // 8)
        ctx.set_host_threads(t8)
        sh_ms, sh_null, sh_kernel, sh_units, sh_steps, sh_m = resident_leg(api, ctx, blocks[lo:hi], params, args.extra_seconds)
        sh_sub = api.stream_plan(ctx, params, hi - lo, 1)
        sh_res_spread = resident_leg.spread
        sh_stream_ms, sh_passes = stream_leg(api, ctx, sh_m, params, sh_sub, args.extra_seconds)
        sh_stream_spread = stream_leg.spread
        sh_cont_ms, sh_cont_passes = continuous_leg(api, ctx, sh_m, hi - lo, params, sh_sub, hi - lo, args.extra_seconds)
        legs["shard"] = {
            "blocks": hi - lo, "of": len(blocks), "ranks": 8, "host_threads": t8, "usable_cpus": cpus,
            "ms_per_step": sh_ms, "ms_per_step_spread": sh_res_spread, "steps": sh_steps, "kernel_ms": sh_null, "ideal_ms": main_ms / 8,
            "roofline": roofline_object(blocks[lo:hi], args.samples, sh_null, sh_kernel, "shard", True),
            "efficiency": (main_ms / 8) / sh_ms, "value": sh_units / (sh_ms * 1e-3),
            "stream": {"ms_per_step": sh_stream_ms, "ms_per_step_spread": sh_stream_spread, "passes": sh_passes, "frac_of_resident": sh_ms / sh_stream_ms, "sub_batches": sh_sub,
                       "efficiency": (main_ms / 8) / sh_stream_ms,
                       "continuous": {"ms_per_step": sh_cont_ms, "passes": sh_cont_passes, "frac_of_resident": sh_ms / sh_cont_ms,
                                      "efficiency": (main_ms / 8) / sh_cont_ms, "sub_batches_first_pass": sh_sub, "sub_batch_later_passes": hi - lo,
                                      "note": "the shard's blocks submitted pass after pass as ONE stream (what a rank sees on a long input): the "
                                              "pipeline fills once, every later pass is one sub-batch queued while its predecessor is scored; "
                                              "ms per pass = whole region / passes, fill and drain included"}},
            "note": "ms_per_step of the extra legs is the MEDIAN step of the timed region (min / median / max / mean in ms_per_step_spread); "
                    "efficiency = (ms per step of the whole 10 000-block workload on this GPU / 8) / ms per step of the shard: what strong "
                    "scaling to eight GPUs can reach before the all-gather; the stream prepares its blocks on usable_cpus // 8 host threads"}
        ctx.set_host_threads(threads)
        del sh_m
        # ---- mixed: ten row-count classes, ragged lengths, as a stream from host blocks
        mx_ms, mx_null, mx_kernel, mx_units, mx_steps, mx_m = resident_leg(api, ctx, mixed, params, args.extra_seconds, warmup=2)
        mx_res_spread = resident_leg.spread
        mx_interval = resident_leg.null_interval_ms
        mx_roof = roofline_object(mixed, args.samples, mx_interval, "%d class launches side by side (largest: %s)" % (len({b.n for b in mixed}), mx_kernel), "mixed", True)
        classes = len({b.n for b in mixed})
        mx_sub = api.stream_plan(ctx, params, len(mixed), classes)
        # (three warm-up passes: the stream's three batch objects each grow their multi-GB staging areas to the largest sub-batch once)
        mx_stream_ms, mx_passes = stream_leg(api, ctx, mx_m, params, mx_sub, args.extra_seconds, warmup=3)
        mx_stream_spread = stream_leg.spread
        mx_cont_ms, mx_cont_passes = continuous_leg(api, ctx, mx_m, len(mixed), params, mx_sub, max(mx_sub), args.extra_seconds)
        legs["mixed"] = {"workload": "synthetic MAF 10000 blocks of 3..12 rows x 60..200 cols, --num-samples %d" % args.samples,
                         "row_count_classes": classes, "value": mx_units / (mx_stream_ms * 1e-3), "unit": "alignment-columns*samples/s",
                         "ms_per_step": mx_stream_ms, "passes": mx_passes, "blocks_per_sub_batch": mx_sub, "host_threads": threads,
                         "ms_per_step_spread": mx_stream_spread,
                         "continuous": {"ms_per_step": mx_cont_ms, "passes": mx_cont_passes, "value": mx_units / (mx_cont_ms * 1e-3),
                                        "sub_batch_later_passes": max(mx_sub),
                                        "note": "the 10 000 blocks pass after pass as ONE stream: the pipeline fills once"},
                         "resident": {"ms_per_step": mx_ms, "steps": mx_steps, "value": mx_units / (mx_ms * 1e-3), "ms_per_step_spread": mx_res_spread,
                                      "sampling_interval_ms": mx_interval, "roofline": mx_roof},
                         "generate_s": genm, "note": "value = the streamed rate (host rc_block arrays -> fitted results on the host), from the median pass"}
        del mx_m, mixed

        # ---- rows: the row-count axis (real multiz MAFs are 17-, 30-, 46-way): uniform resident batches at 150 columns, the kernel
        # that ran and its fraction of the issue roof by the algorithmic flops (tools/rows_sweep.py is the full sweep, profiles/r06/rows_sweep_*.txt)
        legs["rows"] = rows_leg(api, ctx, args.samples, args.seed_base)
        # ---- c5: the long-block slice of configs[4] (last: 0.6 s launches at the HBM roof; the legs after it ran a few per cent slower)
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
                       "seed_base": args.seed_base, "sharding": f"blocks x{world}", "fit_exp_mode": ctx.fit_exp_mode,
                       "exchange": (("all-gather of " + args.exchange + (" (auto)" if exchange_auto else "") + (", step i's gather overlapped with the scoring of step i + 1 (all complete "
                                     "inside the timed region)" if (overlap is not None and mode["overlap"]) else ", every step waits for its gather")) if multi else "none")},
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
        if native_multi is not None:
            out["native_driver_multi_gpu"] = native_multi
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        per_leg = legs.pop("cpu_baselines", None)
        if per_leg:
            for k, v in per_leg.items():
                if k in legs and isinstance(legs[k], dict):
                    legs[k]["cpu_baseline"] = v
                    if "value" in v and legs[k].get("value"):
                        legs[k]["speedup_vs_cpu_baseline"] = legs[k]["value"] / v["value"]
        if cpu_all is not None:
            out["cpu_baseline_all_cores"] = cpu_all
            if "value" in cpu_all:
                out["speedup_vs_cpu_baseline_all_cores"] = value / cpu_all["value"]
        if devices_seen is not None:
            out["n_ranks_seen"] = devices_seen["n_ranks_seen"]
            out["devices"] = devices_seen
        if multi and coll_dev != "cpu":
            out["config"]["rccl_env"] = {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS")}
        if fb["error"] is not None:   # the RCCL path raised: what, and that the numbers above went through host memory
            exchange_ab = dict(exchange_ab or {}, error=fb["error"], chosen="synchronous (host fallback)")
        if exchange_ab is not None:
            out["config"]["exchange_ab"] = exchange_ab
        for k, v in legs.items():
            out[k] = v
        # The driver keeps the last 2000 characters of this process's output: the ONE line printed is therefore a compact one -- the
        # contract's keys, roofline, cpu_baseline and the numbers of every extra leg under short names (KEY below) -- and the full
        # record with its prose goes to a file (gpurun_out/bench_full.json, or --full-json): what profiles/<round>/bench_default_run.json is.
        full_path = args.full_json or os.path.join(ROOT, "gpurun_out", "bench_full.json")
        try:
            if os.path.dirname(full_path):   # (a bare file name has none: makedirs('') raises)
                os.makedirs(os.path.dirname(full_path), exist_ok=True)
            with open(full_path, "w") as fh:
                json.dump(out, fh)
        except OSError as e:
            print(f"bench.py: the full record could not be written to {full_path}: {e}", file=sys.stderr)
        print(json.dumps(compact_line(out), separators=(",", ":")))
    if stream is not None:
        stream.close()
    batch.close()
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
