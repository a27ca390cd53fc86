#!/usr/bin/env python3
"""Where the latency of the two shipped inputs goes (bench.py's c1 / c2 legs): RC_TRACE=1 python tools/trace_files.py 2> trace.txt"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from rnacode_amd import api  # noqa: E402

ctx = api.Context(0)
for key in ("c1", "c2"):
    name, _ = bench.FILE_LEGS[key]
    blocks, n, seed, pk = bench.golden_blocks(name)
    params = api.default_params(sampleN=n, seed_base=seed, **pk)
    for rep in range(4):
        t0 = time.perf_counter()
        batch = api.Batch(ctx, blocks, params)
        t1 = time.perf_counter()
        batch.run()
        t2 = time.perf_counter()
        hss = [batch.scoreAln(i) for i in range(batch.n) if batch.status(i) == api.RC_OK]
        t3 = time.perf_counter()
        t, nl = batch.timing()
        ph, pk_ms, up = batch.prep_timing()
        batch.close()
        t4 = time.perf_counter()
        print(f"{key} rep {rep}: create {1e3 * (t1 - t0):.3f} ms (host prep {ph:.3f}, table kernels {pk_ms:.3f}), run {1e3 * (t2 - t1):.3f} ms, hss {1e3 * (t3 - t2):.3f} ms, close {1e3 * (t4 - t3):.3f} ms; "
              f"device: total {t['total']:.3f} mt {t['mt_stream']:.3f} null {t['null']:.3f} native {t['native']:.3f} fit {t['evd_fit']:.3f}; kernel {batch.null_kernel() if False else ''}")
ctx.close()
