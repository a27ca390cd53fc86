#!/usr/bin/env python3
"""The mixed workload's row-count classes one at a time (each class's ~1000 blocks as a resident batch of its own): ms, the kernel that ran,
and the FP32 issue fraction by the algorithmic flops -- which classes hold the side-by-side batch below the uniform batches' efficiency?"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_block  # noqa: E402

rng = np.random.RandomState(5)
blocks = [synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True).upper() for i in range(10000)]
ctx = api.Context(0)
api.lib().rc_set_stream_cache(0)
p = api.default_params(sampleN=1000, seed_base=42)
tot_ms = tot_fl = 0.0
for n in sorted({b.n for b in blocks}):
    sub = [b for b in blocks if b.n == n]
    batch = api.Batch(ctx, sub, p)
    batch.run()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); batch.run(); ts.append((time.perf_counter() - t0) * 1e3)
    t, nl = batch.timing()
    _, fl = bench.algorithmic_work(sub, 1000)
    ms = sorted(ts)[1]
    print(f"{n:2d} rows: {len(sub):4d} blocks, {ms:6.2f} ms per run (sampling {t['null']:.2f}), {batch.null_kernel():38s} issue fraction {fl / (t['null'] * 1e-3) / 78.6432e12:.3f}")
    tot_ms += ms; tot_fl += fl
    batch.close()
print(f"sum of the classes run alone: {tot_ms:.1f} ms, issue fraction {tot_fl / (tot_ms * 1e-3) / 78.6432e12:.3f}")
batch = api.Batch(ctx, blocks, p)
batch.run()
ts = []
for _ in range(4):
    t0 = time.perf_counter(); batch.run(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"all ten side by side: {sorted(ts)[1]:.1f} ms, issue fraction {tot_fl / (sorted(ts)[1] * 1e-3) / 78.6432e12:.3f}")
ctx.close()
