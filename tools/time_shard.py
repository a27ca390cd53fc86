#!/usr/bin/env python3
"""Resident step time of a shard of the bench workload (what one of N ranks holds): tools/time_shard.py [blocks] [reps]"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
blocks = [b.upper() for b in synth_blocks(nb, 6, 120, seed=42)]
ctx = api.Context(0)
batch = api.Batch(ctx, blocks, api.default_params(sampleN=1000))
for _ in range(3):
    batch.run()
t0 = time.perf_counter()
acc = {}
for _ in range(reps):
    batch.run()
    t, _n = batch.timing()
    for k, v in t.items():
        acc[k] = acc.get(k, 0.0) + v
dt = (time.perf_counter() - t0) / reps
print(f"{nb} blocks of 6 x 120, n = 1000: {dt * 1e3:.3f} ms per run (wall), device " + ", ".join(f"{k} {v / reps:.3f}" for k, v in acc.items()))
batch.close()
ctx.close()
