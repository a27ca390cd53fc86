#!/usr/bin/env python3
"""Resident step of uniform batches (one row-count class each): tools/time_uniform.py [blocks] [cols] [n] [rows,rows,...]
Prints, per row count, the k_null launch time, the instantiation that ran and the step time -- the A/B tool for the launches that read
their code words from L2 (RC_SUFFIX_CACHE, RC_HIGH_OCCUPANCY)."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 150
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rows = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [7, 8, 9, 11, 12, 13]
ctx = api.Context(0)
for r in rows:
    blocks = [b.upper() for b in synth_blocks(nb, r, cols, seed=42)]
    batch = api.Batch(ctx, blocks, api.default_params(sampleN=n))
    for _ in range(2):
        batch.run()
    reps = 5
    acc = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        batch.run()
        t, _nl = batch.timing()
        acc += t["null"]
    dt = (time.perf_counter() - t0) / reps
    print(f"{nb} blocks of {r} x {cols}, n = {n}: k_null {acc / reps:.2f} ms, step {dt * 1e3:.2f} ms, {batch.null_kernel()}", flush=True)
    batch.close()
ctx.close()
