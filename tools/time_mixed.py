#!/usr/bin/env python3
"""A batch shaped like a real genome-wide input: 10 000 blocks of 3..12 rows and 60..200 columns, n = 1000 (ten row-count classes, ragged lengths)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_block  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sub = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rng = np.random.RandomState(5)
blocks = [synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True).upper() for i in range(nb)]
units = sum(b.cols for b in blocks) * 1000
ctx = api.Context(0)
batch = api.Batch(ctx, blocks, api.default_params(sampleN=1000)).run()
t0 = time.perf_counter()
for _ in range(5):
    batch.run()
dt = (time.perf_counter() - t0) / 5
t, nl = batch.timing()
print(f"{nb} mixed blocks (3..12 rows, 60..200 cols), n = 1000: {dt * 1e3:.1f} ms per run, null {t['null']:.1f} ms in {nl['null']} launches, {units / dt:.3g} alignment-columns*samples/s")
m = api.Marshalled(blocks)
m.set_trees()
st = api.Stream(ctx, api.default_params(sampleN=1000), 3)
for rep in range(4):   # the first pass allocates the stream's buffers
    t0 = time.perf_counter()
    parts = list(api.score_stream(ctx, m, api.default_params(sampleN=1000), sub, depth=3, stream=st))
    dt = time.perf_counter() - t0
    print(f"as a stream of {sub}-block sub-batches, pass {rep}: {dt * 1e3:.1f} ms")
    for p in parts:
        p.close()
st.close()
batch.close()
ctx.close()
