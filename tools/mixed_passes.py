#!/usr/bin/env python3
"""Per-pass times of bench.py's `mixed` leg (10 000 blocks of 3..12 rows x 60..200 columns as a stream, the library's sub-batch plan):
   tools/mixed_passes.py [passes]        -- outliers among the passes show up here, not in the leg's mean"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_block  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(5)
blocks = [synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True).upper() for i in range(10000)]
ctx = api.Context(0)
api.lib().rc_set_stream_cache(0)
p = api.default_params(sampleN=1000, seed_base=42)
m = api.Marshalled(blocks)
m.set_trees()
plan = api.stream_plan(ctx, p, len(blocks), len({b.n for b in blocks}))
batch = api.Batch(ctx, m, p)
res = []
for _ in range(8):
    t0 = time.perf_counter()
    batch.run()
    res.append((time.perf_counter() - t0) * 1e3)
batch.close()
st = api.Stream(ctx, p, 3)
times = []
for rep in range(passes):
    t0 = time.perf_counter()
    for sb in api.score_stream(ctx, m, p, plan, stream=st):
        sb.fits()
        sb.close()
    times.append((time.perf_counter() - t0) * 1e3)
print("plan", plan)
print("resident ms:", " ".join(f"{t:.1f}" for t in res))
print("stream passes ms:", " ".join(f"{t:.1f}" for t in times))
print(f"stream: median {np.median(times):.1f}, mean {np.mean(times):.1f}, max {np.max(times):.1f}")
st.close()
ctx.close()
