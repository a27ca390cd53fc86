#!/usr/bin/env python3
"""The mixed workload (10 000 blocks of 3..12 rows x 60..200 columns, n = 1000) as ONE resident batch, a few runs: what tools/prof_mixed.sh profiles."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_block  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.RandomState(5)
blocks = [synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True).upper() for i in range(10000)]
ctx = api.Context(0)
api.lib().rc_set_stream_cache(0)
batch = api.Batch(ctx, blocks, api.default_params(sampleN=1000, seed_base=42))
for _ in range(runs):
    t0 = time.perf_counter()
    batch.run()
    dt = time.perf_counter() - t0
    t, nl = batch.timing()
    print(f"run {dt * 1e3:.1f} ms: sampling interval {t['null']:.1f} ms in {nl['null']} launches, native {t['native']:.1f}, fit {t['evd_fit']:.2f}")
batch.close()
ctx.close()
