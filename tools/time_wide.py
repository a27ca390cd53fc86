#!/usr/bin/env python3
"""Time the generic kernels on wide blocks: tools/time_wide.py [rows] [cols] [blocks] [samples]"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

rows, cols, nb, n = (int(x) for x in (sys.argv[1:5] + ["100", "300", "16", "1000"][len(sys.argv) - 1:]))
blocks = [b.upper() for b in synth_blocks(nb, rows, cols, seed=3)]
ctx = api.Context(0)
batch = api.Batch(ctx, blocks, api.default_params(sampleN=n)).run()
t0 = time.perf_counter()
batch.run()
dt = time.perf_counter() - t0
t, _ = batch.timing()
print(f"{nb} blocks of {rows} x {cols}, n = {n}: {dt * 1e3:.1f} ms per run, null {t['null']:.1f} ms, native {t['native']:.1f} ms, "
      f"{nb * cols * n / dt:.3g} alignment-columns*samples/s")
batch.close()
ctx.close()
