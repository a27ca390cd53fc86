#!/usr/bin/env python3
"""Diagnostic: per-sample maxima of the HIP path against the CPU oracle for a few shapes that take the codes-from-L2 kernels."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

ctx = api.Context(0)
n, seed = 128, 42
for rows, cols, gaps in ((8, 137, True), (8, 137, False), (7, 150, True), (12, 150, True), (12, 300, True), (4, 300, True), (9, 90, True)):
    blocks = [b.upper() for b in synth_blocks(3, rows, cols, seed=7, gaps=gaps)] if gaps is not None else None
    batch = api.Batch(ctx, blocks, api.default_params(sampleN=n, seed_base=seed)).run()
    for i, blk in enumerate(blocks):
        want = ob.run_block([r.seq for r in blk.rows], [r.name for r in blk.rows], blk.rows[0].start, blk.rows[0].length, blk.tree, blk.kappa,
                            ob.default_params(n), seed)
        got = batch.maxScores(i)
        w = np.float32(want.maxScores)
        bad = np.flatnonzero(got != w)
        print(f"{rows} x {cols} gaps={gaps} block {i} L={blk.ref_len} kernel {batch.null_kernel()}: {len(bad)} of {n} samples differ", bad[:8], got[bad[:3]], w[bad[:3]], flush=True)
    batch.close()
ctx.close()
