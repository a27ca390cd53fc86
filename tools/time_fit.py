#!/usr/bin/env python3
"""How long one EVD fit takes on the device (rc_evd_fit: one block, 256 threads) for maxima of the shipped MAF and of a synthetic block."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

ctx = api.Context(0)
blocks, n, seed, pk = bench.golden_blocks("genomic_preprocessed_n1000")
b = api.Batch(ctx, blocks, api.default_params(sampleN=n, seed_base=seed)).run()
rows = [b.maxScores(i).astype(np.float64) for i in range(b.n) if b.status(i) == api.RC_OK]
b.close()
sb = api.Batch(ctx, [x.upper() for x in synth_blocks(4, 6, 120, seed=1)], api.default_params(sampleN=1000)).run()
rows2 = [sb.maxScores(i).astype(np.float64) for i in range(4)]
sb.close()
for name, rr in (("genomic-preprocessed", rows), ("synthetic 6 x 120", rows2)):
    for r in rr[:6]:
        ctx.evd_fit(r)
        t0 = time.perf_counter()
        for _ in range(20):
            ok, mu, lam = ctx.evd_fit(r)
        dt = (time.perf_counter() - t0) / 20
        print(f"{name}: {dt * 1e3:.3f} ms per rc_evd_fit call (malloc + copy + kernel + copy), ok {ok} mu {mu:.4f} lambda {lam:.4f}, min {r.min():.3f} max {r.max():.3f} mean {r.mean():.3f}")
ctx.close()
