#!/usr/bin/env python3
"""Where the 1.5 ms per block of the shim's call sequence go (bench.py's c2.per_block_shim_sequence): every block of
genomic-preprocessed.maf as a batch of one, third repetition timed.  python tools/trace_per_block.py"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from rnacode_amd import api  # noqa: E402

ctx = api.Context(0)
name, _ = bench.FILE_LEGS["c2"]
blocks, n, seed, pk = bench.golden_blocks(name)
params = api.default_params(sampleN=n, seed_base=seed, **pk)
tot = [0.0] * 6
for rep in range(3):
    tot = [0.0] * 6
    rows = []
    for b in blocks:
        t0 = time.perf_counter()
        batch = api.Batch(ctx, [b], params)
        t1 = time.perf_counter()
        batch.run()
        t2 = time.perf_counter()
        if batch.status(0) == api.RC_OK:
            batch.scoreAln(0); batch.getExtremeValuePars(0)
        t3 = time.perf_counter()
        t, nl = batch.timing()
        batch.close()
        t4 = time.perf_counter()
        rows.append((len(b.rows), len(b.rows[0].seq), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), t["null"], t["native"], t["evd_fit"]))
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            tot[i] += 1e3 * v
        tot[4] += t["null"]; tot[5] += t["native"]
print("rows cols | create run results close | device null native fit (ms)")
for r in rows:
    print("%4d %4d | %6.3f %6.3f %6.3f %6.3f | %6.3f %6.3f %6.3f" % r)
print("sum of %d blocks: create %.2f run %.2f results %.2f close %.2f ms; device null %.2f native %.2f" % (len(rows), *tot))
ctx.close()
