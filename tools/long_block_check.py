"""Very long blocks (4 x 6000 against the oracle, 3 x 15000 for completion): the native tile buffer and the staging scratch are bounded by memory, not by the block."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from rnacode_amd import api
from rnacode_amd.synth import synth_block
rng = np.random.RandomState(3)
for n_rows, cols in ((4, 6000), (3, 15000)):
    b = synth_block(rng, n_rows, cols, index=0, gaps=True).upper()
    ctx = api.Context(0)
    n, seed = 4, 99
    t = time.time()
    batch = api.Batch(ctx, [b], api.default_params(sampleN=n, seed_base=seed)).run()
    dt = time.time() - t
    print(n_rows, "x", cols, "status", batch.status(0), "HIP %.2f s" % dt, "maxima", batch.maxScores(0), "hss", len(batch.scoreAln(0)))
    if cols <= 6000:
        t = time.time()
        res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(n), seed)
        print("oracle %.1f s" % (time.time() - t), "equal maxima:", np.array_equal(batch.maxScores(0), np.float32(res.maxScores)), "hss equal:", len(res.hss) == len(batch.scoreAln(0)))
    batch.close(); ctx.close()
