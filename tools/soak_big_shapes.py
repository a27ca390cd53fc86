"""Large-shape parity against the oracle (12 x 300 .. 3 x 1200; `wide`: 48 .. 250 rows): tools/soak_big_shapes.py [wide]"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np
from oracle import binding as ob
from rnacode_amd import api
from rnacode_amd.synth import synth_block
rng = np.random.RandomState(77)
ctx = api.Context(0)
shapes = [(12, 300)] * 12 + [(20, 200)] * 6 + [(32, 150)] * 4 + [(6, 600)] * 4 + [(3, 1200)] * 3
if len(sys.argv) > 1 and sys.argv[1] == "wide":   # the generic kernels: several tiles per DP row, many sequences, long rows
    shapes = [(70, 300)] * 3 + [(100, 200)] * 2 + [(130, 150)] * 2 + [(250, 90)] + [(66, 600)] + [(48, 300)] * 2 + [(64, 200)]
blocks = [synth_block(rng, n, c, index=i, gaps=True).upper() for i, (n, c) in enumerate(shapes)]
n_s, seed = 130, 4242
t = time.time()
batch = api.Batch(ctx, blocks, api.default_params(sampleN=n_s, seed_base=seed)).run()
bad = 0
for i, b in enumerate(blocks):
    res = ob.run_block([r.seq for r in b.rows], [r.name for r in b.rows], b.rows[0].start, b.rows[0].length, b.tree, b.kappa, ob.default_params(n_s), seed)
    ok = np.array_equal(batch.maxScores(i), np.float32(res.maxScores))
    got = sorted(batch.scoreAln(i), key=lambda h: (-h["score"], h["strand"], h["frame"], h["start"]))
    want = sorted(res.hss, key=lambda h: (-h["score"], h["strand"], h["frame"], h["start"]))
    ok = ok and [(g["strand"], g["frame"], g["start"], g["end"], np.float32(g["score"])) for g in got] == [(w["strand"], w["frame"], w["start"], w["end"], np.float32(w["score"])) for w in want]
    bad += (not ok)
    if not ok: print("MISMATCH", i, shapes[i])
print(f"big shapes: {len(blocks)} blocks, {bad} mismatches, {time.time()-t:.0f} s")
