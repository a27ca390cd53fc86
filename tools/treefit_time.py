"""Time rc_fit_trees_device on 10 000 blocks (6 x 120, or `mixed`: 3..12 rows, 60..200 columns): tools/treefit_time.py [mixed]"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ctypes as C
from rnacode_amd import api, synth
blocks = synth.synth_blocks(10000, 6, 120, seed=1)
if len(sys.argv) > 1 and sys.argv[1] == "mixed":   # 3..12 rows, 60..200 columns
    import numpy as np
    rng = np.random.RandomState(5)
    blocks = [synth.synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True) for i in range(10000)]
ctx = api.Context(0)
n = len(blocks)
arr = (api.RcBlock * n)(); keep = []
for i, b in enumerate(blocks):
    rows = (C.c_char_p * b.n)(*[r.seq.encode() for r in b.rows]); names = (C.c_char_p * b.n)(*[r.name.encode() for r in b.rows])
    keep.append((rows, names)); arr[i] = api.RcBlock(b.n, b.cols, rows, names, b.rows[0].start, b.rows[0].length, None, 0.0)
cap = 1 << 12
buf = C.create_string_buffer(n * cap); kap = (C.c_float * n)(); ll = (C.c_double * n)()
for rep in range(3):
    t = time.time(); r = api.lib().rc_fit_trees_device(ctx._h, arr, n, buf, cap, kap, ll); print("rc_fit_trees_device:", r, round(time.time() - t, 4), "s")
