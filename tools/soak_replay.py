#!/usr/bin/env python3
"""Replay a block tools/soak_parity.py could not match (gpurun_out/soak_fail/*.json): the batch it was scored in through the library, the
block through the oracle, both results printed side by side.  python tools/soak_replay.py file.json [alone]   (alone: the block as a batch of one)"""
import json
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, "tests")
from oracle import binding as ob  # noqa: E402
from rnacode_amd import api  # noqa: E402
from rnacode_amd.alnio import AlnBlock, AlnRow  # noqa: E402

d = json.load(open(sys.argv[1]))
alone = len(sys.argv) > 2


def block(e):
    rows = [AlnRow(n, s) for n, s in zip(e["names"], e["rows"])]
    rows[0].start, rows[0].length = e["start"], e["length"]
    return AlnBlock(rows, "r", e["tree"], e["kappa"])


batch_blocks = [block(d)] if alone else [block(e) for e in d["batch"]]
at = 0 if alone else next(i for i, e in enumerate(d["batch"]) if e["rows"] == d["rows"] and e["tree"] == d["tree"])
pars = dict(d["pars"])
hp = api.default_params(sampleN=d["n_samples"], seed_base=d["seed"], **pars)
ctx = api.Context(0)
b = api.Batch(ctx, batch_blocks, hp).run()
p = ob.default_params(d["n_samples"], blosum=pars.get("blosum", 62))
for k, v in pars.items():
    if k != "blosum":
        setattr(p, k, v)
res = ob.run_block(d["rows"], d["names"], d["start"], d["length"], d["tree"], d["kappa"], p, d["seed"])
key = lambda h: (-h["score"], h["strand"], h["frame"], h["start"], h["end"])  # noqa: E731
got, want = sorted(b.scoreAln(at), key=key), sorted(res.hss, key=key)
print("status", b.status(at), "maxima equal", np.array_equal(b.maxScores(at), np.float32(res.maxScores)))
if not np.array_equal(b.maxScores(at), np.float32(res.maxScores)):
    bad = np.flatnonzero(b.maxScores(at) != np.float32(res.maxScores))
    print("  samples that differ", bad[:10], b.maxScores(at)[bad[:10]], np.float32(res.maxScores)[bad[:10]])
print("fit  library", b.getExtremeValuePars(at), " oracle", (res.evd_rc, res.mu, res.lam))
print("maxima", np.sort(b.maxScores(at))[-5:], "...")
print("hss library:")
for h in got:
    print("   ", h["strand"], h["frame"], h["start"], h["end"], repr(np.float32(h["score"])), h.get("pvalue"))
print("hss oracle:")
for h in want:
    print("   ", h["strand"], h["frame"], h["start"], h["end"], repr(np.float32(h["score"])), h.get("pvalue"))
b.close()
ctx.close()
