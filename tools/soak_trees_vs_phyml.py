#!/usr/bin/env python3
"""Randomised check of the built-in tree + kappa estimator (rc_fit_tree) against the reference's treeML (PhyML, run through
oracle/_ref/ref_harness where /root/reference exists): likelihood of both trees on equal terms (rc_tree_lnl).
usage: tools/soak_trees_vs_phyml.py [seconds] [rng_seed]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api  # noqa: E402
from rnacode_amd.alnio import write_harness_blocks  # noqa: E402
from rnacode_amd.synth import synth_block  # noqa: E402

HARNESS = os.path.join("oracle", "_ref", "ref_harness")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    deltas, dk = [], []
    while time.time() - t0 < budget:
        blocks = []
        for i in range(8):
            n = int(rng.choice([3, 4, 5, 6, 8, 12, 20]))
            cols = int(rng.choice([30, 60, 120, 200, 400]))
            b = synth_block(rng, n, cols, index=i, gaps=bool(rng.rand() < 0.7), mean_branch=float(rng.choice([0.02, 0.1, 0.3]))).upper()
            b.tree, b.kappa = None, None    # the harness calls treeML (RNAcode.c:153) when no tree is given
            blocks.append(b)
        with tempfile.NamedTemporaryFile("w", suffix=".blk", delete=False) as fh:
            path = fh.name
        try:
            write_harness_blocks(blocks, path)
            out = subprocess.run([HARNESS, path, "--samples", "1", "--seed", "1", "--no-backtrack", "--no-digest"], capture_output=True)
        finally:
            os.unlink(path)
        refs = [json.loads(re.sub(r"-?nan", "NaN", l)) for l in out.stdout.decode().splitlines() if l.strip()]
        for b, ref in zip(blocks, refs):
            if "skipped" in ref:
                continue
            nwk, k = api.fit_tree(b)
            mine, theirs = api.tree_lnl(b, nwk, k), api.tree_lnl(b, ref["tree"], ref["kappa"])
            deltas.append(mine - theirs)
            dk.append(abs(k - ref["kappa"]) / ref["kappa"])
            if mine - theirs < -0.5:
                print(f"worse by {mine - theirs:.3f}: {b.n} x {b.cols}, kappa {k:.3f} vs {ref['kappa']:.3f}")
    d = np.array(deltas)
    print(f"{len(d)} blocks in {time.time() - t0:.0f} s: lnL(mine) - lnL(PhyML) min {d.min():.3f}, 1 % {np.percentile(d, 1):.3f}, median {np.median(d):.4f}, "
          f"max {d.max():.3f}; worse by more than 0.5: {(d < -0.5).sum()}, by more than 0.05: {(d < -0.05).sum()}; median relative kappa difference {np.median(dk):.4f}")


if __name__ == "__main__":
    main()
