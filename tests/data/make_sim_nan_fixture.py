#!/usr/bin/env python3
"""tests/data/sim_nan_branch_matrix.json: blocks with one purine and one pyrimidine only, scored by the UNMODIFIED reference
(oracle/_ref/ref_harness, built by oracle/Makefile where /root/reference exists).  Run from the repository root in the development container:
    python tests/data/make_sim_nan_fixture.py
The seed block is the one tools/soak_parity.py could not match in round 6 (fat mode, seed 6206, round 308): G and T only.  With frequencies
of zero for both other nucleotides Seq-Gen's alphaR is 0 / 0, every branch matrix NaN, `r > P` false for every draw and every simulated
residue state 0 (evolve.c:167-175); the library's integer thresholds had taken NaN for "every draw exceeds it"."""
import importlib.util
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rnacode_amd.alnio import AlnBlock, AlnRow  # noqa: E402

spec = importlib.util.spec_from_file_location("svr", os.path.join(ROOT, "tools", "soak_oracle_vs_ref.py"))
svr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(svr)

ROWS = ["GGGGGTG---GG", "GGGGGTGTGTGG", "GGGGGTGTGTGG", "GGGGGTGTGTGG", "GGGGGTGT---G"]
NAMES = ["sp%d.chr1" % i for i in range(5)]
TREE = "((sp0.chr1:0.011966,sp3.chr1:0.011072):0.015074,sp1.chr1:0.000950,(sp2.chr1:0.003010,sp4.chr1:0.002991):0.017776);"
KAPPA, SEED, START, N = 2.593970775604248, 13611850, 2000, 63


def main():
    tr = str.maketrans
    variants = {"GT": ROWS, "AC": [r.translate(tr("GT", "AC")) for r in ROWS], "AT": [r.translate(tr("GT", "AT")) for r in ROWS],
                "CG": [r.translate(tr("GT", "GC")) for r in ROWS], "GT_long": [r * 6 for r in ROWS]}
    cases = []
    for name, rows in variants.items():
        rr = [AlnRow(n, s) for n, s in zip(NAMES, rows)]
        for r in rr:
            r.length = sum(ch != "-" for ch in r.seq)
        rr[0].start = START
        svr.write_harness_blocks([AlnBlock(rr, "r", TREE, KAPPA)], "/tmp/one.blk")
        out = subprocess.run([os.path.join(ROOT, svr.HARNESS), "/tmp/one.blk", "--samples", str(N), "--seed", str(SEED), "--no-digest", "--no-backtrack"],
                             capture_output=True, check=True)
        txt = re.sub(r"(?<![A-Za-z])(-?)inf(?![A-Za-z])", r"\1Infinity", re.sub(r"-?nan", "NaN", out.stdout.decode()))
        ref = json.loads(txt.splitlines()[0])
        cases.append(dict(name=name, rows=rows, names=NAMES, start=START, length=rr[0].length, tree=TREE, kappa=KAPPA, n_samples=N, seed=SEED,
                          maxScores=ref["maxScores"], evd_rc=ref["evd_rc"], hss=ref["hss"]))
    what = ("blocks with one purine and one pyrimidine only (Seq-Gen's alphaR = 0/0: NaN branch matrices, every simulated residue state 0), scored by "
            "the UNMODIFIED reference (oracle/_ref/ref_harness); found by tools/soak_parity.py in round 6 (fat mode, seed 6206, round 308)")
    with open(os.path.join(ROOT, "tests", "data", "sim_nan_branch_matrix.json"), "w") as f:
        json.dump(dict(what=what, made_by="tests/data/make_sim_nan_fixture.py", cases=cases), f, indent=1)


if __name__ == "__main__":
    main()
