#!/usr/bin/env python3
"""Dev tool: rc_fit_tree against the PhyML trees stored in the goldens (likelihood on equal terms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from helpers import block_from_golden
from rnacode_amd import api
for name in ("coding_aln_n100", "noncoding_aln_n100", "genomic_preprocessed_n100", "edge_cases_n50"):
    doc = load_golden(name); worst = 1e9; nb = 0; t0 = time.time(); wk = 0
    for e in doc["blocks"]:
        if "skipped" in e["ref"]:
            continue
        b = block_from_golden(e); nb += 1
        nwk, k = api.fit_tree(b)
        mine = api.tree_lnl(b, nwk, k); theirs = api.tree_lnl(b, e["ref"]["tree"], e["ref"]["kappa"])
        worst = min(worst, mine - theirs); wk = max(wk, abs(k - e["ref"]["kappa"]) / e["ref"]["kappa"])
        if mine - theirs < -0.01:
            print("  worse:", name, e["input"]["id"], mine - theirs)
    print(name, "min(lnL_mine - lnL_phyml) = %.5f" % worst, "max rel dkappa %.4f" % wk, "%.1f ms/blk (fit + 2 lnL)" % ((time.time() - t0) / nb * 1e3))
