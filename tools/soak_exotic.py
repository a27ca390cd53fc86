#!/usr/bin/env python3
"""Degenerate inputs for the soaks (tools/soak_parity.py, tools/soak_oracle_vs_ref.py with RC_SOAK_EXOTIC=1): the block classes no random draw of
an ordinary alignment reaches -- the one that was wrong until round 6 was five rows of G and T only.  exotic(rng, block) changes the block in
place (sequences upper-case) and returns the name of what it did, or None."""
import re


def exotic(rng, b):
    r = rng.rand()
    cols = len(b.rows[0].seq)
    kind = None
    if r < 0.02:      # one letter only
        ch = str(rng.choice(list("ACGT")))
        for x in b.rows:
            x.seq = "".join(ch if c in "ACGT" else c for c in x.seq)
        kind = "one letter"
    elif r < 0.05:    # kappa at the ends of what an estimator may print
        b.kappa = float(rng.choice([0.01, 0.3, 40.0, 900.0]))
        kind = "kappa"
    elif r < 0.08:    # every branch zero, or saturated
        v = float(rng.choice([0.0, 0.0, 5.0]))
        b.tree = re.sub(r":[0-9.]+", ":%.6f" % v, b.tree)
        kind = "branches %g" % v
    elif r < 0.10 and len(b.rows) > 3:    # a row of gaps only
        q = int(rng.randint(1, len(b.rows)))
        b.rows[q].seq = "-" * cols
        kind = "gap row"
    elif r < 0.13 and cols >= 12:   # a reference with more gaps than residues
        s = list(b.rows[0].seq)
        for q in rng.permutation(cols)[: int(cols * rng.uniform(0.5, 0.8))]:
            s[q] = "-"
        if sum(c != "-" for c in s) >= 3:
            b.rows[0].seq = "".join(s)
            kind = "gappy reference"
    elif r < 0.15:    # identical rows
        for x in b.rows[1:]:
            x.seq = b.rows[0].seq
        kind = "identical rows"
    elif r < 0.17:    # branch lengths in exponent notation (what %g prints for short branches)
        b.tree = re.sub(r":([0-9.]+)", lambda m: ":%.5e" % float(m.group(1)), b.tree)
        kind = "exponent notation"
    elif r < 0.19:    # lower case and U among the residues
        for x in b.rows:
            x.seq = "".join((c.lower() if rng.rand() < 0.3 else c) for c in x.seq.replace("T", "U" if rng.rand() < 0.5 else "T"))
        kind = "lower case and U"
    if kind:
        for x in b.rows:
            x.length = sum(c != "-" for c in x.seq)
    return kind
