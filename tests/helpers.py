"""Shared helpers for parity tests: golden access and the oracle-side parameter mapping."""
import numpy as np

from rnacode_amd.alnio import AlnBlock, AlnRow


def block_from_golden(entry) -> AlnBlock:
    inp, ref = entry["input"], entry["ref"]
    rows = [AlnRow(r["name"], r["seq"], r["start"], r["length"], r["strand"], r["full_length"]) for r in inp["rows"]]
    b = AlnBlock(rows, inp["id"], ref.get("tree", inp.get("tree")), ref.get("kappa", inp.get("kappa")))
    return b.upper()


def param_kwargs(doc):
    """Golden 'params' -> keyword overrides (names as in RNAcode.h:29-54)."""
    p = dict(doc.get("params") or {})
    p["sampleN"] = doc["samples"]
    return p


def f32(x):
    return np.float32(x)


P_REL = 1e-6


def close_p(a, b, rel=P_REL):
    """p-values are compared by RELATIVE error on p itself (north_star: scores and p-values within 1e-6): an absolute
    tolerance would accept 0 for a p of 2e-9 (RNAcode.c:182; the 99.0 sentinel of a failed fit compares equal)."""
    return abs(a - b) <= rel * abs(b)
