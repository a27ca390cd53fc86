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
