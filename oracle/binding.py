"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY, see rnacode_oracle.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class Params(C.Structure):
    _fields_ = [("Delta", C.c_float), ("Omega", C.c_float), ("omega", C.c_float),
                ("stopPenalty_0", C.c_float), ("stopPenalty_k", C.c_float),
                ("blosum", C.c_int), ("sampleN", C.c_int), ("cutoff", C.c_float), ("stopEarly", C.c_int)]


class Model(C.Structure):
    _fields_ = [("scores", C.c_float * 4), ("probs", C.c_float * 4), ("kappa", C.c_float),
                ("dist", C.c_float), ("freqs", C.c_float * 4)]


class Hss(C.Structure):
    _fields_ = [("start", C.c_int), ("end", C.c_int), ("startGenomic", C.c_int), ("endGenomic", C.c_int),
                ("startSite", C.c_int), ("endSite", C.c_int), ("strand", C.c_int), ("frame", C.c_int),
                ("score", C.c_float), ("pvalue", C.c_float)]

    def as_dict(self):
        return dict(strand=chr(self.strand), frame=self.frame, startSite=self.startSite, endSite=self.endSite,
                    start=self.start, end=self.end, startGenomic=self.startGenomic, endGenomic=self.endGenomic,
                    score=float(np.float32(self.score)), pvalue=float(np.float32(self.pvalue)))


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rnacode_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_prob_hky.restype = C.c_float
        _lib.orc_prob_hky.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), C.c_float]
        _lib.orc_pvalue.restype = C.c_float
        _lib.orc_pvalue.argtypes = [C.c_float, C.c_float, C.c_float]
        _lib.orc_score_matrix.restype = C.POINTER(C.c_float)
        _lib.orc_tie_replacements.restype = C.c_ulong
        _lib.orc_tie_replacements.argtypes = [C.c_int]
    return _lib


def _strs(xs: Sequence[str]):
    arr = (C.c_char_p * len(xs))(*[x.encode() for x in xs])
    return arr


def default_params(sampleN: int = 100, blosum: int = 62, stopEarly: int = 0, cutoff: float = 1.0) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    p.sampleN, p.blosum, p.stopEarly, p.cutoff = sampleN, blosum, stopEarly, cutoff
    return p


def prob_hky(i, j, d, freqs, kappa) -> float:
    f = (C.c_float * 4)(*freqs)
    return float(lib().orc_prob_hky(i, j, C.c_float(d), f, C.c_float(kappa)))


def calculate_bg(dist, freqs, kappa, blosum=62):
    f = (C.c_float * 4)(*freqs)
    s = (C.c_float * 4)()
    p = (C.c_float * 4)()
    lib().orc_calculate_bg(C.c_float(dist), f, C.c_float(kappa), blosum, s, p)
    return list(s), list(p)


def get_models(newick, rows, names, kappa, blosum=62) -> List[Model]:
    n = len(rows)
    m = (Model * n)()
    rc = lib().orc_get_models(newick.encode(), _strs(rows), _strs(names), n, C.c_float(kappa), blosum, m)
    if rc != 0:
        raise RuntimeError("orc_get_models failed")
    return m


def rev_aln(rows) -> List[str]:
    n = len(rows)
    out = (C.c_void_p * n)()
    lib().orc_rev_aln(_strs(rows), n, out)
    res = [C.string_at(out[i]).decode() for i in range(n)]
    libc = C.CDLL(None)
    for i in range(n):
        libc.free(C.c_void_p(out[i]))
    return res


def evd_fit(x) -> Tuple[int, float, float]:
    arr = np.ascontiguousarray(x, dtype=np.float64)
    mu, lam = C.c_double(), C.c_double()
    rc = lib().orc_evd_fit(arr.ctypes.data_as(C.POINTER(C.c_double)), len(arr), C.byref(mu), C.byref(lam))
    return rc, mu.value, lam.value


def pvalue(score, mu, lam) -> float:
    return float(lib().orc_pvalue(C.c_float(score), C.c_float(mu), C.c_float(lam)))


def tie_replacements(reset: bool = False) -> int:
    """How often getHSS's 1e-4 tie rule decided a replacement since the last reset (test diagnostic)."""
    return int(lib().orc_tie_replacements(1 if reset else 0))


def mt_stream(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_mt_stream(C.c_ulong(seed), out.ctypes.data_as(C.POINTER(C.c_uint)), n)
    return out


def simulate_null(newick, rows, names, freqs, kappa, seed) -> Tuple[List[str], int]:
    n = len(rows)
    out = (C.c_void_p * n)()
    clamped = C.c_long(0)
    f = (C.c_float * 4)(*freqs)
    rc = lib().orc_simulate_null(newick.encode(), _strs(rows), _strs(names), n, f, C.c_float(kappa),
                                 C.c_ulong(seed), out, C.byref(clamped))
    if rc != 0:
        raise RuntimeError("orc_simulate_null failed")
    res = [C.string_at(out[i]).decode() for i in range(n)]
    libc = C.CDLL(None)
    for i in range(n):
        libc.free(C.c_void_p(out[i]))
    return res, clamped.value


def score_matrix(rows, models, params) -> np.ndarray:
    """S[b][i] of one strand ((L+1) x (L+1), score.c:811-848)."""
    n = len(rows)
    L = C.c_int()
    ptr = lib().orc_score_matrix(_strs(rows), n, models, C.byref(params), C.byref(L))
    S = np.ctypeslib.as_array(ptr, shape=((L.value + 1), (L.value + 1))).copy()
    C.CDLL(None).free(ptr)
    return S


def score_aln(rows, ref_start, ref_length, models, modelsRev, params, want_sk=False, cap=4096):
    n = len(rows)
    hss = (Hss * cap)()
    skf, skr = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    cnt = lib().orc_score_aln(_strs(rows), n, ref_start, ref_length, models, modelsRev, C.byref(params), hss, cap,
                              C.byref(skf) if want_sk else None, C.byref(skr) if want_sk else None)
    out = [hss[i].as_dict() for i in range(min(cnt, cap))]
    if not want_sk:
        return out
    L = sum(1 for c in rows[0] if c != "-")
    shape = (n, 3, L + 1, L + 1)
    f = np.ctypeslib.as_array(skf, shape=shape).copy()
    r = np.ctypeslib.as_array(skr, shape=shape).copy()
    libc = C.CDLL(None)
    libc.free(skf)
    libc.free(skr)
    return out, f, r


def backtrack(b, i, sk, rows, params):
    n = len(rows)
    cols = len(rows[0])
    st = np.zeros((n, cols + 1), dtype=np.int32)
    z = np.zeros_like(st)
    tr = np.zeros_like(st)
    skc = np.ascontiguousarray(sk, dtype=np.float32)
    ip = C.POINTER(C.c_int)
    lib().orc_backtrack(b, i, skc.ctypes.data_as(C.POINTER(C.c_float)), _strs(rows), n, C.byref(params),
                        st.ctypes.data_as(ip), z.ctypes.data_as(ip), tr.ctypes.data_as(ip))
    return st, z, tr


@dataclass
class BlockResult:
    hss: list
    models: list
    modelsRev: list
    maxScores: np.ndarray
    mu: float
    lam: float
    evd_rc: int
    clamped: int


def run_block(rows, names, ref_start, ref_length, newick, kappa, params, seed_base=42, cap=4096) -> BlockResult:
    n = len(rows)
    m, mr = (Model * n)(), (Model * n)()
    hss = (Hss * cap)()
    mx = np.zeros(params.sampleN, dtype=np.float64)
    mu, lam = C.c_float(), C.c_float()
    rc, clamped = C.c_int(), C.c_long(0)
    cnt = lib().orc_run_block(_strs(rows), _strs(names), n, ref_start, ref_length, newick.encode(), C.c_float(kappa),
                              C.byref(params), C.c_ulong(seed_base), m, mr, hss, cap,
                              mx.ctypes.data_as(C.POINTER(C.c_double)), C.byref(mu), C.byref(lam), C.byref(rc),
                              C.byref(clamped))
    if cnt < 0:
        raise RuntimeError("orc_run_block failed")
    md = lambda a: [dict(dist=float(x.dist), kappa=float(x.kappa), freqs=list(x.freqs), scores=list(x.scores),
                         probs=list(x.probs)) for x in a]
    return BlockResult([hss[i].as_dict() for i in range(cnt)], md(m), md(mr), mx, float(mu.value), float(lam.value),
                       rc.value, clamped.value)
