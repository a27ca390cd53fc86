"""Host-side mirror of RNAcode's scoring interface on top of librnacode_hip.so (ctypes).

The reference is compiled C with no Python layer; this module exists so that tests, the
benchmark and the multi-GPU launcher can drive the C-ABI of include/rnacode_hip.h.  Names
follow the reference's score.h surface (src/score.h:83-118):

    getModels(block)                       -> rc_batch_models
    scoreAln(block)                        -> native HSS list      (score.c:1067-1147)
    getExtremeValuePars(block)             -> (rc, mu, lambda)     (score.c:976-1064)
    backtrack(block, strand, b, i)         -> states / z / transitions (score.c:558-797)

There is no CPU implementation behind these calls: if the HIP library or a GPU is missing
they raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .alnio import AlnBlock

# the library asks the HIP runtime for eight hardware queues (rc_context.cpp, want_hw_queues); in a Python process torch may start the runtime
# first, so the request is made here as well (an explicit setting wins)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
# RC_LIB_PATH: load another build of the library (tools/ab.sh compares builds without touching the product .so)
LIB_PATH = os.environ.get("RC_LIB_PATH") or os.path.join(_HERE, "librnacode_hip.so")

RC_OK, RC_ERR_ARG, RC_ERR_DEVICE, RC_ERR_UNSUPPORTED, RC_ERR_SKIP = 0, -1, -2, -3, -4


class RcParams(C.Structure):
    _fields_ = [("Delta", C.c_float), ("Omega", C.c_float), ("omega", C.c_float),
                ("stopPenalty_0", C.c_float), ("stopPenalty_k", C.c_float), ("blosum", C.c_int32),
                ("sampleN", C.c_int32), ("cutoff", C.c_float), ("stopEarly", C.c_int32), ("seed_base", C.c_uint32)]


class RcBlock(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("n_cols", C.c_int32), ("rows", C.POINTER(C.c_char_p)),
                ("names", C.POINTER(C.c_char_p)), ("ref_start", C.c_int32), ("ref_length", C.c_int32),
                ("newick", C.c_char_p), ("kappa", C.c_float)]


class RcHss(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("startGenomic", C.c_int32), ("endGenomic", C.c_int32),
                ("startSite", C.c_int32), ("endSite", C.c_int32), ("strand", C.c_int32), ("frame", C.c_int32),
                ("score", C.c_float), ("pvalue", C.c_float)]

    def as_dict(self):
        return dict(strand=chr(self.strand), frame=self.frame, startSite=self.startSite, endSite=self.endSite,
                    start=self.start, end=self.end, startGenomic=self.startGenomic, endGenomic=self.endGenomic,
                    score=float(np.float32(self.score)), pvalue=float(np.float32(self.pvalue)))


_HSS_DTYPE = np.dtype([(name, "<i4") for name in ("start", "end", "startGenomic", "endGenomic", "startSite", "endSite", "strand", "frame")]
                      + [("score", "<f4"), ("pvalue", "<f4")])


class RcModel(C.Structure):
    _fields_ = [("scores", C.c_float * 4), ("probs", C.c_float * 4), ("kappa", C.c_float), ("dist", C.c_float),
                ("freqs", C.c_float * 4)]

    def as_dict(self):
        return dict(dist=float(self.dist), kappa=float(self.kappa), freqs=list(self.freqs),
                    scores=list(self.scores), probs=list(self.probs))


class RnacodeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librnacode_hip error {code}: {msg}")
        self.code = code


def build_library(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcdir = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", srcdir, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", srcdir])
    return LIB_PATH


_lib = None


def lib():
    """Load librnacode_hip.so.  Fails loudly if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RnacodeError(RC_ERR_DEVICE, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        l = C.CDLL(LIB_PATH)
        l.rc_last_error.restype = C.c_char_p
        l.rc_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        l.rc_ctx_destroy.argtypes = [C.c_void_p]
        l.rc_ctx_trim.argtypes = [C.c_void_p]
        l.rc_ctx_trim.restype = None
        l.rc_batch_create.argtypes = [C.c_void_p, C.POINTER(RcBlock), C.c_int32, C.POINTER(RcParams), C.POINTER(C.c_void_p)]
        l.rc_batch_destroy.argtypes = [C.c_void_p]
        l.rc_batch_run.argtypes = [C.c_void_p]
        l.rc_batch_run_async.argtypes = [C.c_void_p]
        l.rc_batch_wait.argtypes = [C.c_void_p]
        l.rc_batch_size.argtypes = [C.c_void_p]
        l.rc_batch_block_error.argtypes = [C.c_void_p, C.c_int32]
        l.rc_batch_block_error.restype = C.c_char_p
        l.rc_batch_prep_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        l.rc_ctx_set_host_threads.argtypes = [C.c_void_p, C.c_int32]
        l.rc_ctx_host_threads.argtypes = [C.c_void_p]
        l.rc_ctx_fit_exp_mode.argtypes = [C.c_void_p]
        l.rc_stream_create.argtypes = [C.c_void_p, C.POINTER(RcParams), C.c_int32, C.POINTER(C.c_void_p)]
        l.rc_stream_submit.argtypes = [C.c_void_p, C.POINTER(RcBlock), C.c_int32]
        l.rc_stream_submit_bound.argtypes = [C.c_void_p, C.POINTER(RcBlock), C.c_int32, C.c_void_p]
        l.rc_stream_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        l.rc_stream_recycle.argtypes = [C.c_void_p, C.c_void_p]
        l.rc_stream_recycle.restype = None
        l.rc_stream_pending.argtypes = [C.c_void_p]
        l.rc_stream_plan.argtypes = [C.c_void_p, C.POINTER(RcParams), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
        l.rc_stream_destroy.argtypes = [C.c_void_p]
        l.rc_stream_destroy.restype = None
        l.rc_batch_bind_maxima.argtypes = [C.c_void_p, C.c_void_p]
        l.rc_batch_work.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        l.rc_batch_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        l.rc_batch_status.argtypes = [C.c_void_p, C.c_int32]
        l.rc_batch_null_kernel.argtypes = [C.c_void_p]
        l.rc_batch_null_kernel.restype = C.c_char_p
        l.rc_batch_models.argtypes = [C.c_void_p, C.c_int32, C.POINTER(RcModel), C.POINTER(RcModel)]
        l.rc_batch_maxima.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float)]
        l.rc_batch_maxima_all.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.rc_batch_fit.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.rc_batch_fit_all.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.rc_batch_hss_all.argtypes = [C.c_void_p, C.POINTER(RcHss), C.c_int64, C.POINTER(C.c_int64)]
        l.rc_batch_hss.argtypes = [C.c_void_p, C.c_int32, C.POINTER(RcHss), C.c_int32]
        l.rc_batch_clamped.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        l.rc_batch_native_S.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32]
        l.rc_batch_backtrack.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.rc_evd_fit.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        l.rc_mt_stream.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_int32]
        l.rc_pvalue.argtypes = [C.c_float, C.c_float, C.c_float]
        l.rc_pvalue.restype = C.c_float
        l.rc_fit_tree.argtypes = [C.POINTER(RcBlock), C.c_char_p, C.c_int32, C.POINTER(C.c_float)]
        l.rc_fit_trees.argtypes = [C.POINTER(RcBlock), C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_float), C.c_int32]
        l.rc_tree_lnl.argtypes = [C.POINTER(RcBlock), C.POINTER(C.c_double)]
        l.rc_fit_trees_device.argtypes = [C.c_void_p, C.POINTER(RcBlock), C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_float),
                                          C.POINTER(C.c_double)]
        l.rc_code_tables.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.rc_set_stream_cache.argtypes = [C.c_int]
        l.rc_set_stream_cache.restype = None
        _lib = l
    return _lib


EXPORTED_SYMBOLS = [
    "rc_default_params", "rc_last_error", "rc_device_count", "rc_ctx_create", "rc_ctx_destroy", "rc_ctx_trim", "rc_batch_create",
    "rc_batch_destroy", "rc_batch_bind_maxima", "rc_batch_run", "rc_batch_run_async", "rc_batch_wait", "rc_batch_size", "rc_batch_block_error",
    "rc_batch_prep_timing", "rc_host_cpus", "rc_ctx_set_host_threads", "rc_ctx_host_threads", "rc_ctx_fit_exp_mode", "rc_stream_create", "rc_stream_submit", "rc_stream_submit_bound",
    "rc_stream_next", "rc_stream_recycle", "rc_stream_pending", "rc_stream_plan", "rc_stream_destroy", "rc_set_stream_cache", "rc_batch_work", "rc_batch_timing", "rc_batch_null_kernel", "rc_batch_status",
    "rc_batch_models", "rc_batch_maxima", "rc_batch_maxima_all", "rc_batch_fit", "rc_batch_fit_all", "rc_batch_hss", "rc_batch_hss_all", "rc_batch_clamped",
    "rc_batch_native_S", "rc_batch_backtrack", "rc_fit_tree", "rc_fit_trees", "rc_fit_trees_device", "rc_tree_lnl", "rc_evd_fit", "rc_pvalue", "rc_mt_stream", "rc_code_tables",
]


def _check(code: int) -> int:
    if code < 0:
        raise RnacodeError(code, lib().rc_last_error().decode())
    return code


def default_params(**kw) -> RcParams:
    p = RcParams()
    lib().rc_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def pvalue(score: float, mu: float, lam: float) -> float:
    """RNAcode.c:182 with its float/double promotions (rc_pvalue)."""
    return float(lib().rc_pvalue(score, mu, lam))


def code_tables(blosum: int = 62) -> Tuple[np.ndarray, np.ndarray]:
    """(pep[64], matrix[20][20]) of rc_code_tables: genetic code (-1 = stop) and BLOSUM62/90 as the scorer uses them."""
    pep = np.zeros(64, dtype=np.int32)
    mat = np.zeros((20, 20), dtype=np.int32)
    ip = C.POINTER(C.c_int32)
    _check(lib().rc_code_tables(blosum, pep.ctypes.data_as(ip), mat.ctypes.data_as(ip)))
    return pep, mat


def fit_tree(block: AlnBlock) -> Tuple[str, float]:
    """Tree + kappa of one block (what treeML hands to the scorer, src/treeML.c:35-152); host code."""
    rows = (C.c_char_p * block.n)(*[r.seq.encode() for r in block.rows])
    names = (C.c_char_p * block.n)(*[r.name.encode() for r in block.rows])
    blk = RcBlock(block.n, block.cols, rows, names, block.rows[0].start, block.rows[0].length, None, 0.0)
    buf = C.create_string_buffer(1 << 16)
    kappa = C.c_float()
    _check(lib().rc_fit_tree(C.byref(blk), buf, len(buf), C.byref(kappa)))
    return buf.value.decode(), float(kappa.value)


class Marshalled:
    """The rc_block array of a list of AlnBlocks (row and name pointers), built once and shared by the tree fit and
    the batch: building it is the larger part of both calls' Python time."""

    def __init__(self, blocks: Sequence[AlnBlock]):
        self.blocks = list(blocks)
        n = len(self.blocks)
        self.arr = (RcBlock * max(n, 1))()
        self.keep = []
        for i, b in enumerate(self.blocks):
            rows = (C.c_char_p * b.n)(*[r.seq.encode() for r in b.rows])
            names = (C.c_char_p * b.n)(*[r.name.encode() for r in b.rows])
            self.keep.append((rows, names))
            self.arr[i] = RcBlock(b.n, b.cols, rows, names, b.rows[0].start, b.rows[0].length, None, 0.0)

    def set_trees(self, strict: bool = True):
        """Copy tree and kappa of the AlnBlocks into the array (after they were fitted or read from a sidecar).
        A block without a tree stays without one: the library leaves it out (RC_ERR_ARG, "no tree") like the
        reference's driver skips a block whose tree fit failed (RNAcode.c:153-156)."""
        self.trees = []
        for i, b in enumerate(self.blocks):
            if b.tree is None or b.kappa is None:
                if strict:
                    raise ValueError(f"block {b.block_id}: tree and kappa are required (sidecar or fitted upstream)")
                self.trees.append(None)
                self.arr[i].newick = None
                self.arr[i].kappa = 0.0
                continue
            t = b.tree.encode()
            self.trees.append(t)
            self.arr[i].newick = t
            self.arr[i].kappa = b.kappa


def fit_trees(blocks, threads: int = 0, cap: int = 0, ctx: "Optional[Context]" = None,
              lnl: Optional[list] = None) -> List[Optional[Tuple[str, float]]]:
    """Trees + kappas of many blocks (a sequence of AlnBlocks or a Marshalled); None for blocks the driver skips.
    With ctx: rc_fit_trees_device (one wavefront per block on that context's GPU; `lnl`, if a list, receives the
    log-likelihoods); without: rc_fit_trees on host threads."""
    m = blocks if isinstance(blocks, Marshalled) else Marshalled(blocks)
    n = len(m.blocks)
    arr = m.arr
    if cap <= 0:   # room for the widest block's Newick text (name + ":0.123456" + brackets per node)
        cap = max(1 << 14, 96 * max((b.n for b in m.blocks), default=0))
    buf = C.create_string_buffer(max(n, 1) * cap)
    kap = (C.c_float * max(n, 1))()
    if ctx is not None:
        ll = (C.c_double * max(n, 1))()
        _check(lib().rc_fit_trees_device(ctx._h, arr, n, buf, cap, kap, ll))
        if lnl is not None:
            lnl[:] = [float(x) for x in ll[:n]]
    else:
        _check(lib().rc_fit_trees(arr, n, buf, cap, kap, threads))
    out = []
    raw = buf.raw   # one copy (buf.raw copies the whole buffer on every access)
    for i in range(n):
        s = raw[i * cap:(i + 1) * cap].split(b"\0", 1)[0].decode()
        out.append((s, float(kap[i])) if s else None)
    return out


def tree_lnl(block: AlnBlock, newick: str, kappa: float) -> float:
    """HKY85 log-likelihood of a given tree + kappa on the block (same model/data handling as fit_tree)."""
    rows = (C.c_char_p * block.n)(*[r.seq.encode() for r in block.rows])
    names = (C.c_char_p * block.n)(*[r.name.encode() for r in block.rows])
    blk = RcBlock(block.n, block.cols, rows, names, block.rows[0].start, block.rows[0].length, newick.encode(), kappa)
    out = C.c_double()
    _check(lib().rc_tree_lnl(C.byref(blk), C.byref(out)))
    return out.value


class Context:
    """One per process and GPU (rc_ctx)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().rc_ctx_create(device, C.byref(self._h)))
        self.device = device

    def trim(self):
        """Give back the buffers of destroyed batches and streams the context keeps for the next ones (rc_ctx_trim)."""
        lib().rc_ctx_trim(self._h)

    def close(self):
        if self._h:
            lib().rc_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_host_threads(self, threads: int):
        """Host threads that prepare blocks (rc_ctx_set_host_threads)."""
        _check(lib().rc_ctx_set_host_threads(self._h, threads))
        return self

    @property
    def host_threads(self) -> int:
        return lib().rc_ctx_host_threads(self._h)

    @property
    def fit_exp_mode(self) -> str:
        """Which exp() the EVD fit uses (rc_ctx_fit_exp_mode): the host glibc's generic or fused variant, or the device library's."""
        return {0: "device", 1: "generic", 2: "fused"}[_check(lib().rc_ctx_fit_exp_mode(self._h))]

    def evd_fit(self, x) -> Tuple[int, float, float]:
        """EVDMaxLikelyFit (extreme_fit.c:157-251) on the device."""
        arr = np.ascontiguousarray(x, dtype=np.float64)
        mu, lam = C.c_double(), C.c_double()
        rc = _check(lib().rc_evd_fit(self._h, arr.ctypes.data_as(C.POINTER(C.c_double)), len(arr), C.byref(mu), C.byref(lam)))
        return rc, mu.value, lam.value

    def mt_stream(self, seed: int, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.uint32)
        _check(lib().rc_mt_stream(self._h, seed, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out


@dataclass
class BlockScores:
    """What RNAcode.c:164-188 has for one block before printing."""
    status: int
    hss: List[dict]
    evd_rc: int = -1
    mu: float = 0.0
    lam: float = 0.0
    maxScores: Optional[np.ndarray] = None


class Batch:
    """Alignment blocks resident in HBM (rc_batch)."""

    def __init__(self, ctx: Context, blocks, params: RcParams):
        """blocks: a sequence of AlnBlocks with tree and kappa set, or a Marshalled of such blocks."""
        self.ctx = ctx
        self.params = params
        self._stream = None
        m = blocks if isinstance(blocks, Marshalled) else Marshalled(blocks)
        m.set_trees(strict=False)
        self._keep = m
        self.blocks = m.blocks
        n = len(self.blocks)
        self._h = C.c_void_p()
        _check(lib().rc_batch_create(ctx._h, m.arr, n, C.byref(params), C.byref(self._h)))
        self.n = n

    @classmethod
    def _from_stream(cls, stream: "Stream", handle, blocks):
        b = cls.__new__(cls)
        b.ctx, b.params, b._stream, b._keep = stream.ctx, stream.params, stream, None
        b.blocks = blocks
        b._h = handle
        b.n = lib().rc_batch_size(handle)
        return b

    def close(self):
        if self._h:
            if self._stream is not None and self._stream._h:
                lib().rc_stream_recycle(self._stream._h, self._h)   # buffers go back to the stream
            else:
                lib().rc_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind_maxima(self, device_ptr: int):
        """Write maxima into a caller-owned device buffer (e.g. torch tensor .data_ptr())."""
        _check(lib().rc_batch_bind_maxima(self._h, C.c_void_p(device_ptr)))
        return self

    def run(self):
        _check(lib().rc_batch_run(self._h))
        return self

    def run_async(self):
        _check(lib().rc_batch_run_async(self._h))
        return self

    def wait(self):
        _check(lib().rc_batch_wait(self._h))
        return self

    def block_error(self, blk: int) -> str:
        """Why a block was left out (status RC_ERR_ARG / RC_ERR_UNSUPPORTED), '' otherwise."""
        return lib().rc_batch_block_error(self._h, blk).decode()

    def prep_timing(self):
        """(host preparation ms, device table kernels ms, bytes copied to the device)."""
        h, k, by = C.c_double(), C.c_float(), C.c_int64()
        _check(lib().rc_batch_prep_timing(self._h, C.byref(h), C.byref(k), C.byref(by)))
        return h.value, float(k.value), by.value

    def work(self) -> Tuple[int, int]:
        a, c = C.c_int64(), C.c_int64()
        _check(lib().rc_batch_work(self._h, C.byref(a), C.byref(c)))
        return a.value, c.value

    def timing(self):
        t = (C.c_float * 5)()
        nl = (C.c_int32 * 5)()
        _check(lib().rc_batch_timing(self._h, t, nl))
        keys = ("total", "mt_stream", "null", "native", "evd_fit")
        return {k: float(t[i]) for i, k in enumerate(keys)}, {k: int(nl[i]) for i, k in enumerate(keys)}

    def null_kernel(self) -> str:
        """The k_null instantiation behind most of the last run's sampling, as rocprofv3 names it."""
        return lib().rc_batch_null_kernel(self._h).decode()

    def status(self, blk: int) -> int:
        return lib().rc_batch_status(self._h, blk)

    # ---- score.h-shaped accessors -------------------------------------------------------
    def getModels(self, blk: int):
        n = self.blocks[blk].n
        f, r = (RcModel * n)(), (RcModel * n)()
        _check(lib().rc_batch_models(self._h, blk, f, r))
        return [m.as_dict() for m in f], [m.as_dict() for m in r]

    def scoreAln(self, blk: int, cap: int = 4096) -> List[dict]:
        out = (RcHss * cap)()
        n = _check(lib().rc_batch_hss(self._h, blk, out, cap))
        return [out[i].as_dict() for i in range(min(n, cap))]

    def scoreAln_all(self) -> List[List[dict]]:
        """rc_batch_hss_all: the HSS lists of all blocks with two library calls (sizing, then the records)."""
        offs = (C.c_int64 * (self.n + 1))()
        _check(lib().rc_batch_hss_all(self._h, None, 0, offs))
        total = int(offs[self.n])
        out = (RcHss * max(total, 1))()
        _check(lib().rc_batch_hss_all(self._h, out, total, offs))
        a = np.frombuffer(out, dtype=_HSS_DTYPE, count=total)   # columns at once: per-record ctypes access is ~3 us
        keys = ("strand", "frame", "startSite", "endSite", "start", "end", "startGenomic", "endGenomic", "score", "pvalue")
        cols = [[chr(v) for v in a["strand"].tolist()]] + [a[k].tolist() for k in keys[1:8]] + \
               [a["score"].astype(np.float64).tolist(), a["pvalue"].astype(np.float64).tolist()]
        recs = [dict(zip(keys, vals)) for vals in zip(*cols)]
        o = list(offs)
        return [recs[o[i]:o[i + 1]] for i in range(self.n)]

    def getExtremeValuePars(self, blk: int) -> Tuple[int, float, float]:
        rc, mu, lam = C.c_int32(), C.c_float(), C.c_float()
        _check(lib().rc_batch_fit(self._h, blk, C.byref(rc), C.byref(mu), C.byref(lam)))
        return rc.value, float(mu.value), float(lam.value)

    def fits(self) -> np.ndarray:
        """rc_batch_fit_all: [n_blocks, 4] float32 rows (evd_rc or status, mu, lambda, samples above the best native score)."""
        out = np.zeros((self.n, 4), dtype=np.float32)
        _check(lib().rc_batch_fit_all(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def maxScores(self, blk: int) -> np.ndarray:
        out = np.zeros(self.params.sampleN, dtype=np.float32)
        _check(lib().rc_batch_maxima(self._h, blk, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def maxScores_all(self) -> np.ndarray:
        out = np.zeros((self.n, self.params.sampleN), dtype=np.float32)
        if self.n:
            _check(lib().rc_batch_maxima_all(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def clamped(self) -> int:
        c = C.c_int64()
        _check(lib().rc_batch_clamped(self._h, C.byref(c)))
        return c.value

    def native_S(self, blk: int, strand: int, frame: int) -> np.ndarray:
        L = self.blocks[blk].ref_len
        sites = (L - frame) // 3
        out = np.zeros(max(sites * sites, 1), dtype=np.float32)
        got = _check(lib().rc_batch_native_S(self._h, blk, strand, frame, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out[: got * got].reshape(got, got)

    def backtrack(self, blk: int, strand: int, b: int, i: int):
        blkobj = self.blocks[blk]
        shape = (blkobj.n, blkobj.cols + 1)
        st, z, tr = (np.zeros(shape, dtype=np.int32) for _ in range(3))
        ip = C.POINTER(C.c_int32)
        _check(lib().rc_batch_backtrack(self._h, blk, strand, b, i, st.ctypes.data_as(ip), z.ctypes.data_as(ip), tr.ctypes.data_as(ip)))
        return st, z, tr

    def results(self, blk: int, with_maxima: bool = False) -> BlockScores:
        st = self.status(blk)
        if st != RC_OK:
            return BlockScores(st, [])
        rc, mu, lam = self.getExtremeValuePars(blk)
        return BlockScores(st, self.scoreAln(blk), rc, mu, lam, self.maxScores(blk) if with_maxima else None)


class Stream:
    """A stream of batches (rc_stream): submit() prepares a slice of blocks on the host threads and queues its
    copy and launches; next() hands over the oldest finished batch.  While the GPU scores one batch the caller
    submits the next, so host preparation, PCIe and kernels overlap (the reference's per-block loop,
    RNAcode.c:115-221, as a pipeline)."""

    def __init__(self, ctx: Context, params: RcParams, depth: int = 3):
        self.ctx, self.params, self.depth = ctx, params, depth
        self._h = C.c_void_p()
        self._blocks = []   # block lists of the batches in flight, oldest first
        _check(lib().rc_stream_create(ctx._h, C.byref(params), depth, C.byref(self._h)))

    def submit(self, m: Marshalled, lo: int = 0, hi: Optional[int] = None, maxima_ptr: int = 0):
        """Submit blocks [lo, hi) of a Marshalled whose trees are set (set_trees).  maxima_ptr: device address that
        receives this batch's [hi - lo][sampleN] per-sample maxima (rc_stream_submit_bound)."""
        hi = len(m.blocks) if hi is None else hi
        arr = C.cast(C.byref(m.arr, lo * C.sizeof(RcBlock)), C.POINTER(RcBlock))
        _check(lib().rc_stream_submit_bound(self._h, arr, hi - lo, C.c_void_p(maxima_ptr or None)))
        self._blocks.append(m.blocks[lo:hi])

    @property
    def pending(self) -> int:
        return lib().rc_stream_pending(self._h)

    def next(self) -> Batch:
        h = C.c_void_p()
        r = lib().rc_stream_next(self._h, C.byref(h))
        blocks = self._blocks.pop(0) if self._blocks else []   # the C side has taken the oldest batch off its queue, whatever the outcome
        _check(r)
        return Batch._from_stream(self, h, blocks)

    def close(self):
        if self._h:
            lib().rc_stream_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stream_plan(ctx: Context, params: RcParams, n_blocks: int, row_classes: int = 1) -> List[int]:
    """rc_stream_plan: sub-batch sizes for streaming n_blocks blocks (small first, then doubling, whole rounds of the chip)."""
    sizes = (C.c_int32 * 64)()
    k = _check(lib().rc_stream_plan(ctx._h, C.byref(params), n_blocks, row_classes, sizes, 64))
    return [int(sizes[i]) for i in range(k)]


def score_stream(ctx: Context, m: Marshalled, params: RcParams, sub_blocks=0, depth: int = 3, stream: Optional[Stream] = None,
                 maxima_ptr: int = 0, ramp: bool = True):
    """Score the blocks of `m` as a pipeline of sub-batches.  Yields finished Batches in submission order; the consumer
    closes (recycles) each one.  sub_blocks: 0 -> the library's schedule (rc_stream_plan); a list -> these sizes; a
    number -> sub-batches of that many blocks (with ramp: the first two a quarter and a half of it).  maxima_ptr:
    device address of a [len(m.blocks)][sampleN] float32 buffer that receives every block's per-sample maxima."""
    own = stream is None
    s = stream or Stream(ctx, params, depth)
    n = len(m.blocks)
    if isinstance(sub_blocks, (list, tuple)):
        sizes = list(sub_blocks)
    elif not sub_blocks:
        sizes = stream_plan(ctx, params, n, len({b.n for b in m.blocks}) or 1)
    else:
        sizes, left = [], n
        while left > 0:
            size = sub_blocks
            if ramp and len(sizes) < 2 and sub_blocks >= 256:
                size = sub_blocks // (4 >> len(sizes))
            size = min(size, left)
            sizes.append(size)
            left -= size
    assert sum(sizes) == n and all(x > 0 for x in sizes)
    try:
        lo = 0
        sent = 0
        while sent < len(sizes) or s.pending:
            while sent < len(sizes) and s.pending < s.depth:
                hi = lo + sizes[sent]
                s.submit(m, lo, hi, maxima_ptr + 4 * lo * params.sampleN if maxima_ptr else 0)
                lo = hi
                sent += 1
            yield s.next()
    finally:
        if own:
            s.close()


def score_blocks(blocks: Sequence[AlnBlock], device: int = 0, **params) -> List[BlockScores]:
    """Convenience: one context, one batch, everything RNAcode.c:164-188 computes per block."""
    ctx = Context(device)
    batch = None
    try:
        batch = Batch(ctx, blocks, default_params(**params)).run()
        return [batch.results(i, with_maxima=True) for i in range(batch.n)]
    finally:
        if batch is not None:
            batch.close()   # before the context: a batch must not outlive it
        ctx.close()
