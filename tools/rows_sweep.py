#!/usr/bin/env python3
"""The whole row-count axis: uniform resident batches of N rows x C columns at n = 1000, enough blocks to fill the
chip several times over; per shape the sampling time, units/s, the FP32 issue fraction by the algorithmic flops, the kernel
that ran with its registers / scratch / wavefronts per SIMD from the code object's metadata (tools/kernel_resources.py), and the same
shape forced through the wide-block kernels (RC_GENERIC_MIN_ROWS=3).

  tools/rows_sweep.py [--rows 3,4,...] [--cols 150,300] [--out profiles/r06/rows_sweep.json] [--no-generic] [--target-ms 150]

Every (mode, shape) runs in this process on contexts of its own (the switch is read when a context is created)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from kernel_resources import kernel_resources  # noqa: E402
from rnacode_amd import api  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

ROWS = [3, 4, 6, 8, 10, 13, 16, 20, 24, 28, 32, 33, 40, 48, 56, 64, 65, 80, 100]
PEAK = bench.VALU_PEAK_GOPS * 1e9


def shape_blocks(rows, cols, target_ms, distinct=48):
    """Blocks for about target_ms of sampling at a quarter of the issue roof: `distinct` different blocks, repeated."""
    base = [b.upper() for b in synth_blocks(distinct, rows, cols, seed=7 + rows)]
    _, fl_one = bench.algorithmic_work(base, 1000)
    fl_one /= len(base)
    want = int(target_ms * 1e-3 * 0.25 * PEAK / fl_one)
    want = max(256, min(12000, want))
    return [base[i % distinct] for i in range(want)]


def run_shape(ctx, rows, cols, target_ms, n=1000):
    blocks = shape_blocks(rows, cols, target_ms)
    p = api.default_params(sampleN=n, seed_base=42)
    batch = api.Batch(ctx, blocks, p)
    batch.run()
    ts, nulls = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        batch.run()
        ts.append((time.perf_counter() - t0) * 1e3)
        nulls.append(batch.timing()[0]["null"])
    _, fl = bench.algorithmic_work(blocks, n)
    kern = batch.null_kernel()
    ms, null_ms = sorted(ts)[1], sorted(nulls)[1]
    mx = batch.maxScores_all()
    digest = float(mx[: min(len(mx), 48)].astype("float64").sum())
    batch.close()
    return dict(rows=rows, cols=cols, blocks=len(blocks), ms=ms, null_ms=null_ms, units_per_s=len(blocks) * cols * n / (ms * 1e-3),
                issue_frac=fl / (null_ms * 1e-3) / PEAK, kernel=kern, maxima_digest=digest)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default=",".join(str(r) for r in ROWS))
    ap.add_argument("--cols", default="150,300")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-generic", action="store_true")
    ap.add_argument("--target-ms", type=float, default=150.0)
    ap.add_argument("--env", default="", help="KEY=VALUE,... set for the default-mode contexts")
    a = ap.parse_args()
    rows = [int(x) for x in a.rows.split(",")]
    cols = [int(x) for x in a.cols.split(",")]
    res = kernel_resources()
    api.lib().rc_set_stream_cache(0)
    out = []
    modes = [("default", dict(kv.split("=") for kv in a.env.split(",") if kv))]
    if not a.no_generic:
        modes.append(("generic", {"RC_GENERIC_MIN_ROWS": "3"}))
    for mode, env in modes:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        ctx = api.Context(0)
        for c in cols:
            for r in rows:
                if mode == "generic" and r > 64:
                    continue   # the default mode already ran them there
                rec = run_shape(ctx, r, c, a.target_ms)
                rec["mode"] = mode
                k = res.get(rec["kernel"]) or res.get(rec["kernel"].replace("rc::", "rc::", 1)) or {}
                rec.update({x: k.get(x) for x in ("vgpr", "agpr", "scratch_bytes", "spill_vgpr", "waves_per_simd")})
                out.append(rec)
                print(f"{mode:8s} {r:3d} x {c:3d}: {rec['blocks']:5d} blocks {rec['ms']:8.2f} ms (sampling {rec['null_ms']:8.2f}) {rec['units_per_s']:.3e} units/s "
                      f"issue {rec['issue_frac']:.3f}  {rec['kernel']:42s} vgpr {rec['vgpr']} scratch {rec['scratch_bytes']} waves/SIMD {rec['waves_per_simd']}", flush=True)
        ctx.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    # the two modes must agree bit for bit on what they computed
    by = {}
    for rec in out:
        by.setdefault((rec["rows"], rec["cols"]), []).append(rec)
    bad = [k for k, v in by.items() if len({x["maxima_digest"] for x in v}) > 1]
    print("default and generic modes agree on the maxima of every shape" if not bad else f"MISMATCH between modes at {bad}")
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
