"""Thin stand-alone driver around the HIP scoring path.

Mirrors the reference's command line (src/cmdline.ggo:6-20, src/RNAcode.c:236-373) for the
options that concern the scoring path and the listings:

    python -m rnacode_amd.cli [-n N] [-p CUTOFF] [-g | -t] [-b] [-r] [-s] [-m 62|90] [-c D,O,o,S]
                              [-e [-i CUTOFF] [-d DIR]] [-o OUT] [--trees SIDECAR] [FILE]

Tree + kappa per block (PhyML in the reference, RNAcode.c:153) come either from a sidecar
(`--trees`: one `<newick> TAB <kappa>` line per alignment block, in file order; `-` for blocks the
driver skips) or, without it, from the built-in estimator (rc_fit_trees_device, one wavefront per
block).  The blocks are scored on the GPU as a stream of sub-batches (--sub-blocks, rc_stream_*).  -e writes the reference's colored
alignment plots (src/postscript.c) as <DIR>/hss-<n>.eps.
Quirk kept from the reference: the 4th value of --pars goes to stopPenalty_0 (RNAcode.c:318)."""
from __future__ import annotations

import argparse
import os
import sys
import time
from typing import List, Optional

from . import api, eps, report
from .alnio import AlnBlock, read_alignment_file


def read_sidecar(path: str) -> List[Optional[tuple]]:
    out = []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if not line.strip():
                continue
            if line.strip() == "-":
                out.append(None)
                continue
            tree, kappa = line.split("\t")
            out.append((tree, float(kappa)))
    return out


def fit_trees(blocks, threads: int = 0, ctx: "Optional[api.Context]" = None) -> List[Optional[tuple]]:
    """Trees + kappas for every block the driver will score: on ctx's GPU (rc_fit_trees_device) when a
    context is given, else on host threads (rc_fit_trees)."""
    return api.fit_trees(blocks, threads, ctx=ctx)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="rnacode-hip", description="RNAcode scoring path on MI355X")
    ap.add_argument("file", nargs="?", help="alignment file (MAF or ClustalW); default stdin")
    ap.add_argument("-o", "--outfile")
    ap.add_argument("-g", "--gtf", action="store_true")
    ap.add_argument("-t", "--tabular", action="store_true")
    ap.add_argument("-b", "--best-only", action="store_true")
    ap.add_argument("-r", "--best-region", action="store_true")
    ap.add_argument("-s", "--stop-early", action="store_true")
    ap.add_argument("-n", "--num-samples", type=int, default=100)
    ap.add_argument("-p", "--cutoff", type=float, default=1.0)
    ap.add_argument("-c", "--pars")
    ap.add_argument("-m", "--blosum", type=int, default=62)
    ap.add_argument("-e", "--eps", action="store_true", help="Create colored plots in EPS format")
    ap.add_argument("-i", "--eps-cutoff", type=float, default=0.05, help="Create plots only if p better than this cutoff")
    ap.add_argument("-d", "--eps-dir", default="eps", help="Directory to put eps-files")
    ap.add_argument("-l", "--limit", help="limit to species: keep the rows whose name starts with one of these comma-separated strings")
    ap.add_argument("--trees", help="sidecar: one '<newick>\\t<kappa>' line per block (default: fit them)")
    ap.add_argument("--sub-blocks", type=int, default=0,
                    help="alignment blocks per sub-batch of the GPU stream (default: 2048, or 512 per distinct row count if that is more)")
    ap.add_argument("--seed-base", type=int, default=42)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)

    if a.blosum not in (62, 90):
        print("ERROR: Currently only BLOSUM62 and BLOSUM90 are supported.", file=sys.stderr)
        return 1
    kw = dict(sampleN=a.num_samples, cutoff=a.cutoff, stopEarly=int(a.stop_early), blosum=a.blosum, seed_base=a.seed_base)
    if a.pars:
        vals = [float(x) for x in a.pars.split(",")]
        for key, v in zip(("Delta", "Omega", "omega", "stopPenalty_0"), vals):
            kw[key] = v
    params = api.default_params(**kw)

    if a.file:
        blocks = read_alignment_file(a.file)
    else:
        import tempfile
        with tempfile.NamedTemporaryFile("w", delete=False) as fh:
            fh.write(sys.stdin.read())
        blocks = read_alignment_file(fh.name)
    blocks = [b.upper() for b in blocks]
    n_read = len(blocks)
    if a.limit:   # pruneAln (rnaz_utils.c:724-752, called at RNAcode.c:130-132): rows only, the columns stay
        keep = [x for x in a.limit.split(",") if x]
        kept = []
        for b in blocks:
            rows = [r for r in b.rows if any(r.name.startswith(x) for x in keep)]
            if not rows:   # (the reference dereferences the missing first row here)
                print("Skipping alignment. There must be at least three sequences in the alignment.", file=sys.stderr)
                continue
            kept.append(AlnBlock(rows, b.block_id, b.tree, b.kappa))
        blocks = kept
    ctx = api.Context(a.device)
    marshalled = api.Marshalled(blocks)   # one rc_block array for the tree fit and the batch
    if a.trees:
        side = read_sidecar(a.trees)
        if len(side) != len(blocks):
            print(f"ERROR: {len(blocks)} alignment blocks but {len(side)} sidecar entries", file=sys.stderr)
            ctx.close()
            return 1
    else:
        side = fit_trees(marshalled, ctx=ctx)
    prepared: List[AlnBlock] = []
    for b, s in zip(blocks, side):
        # no tree: either a block the driver skips anyway (N <= 2, too short) or one whose tree could not be built;
        # the library leaves the latter out with a per-block status, the other blocks are scored (RNAcode.c:153-156)
        b.tree, b.kappa = s if s is not None else (None, None)
        prepared.append(b)

    out = open(a.outfile, "w") if a.outfile else sys.stdout
    fmt = 2 if a.tabular else (1 if a.gtf else 0)
    t0 = time.perf_counter()
    # the blocks go through the GPU as a stream of sub-batches (rc_stream_*): while one is being scored the next is prepared
    # on the host threads, and the listing of a finished one is written meanwhile (the reference's loop, RNAcode.c:115-221)
    marshalled.set_trees(strict=False)
    st = report.ReportState()
    base = 0
    # sub-batch sizes: --sub-blocks, or the library's schedule (rc_stream_plan: a small first sub-batch, then doubling, whole rounds of
    # the chip; every row count is a launch of its own, so more classes mean larger sub-batches)
    sub_blocks = a.sub_blocks if a.sub_blocks > 0 else 0
    for batch in api.score_stream(ctx, marshalled, params, sub_blocks, depth=3):
        all_hss = batch.scoreAln_all()
        for i in range(batch.n):
            b = prepared[base + i]
            code = batch.status(i)
            if code == api.RC_ERR_SKIP:   # RNAcode.c:142-150
                msg = "There must be at least three sequences in the alignment." if b.n <= 2 else "Too short."
                print(f"Skipping alignment. {msg}", file=sys.stderr)
                continue
            if code != api.RC_OK:         # RNAcode.c:153-156: the reference has no tree for this block either
                print(f"Skipping alignment. Failed to build ML tree. ({batch.block_error(i) or 'not scored'})", file=sys.stderr)
                continue
            hook = None
            if a.eps:   # misc.c:461-474: hss-<counter>.eps for every listed HSS with p below the plot cutoff
                def hook(counter, h, i=i, b=b, batch=batch):
                    os.makedirs(a.eps_dir, exist_ok=True)
                    text = eps.color_aln(b, h, lambda strand, lo, hi: batch.backtrack(i, 0 if strand == "+" else 1, lo, hi), a.blosum)
                    with open(os.path.join(a.eps_dir, f"hss-{counter}.eps"), "w") as fh:
                        fh.write(text)
            report.print_results(out, fmt, all_hss[i], b.rows[0].name, st, cutoff=a.cutoff, best_only=a.best_only,
                                 best_region=a.best_region, eps=hook, eps_cutoff=a.eps_cutoff)
        base += batch.n
        batch.close()
    if fmt == 0:
        report.print_footer(out, n_read, time.perf_counter() - t0, params.sampleN, params.Delta, params.Omega,
                            params.omega, params.stopPenalty_k)
    ctx.close()
    if a.outfile:
        out.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
