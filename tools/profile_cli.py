"""Where the stand-alone driver spends its wall time on a synthetic MAF (stage by stage)."""
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rnacode_amd import api, cli, report  # noqa: E402
from rnacode_amd.alnio import read_alignment_file  # noqa: E402
from rnacode_amd.synth import synth_blocks, to_maf  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = tempfile.mkdtemp()
maf = os.path.join(d, "in.maf")
with open(maf, "w") as fh:
    fh.write(to_maf(synth_blocks(nb, 6, 120, seed=1)))
ctx = api.Context(0)
api.fit_trees(list(read_alignment_file(maf))[:4], ctx=ctx)   # warm-up
T = {}
t = time.time(); blocks = [b.upper() for b in read_alignment_file(maf)]; T["read MAF"] = time.time() - t
t = time.time(); side = cli.fit_trees(blocks, ctx=ctx); T["fit trees (GPU)"] = time.time() - t
for b, s in zip(blocks, side):
    b.tree, b.kappa = s
t = time.time(); batch = api.Batch(ctx, blocks, api.default_params(sampleN=1000)); T["models + upload"] = time.time() - t
t = time.time(); batch.run(); T["score (GPU)"] = time.time() - t
t = time.time()
out = io.StringIO(); st = report.ReportState()
all_hss = batch.scoreAln_all()
T["fetch HSS"] = time.time() - t
t = time.time()
for i, b in enumerate(blocks):
    report.print_results(out, 2, all_hss[i], b.rows[0].name, st)
T["listing"] = time.time() - t
print(f"{nb} blocks: " + ", ".join(f"{k} {v:.3f} s" for k, v in T.items()) + f"; total {sum(T.values()):.3f} s")
