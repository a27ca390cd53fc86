#!/usr/bin/env python3
"""Work saved by --stop-early (score.c:992,1036-1042) on examples/genomic-preprocessed.maf (the golden's blocks and
trees), n = 1000: share of (block, sample) units that were simulated, null-sampling time with and without."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from helpers import block_from_golden  # noqa: E402
from rnacode_amd import api  # noqa: E402

from rnacode_amd.synth import synth_blocks  # noqa: E402

doc = load_golden("genomic_preprocessed_n1000")
ctx = api.Context(0)
n = 1000
for name, blocks in (("genomic-preprocessed.maf", [block_from_golden(e) for e in doc["blocks"]]),
                     ("%d synthetic non-coding blocks 6 x 120" % int(os.environ.get("NB", "2000")), [b.upper() for b in synth_blocks(int(os.environ.get("NB", "2000")), 6, 120, seed=1)])):
  print(name)
  for cutoff in (0.05, 0.01):
      res = {}
      for stop in (0, 1):
          p = api.default_params(sampleN=n, seed_base=42, stopEarly=stop, cutoff=cutoff)
          b = api.Batch(ctx, blocks, p).run()
          b.run()
          t, _ = b.timing()
          mx = b.maxScores_all()
          ok = [i for i in range(b.n) if b.status(i) == api.RC_OK]
          done = float((mx[ok] != -1.0).sum() + ((mx[ok] == -1.0) & (np.arange(n)[None, :] < 0)).sum())
          fits = b.fits()
          res[stop] = (t["null"], done / (len(ok) * n), int((fits[ok, 0] == -1).sum()), len(ok))
          b.close()
      print(f"cutoff {cutoff}: {res[1][3]} scored blocks; stop-early decides {res[1][2]} of them; "
            f"null sampling {res[0][0]:.2f} ms -> {res[1][0]:.2f} ms; samples with a result {res[0][1]:.3f} -> {res[1][1]:.3f}")
ctx.close()
