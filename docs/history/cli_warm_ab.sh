#!/bin/bash
# The context's warm-up thread (RC_WARM: 0 none, 1 the streams, 2 and the code objects of a first batch of up to 13 rows, 3 all code objects) on the
# native driver as a fresh process: wall time from exec to exit, three inputs, four runs per setting, interleaved.  tools/cli_warm_ab.sh -> stdout
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import sys
sys.path.insert(0, ".")
import numpy as np
from rnacode_amd.synth import synth_blocks, synth_block, to_maf
m = to_maf(synth_blocks(10000, 6, 120, seed=1))
open("/tmp/c3.maf", "w").write(m)
open("/tmp/c3_100k.maf", "w").write(m * 10)
rng = np.random.RandomState(5)
open("/tmp/mixed.maf", "w").write(to_maf([synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True) for i in range(10000)]))
PY
rnacode_amd/rnacode_hip /tmp/c3.maf -n 1000 -t -o /tmp/o.tsv > /dev/null 2>&1   # page the binary in
for f in c3 mixed c3_100k; do
  for round in 1 2 3 4; do
    for w in ${WARM_LEVELS:-0 1 2 3}; do
      python3 - "$f" "$w" <<'PY'
import os, subprocess, sys, time
f, w = sys.argv[1], sys.argv[2]
t = time.time()
r = subprocess.run(["rnacode_amd/rnacode_hip", "/tmp/%s.maf" % f, "-n", "1000", "-t", "-o", "/tmp/o_%s.tsv" % w], env=dict(os.environ, RC_WARM=w, RC_CLI_TIMES="1"), capture_output=True, text=True)
wall = time.time() - t
last = [l for l in r.stderr.splitlines() if l.startswith("[rnacode_hip] read")]
print("%-8s RC_WARM=%s wall %.3f s  %s" % (f, w, wall, last[-1][14:] if last else r.stderr[-200:]))
PY
    done
  done
  cmp /tmp/o_0.tsv /tmp/o_2.tsv && echo "$f: listings identical with and without the thread"
done
