#!/bin/bash
# Where a fresh process spends its time before the GPU is busy: the native driver with RC_TRACE=1 on the 10 000-block inputs of
# bench.py's end_to_end legs (6 x 120, and 3..12 rows x 60..200 columns).  tools/cli_cold_trace.sh -> stdout
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import sys
sys.path.insert(0, ".")
import numpy as np
from rnacode_amd.synth import synth_blocks, synth_block, to_maf
open("/tmp/c3.maf", "w").write(to_maf(synth_blocks(10000, 6, 120, seed=1)))
rng = np.random.RandomState(5)
open("/tmp/mixed.maf", "w").write(to_maf([synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True) for i in range(10000)]))
PY
for f in c3 mixed; do
  echo "=== $f"
  for i in 1 2; do RC_CLI_TIMES=1 rnacode_amd/rnacode_hip /tmp/$f.maf -n 1000 -t -o /tmp/o.tsv 2>&1 | tail -1; sleep 0.3; done
  RC_TRACE=1 RC_CLI_TIMES=1 rnacode_amd/rnacode_hip /tmp/$f.maf -n 1000 -t -o /tmp/o.tsv 2>&1 | grep -v "^\[rc device" | head -150
done
