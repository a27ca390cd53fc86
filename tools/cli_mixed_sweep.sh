#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import sys, os
sys.path.insert(0, ".")
import numpy as np
from rnacode_amd.synth import synth_block, to_maf
rng = np.random.RandomState(5)
blocks = [synth_block(rng, int(rng.randint(3, 13)), int(rng.randint(60, 201)), index=i, gaps=True) for i in range(10000)]
open("/tmp/mixed.maf", "w").write(to_maf(blocks))
PY
for mb in 0 2048 8192 default; do
  for rep in 1 2 3; do
    if [ $mb = default ]; then env RC_CLI_TIMES=1 rnacode_amd/rnacode_hip /tmp/mixed.maf -n 1000 -t -o /tmp/o.tsv 2>&1 | tail -1 | sed "s/^/together default: /"
    else env RC_TOGETHER_MAX_MB=$mb RC_CLI_TIMES=1 rnacode_amd/rnacode_hip /tmp/mixed.maf -n 1000 -t -o /tmp/o.tsv 2>&1 | tail -1 | sed "s/^/together $mb MB: /"; fi
  done
done
