#!/bin/bash
# A/B two builds of the library on the same GPU box: tools/ab.sh [bench args]; expects tools/ab_A.so, tools/ab_B.so
cd "$(dirname "$0")/.."
for round in 1 2; do
  for v in A B; do
    cp tools/ab_$v.so rnacode_amd/librnacode_hip.so
    for s in 0 2; do
      RC_DEBUG_SKIP=$s python bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v skip', $s, round(d['roofline']['kernel_ms'],3))"
    done
  done
done
