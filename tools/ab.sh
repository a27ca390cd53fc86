#!/bin/bash
# A/B two profiling builds of the library on the same GPU box: tools/ab.sh [bench args]; expects tools/ab_A.so, tools/ab_B.so
# (tools/mk_ab.sh).  They are loaded through RC_LIB_PATH; RC_DEBUG_SKIP=2 leaves out the simulation phase of k_null.
cd "$(dirname "$0")/.."
for round in 1 2; do
  for v in A B; do
    for s in 0 2; do
      RC_LIB_PATH=$(pwd)/tools/ab_$v.so RC_DEBUG_SKIP=$s python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stream "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v skip', $s, round(d['roofline']['kernel_ms'],3))"
    done
  done
done
