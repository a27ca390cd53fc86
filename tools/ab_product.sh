#!/bin/bash
# A/B of two PRODUCT builds on the headline workload (the profiling builds of tools/mk_ab.sh carry cell counters that cost the two-row kernel its
# fourth wavefront per SIMD): tools/ab_A.so (built from a revision: see the commit message / profiles note) against rnacode_amd/librnacode_hip.so.
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for v in A B; do
    lib=$(pwd)/rnacode_amd/librnacode_hip.so; [ $v = A ] && lib=$(pwd)/tools/ab_A.so
    RC_LIB_PATH=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stream --no-extra "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms_per_step', round(d['ms_per_step'],3), d['roofline']['kernel'])"
  done
done
