#!/bin/bash
# A/B of two PRODUCT builds on the headline workload: tools/ab_two.sh <libA.so> <libB.so> [bench args]   (paths relative to the repository)
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
for round in 1 2 3; do
  for v in A B; do
    lib=$(pwd)/$A; [ $v = B ] && lib=$(pwd)/$B
    RC_LIB_PATH=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stream --no-extra "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'ms_per_step', round(d['ms_per_step'],3), d['roofline']['kernel'])"
  done
done
