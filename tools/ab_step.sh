#!/bin/bash
# like ab.sh but prints whole-step time and the per-kernel breakdown
cd "$(dirname "$0")/.."
for round in 1 2; do
  for v in A B; do
    RC_LIB_PATH=$(pwd)/tools/ab_$v.so python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), round(d.get('stream',{}).get('ms_per_step',0),3), {k: round(x,3) for k,x in d['device_ms_per_step'].items()})"
  done
done
