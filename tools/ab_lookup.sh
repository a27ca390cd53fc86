#!/bin/bash
# What the sigma look-ups and their address shifts cost in k_null's cell loop (profiling build tools/ab_B.so, tools/mk_ab.sh; results are
# wrong with a skip set): RC_DEBUG_SKIP 0 as built, 4 no shifts (4 single-rate VALU less per cell, LDS unchanged), 8 one look-up
# per cell instead of five (4 ds_bpermute less, the shifts stay), 12 both -- two rows of a pass sharing their look-ups would sit between 0 and 12 / 2.
cd "$(dirname "$0")/.."
for round in 1 2; do
  for s in 0 4 8 12; do
    RC_LIB_PATH=$(pwd)/tools/ab_B.so RC_DEBUG_SKIP=$s python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stream --no-extra "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip', $s, 'k_null ms', round(d['roofline']['kernel_ms'],3))"
  done
done
