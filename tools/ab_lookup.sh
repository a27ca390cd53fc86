#!/bin/bash
# What the sigma look-ups and their address shifts cost in k_null's cell loop: four builds of the library (results are wrong in
# three of them), timed on the bench workload.
#   tools/ab_lookup.sh build    (here, hipcc)  -> tools/exp_{base,noshift,onelookup,both}.so
#   tools/ab_lookup.sh          (GPU box)      -> k_null ms per build
# base as shipped; noshift: 4 single-rate VALU instructions less per cell, LDS unchanged; onelookup: 4 of 5 ds_bpermute less, the shifts
# stay; both.  Two rows of a pass sharing their look-ups would save half of "both" per cell, minus the buffer moves.
cd "$(dirname "$0")/.."
ROOT=$(pwd)
if [ "$1" = build ]; then
  for v in base noshift onelookup both; do
    case $v in base) X="";; noshift) X="-DRC_EXP_NO_SHIFTS";; onelookup) X="-DRC_EXP_ONE_LOOKUP";; both) X="-DRC_EXP_NO_SHIFTS -DRC_EXP_ONE_LOOKUP";; esac
    make -s -C rnacode_amd/csrc -j6 OUT="$ROOT/tools/exp_$v.so" OBJDIR="$ROOT/tools/build_exp_$v" EXTRA="$X" lib > /dev/null && echo "built tools/exp_$v.so"
  done
  exit 0
fi
for round in 1 2; do
  for v in base noshift onelookup both; do
    RC_LIB_PATH=$ROOT/tools/exp_$v.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stream --no-extra "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'k_null ms', round(d['roofline']['kernel_ms'],3))"
  done
done
