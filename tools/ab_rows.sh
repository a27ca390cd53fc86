#!/bin/bash
# A/B of PRODUCT builds along the row-count axis: tools/ab_rows.sh <lib.so> [<lib.so> ...] -- <tools/rows_sweep.py arguments>
# (paths relative to the repository; two interleaved passes, one line per shape and build)
cd "$(dirname "$0")/.."
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
shift
for pass in 1 2; do
  for lib in "${libs[@]}"; do
    echo "== pass $pass: $lib"
    RC_LIB_PATH=$(pwd)/$lib python tools/rows_sweep.py "$@" 2>/dev/null | grep -E "^default" | cut -c1-125
  done
done
