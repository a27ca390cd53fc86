#!/bin/bash
# A/B of the launches that read their sigma codes from L2: the LDS suffix cache and the high-occupancy build, each off and on.
# usage (GPU box): bash tools/ab_l2path.sh [short] > gpurun_out/ab_l2path.txt      short: only the two configurations with the cache on
CFGS=("RC_SUFFIX_CACHE=0 RC_HIGH_OCCUPANCY=0" "RC_SUFFIX_CACHE=1 RC_HIGH_OCCUPANCY=0" "RC_SUFFIX_CACHE=0 RC_HIGH_OCCUPANCY=1" "RC_SUFFIX_CACHE=1 RC_HIGH_OCCUPANCY=1")
if [ "$1" = "short" ]; then CFGS=("RC_SUFFIX_CACHE=1 RC_HIGH_OCCUPANCY=0" "RC_SUFFIX_CACHE=1 RC_HIGH_OCCUPANCY=1"); fi
for cfg in "${CFGS[@]}"; do
  echo "== $cfg"
  env $cfg python tools/time_uniform.py 4000 150 1000
  env $cfg python tools/time_uniform.py 1000 300 10000 12
  env $cfg python tools/time_mixed.py 10000 0 2>&1 | tail -5
done
