#!/bin/bash
# A/B of the launches that read their sigma codes from L2: the LDS suffix cache and the high-occupancy build, each off and on.
# usage (GPU box): bash tools/ab_l2path.sh > gpurun_out/ab_l2path.txt
for cfg in "RC_SUFFIX_CACHE=0 RC_HIGH_OCCUPANCY=0" "RC_SUFFIX_CACHE=1 RC_HIGH_OCCUPANCY=0" "RC_SUFFIX_CACHE=0 RC_HIGH_OCCUPANCY=1" "RC_SUFFIX_CACHE=1 RC_HIGH_OCCUPANCY=1"; do
  echo "== $cfg"
  env $cfg python tools/time_uniform.py 4000 150 1000
  env $cfg python tools/time_uniform.py 1000 300 10000 12
  env $cfg python tools/time_mixed.py 10000 0 2>&1 | tail -5
done
