#!/bin/bash
# the headline workload's k_null launch with two builds of the library on the same box: tools/ab_A.so (a previous round's) and tools/ab_B.so
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for v in A B; do
    for cfg in "RC_X=0" "RC_TAIL_SHARING=0" "RC_TAIL_RELEASE=workgroup"; do
      echo -n "$v $cfg: "; env $cfg RC_LIB_PATH=$(pwd)/tools/ab_$v.so python tools/time_shard.py 10000 15 | sed 's/.*device total //; s/, mt_stream 0.000//'
    done
  done
done
