#!/bin/bash
# Everything profiles/r04 holds that comes from rocprofv3, in one gpurun call: profiles/collect_r04.sh  (then copy the summaries: see profiles/r04/README.md)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
mkdir -p gpurun_out
profiles/collect.sh r04_c3 > gpurun_out/collect_r04_c3.log 2>&1
profiles/collect.sh r04_c5 --config c5 > gpurun_out/collect_r04_c5.log 2>&1
# the same kernel without this round's two changes to the launches that read their codes from L2 (no suffix cache, the r03 occupancy):
# what the counters say each of them removes
RC_SUFFIX_CACHE=0 RC_HIGH_OCCUPANCY=0 profiles/collect.sh r04_c5_nocache_noocc --config c5 > gpurun_out/collect_r04_c5_nocache_noocc.log 2>&1
RC_SUFFIX_CACHE=1 RC_HIGH_OCCUPANCY=0 profiles/collect.sh r04_c5_cache_noocc --config c5 > gpurun_out/collect_r04_c5_cache_noocc.log 2>&1
profiles/collect.sh r04_shard --blocks 1250 > gpurun_out/collect_r04_shard.log 2>&1
for t in r04_c3 r04_c5 r04_c5_nocache_noocc r04_c5_cache_noocc r04_shard; do
  rm -rf gpurun_out/prof_$t/pmc_*/*/ 2>/dev/null
  find gpurun_out/prof_$t -name "*counter_collection.csv" -size +2M -delete
done
ls gpurun_out/prof_r04_c3 gpurun_out/prof_r04_c5
