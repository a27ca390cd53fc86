#!/bin/bash
# Everything profiles/r05 holds that comes from rocprofv3 or from a timing tool, in one gpurun call: profiles/collect_r05.sh
# (then copy the summaries: profiles/r05/README.md says which file is what).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
mkdir -p gpurun_out
profiles/collect.sh r05_c3 > gpurun_out/collect_r05_c3.log 2>&1
profiles/collect.sh r05_c5 --config c5 > gpurun_out/collect_r05_c5.log 2>&1
profiles/collect.sh r05_shard --blocks 1250 > gpurun_out/collect_r05_shard.log 2>&1
for t in r05_c3 r05_c5 r05_shard; do
  rm -rf gpurun_out/prof_$t/pmc_*/*/ 2>/dev/null
  find gpurun_out/prof_$t -name "*counter_collection.csv" -size +2M -delete
done
bash tools/prof_tree.sh r05 > gpurun_out/collect_r05_tree.log 2>&1
bash tools/prof_tree.sh r05m mixed > gpurun_out/collect_r05_tree_mixed.log 2>&1
bash tools/prof_mixed.sh r05 > gpurun_out/collect_r05_mixed.log 2>&1
python tools/mixed_by_class.py > gpurun_out/mixed_by_class.txt 2>&1
bash tools/tree_phases.sh > gpurun_out/tree_phases_c3.txt 2>&1
bash tools/tree_phases.sh mixed > gpurun_out/tree_phases_mixed.txt 2>&1
ls gpurun_out/prof_r05_c3 gpurun_out/prof_r05_c5 gpurun_out/prof_tree_r05 gpurun_out/prof_mixed_r05
