#!/bin/bash
# rocprofv3 kernel stats (+ two PMC passes) of k_tree_fit: tools/prof_tree.sh <tag> [mixed]   -> gpurun_out/prof_tree_<tag>/
TAG=${1:-r03}
shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_tree_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/tools/treefit_time.py" $* > "$OUT/run.log" 2> "$OUT/stats.log"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/tools/treefit_time.py" $* > /dev/null 2> "$OUT/pmc_sq.log"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA \
  --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/tools/treefit_time.py" $* > /dev/null 2> "$OUT/pmc_sq2.log"
cd "$REPO" && python3 profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "k_tree_fit|==" "$OUT/summary.txt"
find "$OUT" -name "*.csv" ! -name "*kernel_stats.csv" -delete; find "$OUT" -name "*agent_info*" -delete
