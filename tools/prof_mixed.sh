#!/bin/bash
# rocprofv3 kernel stats, a timeline of the last resident run and two SQ counter passes of the mixed workload: tools/prof_mixed.sh <tag> -> gpurun_out/prof_mixed_<tag>/
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_mixed_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/tools/mixed_resident.py" 4 > "$OUT/run.log" 2> "$OUT/stats.log"
TRACE=$(find "$OUT/stats" -name "*kernel_trace.csv" | head -1)
[ -n "$TRACE" ] && python3 "$REPO/tools/kernel_timeline.py" "$TRACE" 100 > "$OUT/timeline_last_run.txt" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/tools/mixed_resident.py" 2 > /dev/null 2> "$OUT/pmc_sq.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/tools/mixed_resident.py" 2 > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/tools/mixed_resident.py" 2 > /dev/null 2> "$OUT/pmc_write.log"
cd "$REPO" && python3 profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/run.log"; grep -E "calls=" "$OUT/summary.txt" | head -40
find "$OUT" -name "*.csv" ! -name "*kernel_stats.csv" -delete; find "$OUT" -name "*agent_info*" -delete
