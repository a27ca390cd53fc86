#!/bin/bash
# rocprofv3 counters for the wide-block kernels (k_generic_sim + k_generic_dp, or k_tiled_dp for 34..64 rows of short blocks):
#   [PROF_TAG=tag] tools/prof_wide.sh [time_wide args: rows cols blocks samples]  -> gpurun_out/prof_wide[_tag]/summary.txt
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_wide${PROF_TAG:+_$PROF_TAG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$REPO
ARGS="${*:-100 300 256 1000}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/tools/time_wide.py" $ARGS > "$OUT/run.txt" 2> "$OUT/stats.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/tools/time_wide.py" $ARGS > /dev/null 2> "$OUT/pmc_fetch.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/tools/time_wide.py" $ARGS > /dev/null 2> "$OUT/pmc_write.log"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/tools/time_wide.py" $ARGS > /dev/null 2> "$OUT/pmc_sq.log"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA \
  --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/tools/time_wide.py" $ARGS > /dev/null 2> "$OUT/pmc_sq2.log"
python3 "$REPO/profiles/summarize.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -i "generic\|tiled\|== " "$OUT/summary.txt"
find "$OUT" -name "*agent_info*" -delete
