#!/bin/bash
# Quick SQ counter pass for k_null only (two PMC groups), used while tuning.
TAG=${1:-tune}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_sq.log"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
  --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_sq2.log"
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_VMEM_WR \
  --output-format csv -d "$OUT/pmc_sq3" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_sq3.log"
python3 "$REPO/profiles/summarize.py" "$OUT" | grep -E "k_null|==" 
find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*agent_info*" -delete
