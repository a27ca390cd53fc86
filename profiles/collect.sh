#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun).
#   profiles/collect.sh <tag> [bench args]     -> gpurun_out/prof_<tag>/{stats,pmc_*}     (e.g. collect.sh r02_c5 --config c5)
# Kernel trace/stats and each PMC group are separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
TAG=${1:-r02}
shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-extra $*"
# one kernel at a time while profiling: by default the native-block kernels run on a second stream beside k_null and
# their trace intervals then span its whole duration
export RC_SERIAL_NATIVE=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_under_trace.json" 2> "$OUT/stats.log"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_fetch.log"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_write.log"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_sq.log"
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM \
  --output-format csv -d "$OUT/pmc_sq2" -- python3 "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_sq2.log"
find "$OUT" -name "*.csv" | head -40
# keep the merged output small: the per-dispatch PMC csvs are summarised by profiles/summarize.py
python3 "$REPO/profiles/summarize.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
find "$OUT" -name "*agent_info*" -delete
