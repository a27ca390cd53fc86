#!/bin/bash
# Everything profiles/r06 holds that comes from rocprofv3 or from a timing tool, in one gpurun call: profiles/collect_r06.sh
# (then copy the summaries: profiles/copy_r06.sh; profiles/r06/README.md says which file is what).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
mkdir -p gpurun_out
profiles/collect.sh r06_c3 > gpurun_out/collect_r06_c3.log 2>&1
profiles/collect.sh r06_c5 --config c5 > gpurun_out/collect_r06_c5.log 2>&1
profiles/collect.sh r06_shard --blocks 1250 > gpurun_out/collect_r06_shard.log 2>&1
for t in r06_c3 r06_c5 r06_shard; do
  rm -rf gpurun_out/prof_$t/pmc_*/*/ 2>/dev/null
  find gpurun_out/prof_$t -name "*counter_collection.csv" -size +2M -delete
done
# the kernels of the row-count axis: 40 x 150 (k_tiled_dp<13>), 48 x 300 (k_generic_dp at 48 rows), 100 x 300 (the wide benchmark of rounds 4 and 5), 28 x 150 (k_null<27> held to three wavefronts),
# 200 x 90 (k_tiled_dp past the per-row-count kernels' range)
PROF_TAG=tiled40 bash tools/prof_wide.sh 40 150 1662 1000 > gpurun_out/collect_r06_tiled40.log 2>&1
PROF_TAG=generic48 bash tools/prof_wide.sh 48 300 344 1000 > gpurun_out/collect_r06_generic48.log 2>&1
PROF_TAG=generic100 bash tools/prof_wide.sh 100 300 1024 1000 > gpurun_out/collect_r06_generic100.log 2>&1
PROF_TAG=rows28 bash tools/prof_wide.sh 28 150 2384 1000 > gpurun_out/collect_r06_rows28.log 2>&1
PROF_TAG=tiled200 bash tools/prof_wide.sh 200 90 500 1000 > gpurun_out/collect_r06_tiled200.log 2>&1
for t in tiled40 generic48 generic100 rows28 tiled200; do
  rm -rf gpurun_out/prof_wide_$t/pmc_*/*/ 2>/dev/null
  find gpurun_out/prof_wide_$t -name "*counter_collection.csv" -size +2M -delete
done
bash tools/prof_mixed.sh r06 > gpurun_out/collect_r06_mixed.log 2>&1
python tools/mixed_by_class.py > gpurun_out/mixed_by_class.txt 2>&1
# the simulation's share per class: a profiling build (tools/mk_ab.sh: tools/ab_B.so) with the DP (1) or the simulation (2) left out
if [ -f tools/ab_B.so ]; then
  for s in 0 1 2; do RC_LIB_PATH=$(pwd)/tools/ab_B.so RC_DEBUG_SKIP=$s python tools/mixed_by_class.py > gpurun_out/mixed_by_class_skip$s.txt 2>&1; done
fi
# the sweeps, the wide timings and the default bench run on the same code
python tools/rows_sweep.py --out gpurun_out/rows_sweep_after.json > gpurun_out/rows_sweep_after.txt 2>&1
{ python tools/time_wide.py 100 300 1024 1000; python tools/time_wide.py 200 90 500 1000; python tools/time_wide.py 100 300 16 1000; python tools/time_wide.py 500 150 100 1000; } > gpurun_out/time_wide.txt 2>&1
python bench.py --full-json gpurun_out/bench_default_run.json > gpurun_out/bench_default_line.txt 2> gpurun_out/bench_default_err.txt
ls gpurun_out/prof_r06_c3 gpurun_out/prof_r06_c5 gpurun_out/prof_wide_tiled40 gpurun_out/prof_mixed_r06
