#!/bin/bash
# gpurun_out/ (scratch) -> profiles/r06/ (tracked): the summaries of profiles/collect_r06.sh under the names profiles/r06/README.md lists
cd "$(dirname "$0")/.."
D=profiles/r06
mkdir -p $D
for t in c3 c5 shard; do
  cp gpurun_out/prof_r06_$t/summary.txt $D/rocprofv3_summary_$t.txt
  cp gpurun_out/prof_r06_$t/k_null_pmc.json $D/k_null_${t}_pmc.json
  cp gpurun_out/prof_r06_$t/bench_under_trace.json $D/bench_under_trace_$t.json
  cp $(find gpurun_out/prof_r06_$t/stats -name "*kernel_stats.csv" | head -1) $D/kernel_stats_$t.csv
done
for t in tiled40 generic48 generic100 rows28 tiled200; do
  cp gpurun_out/prof_wide_$t/summary.txt $D/rocprofv3_summary_$t.txt
  cp gpurun_out/prof_wide_$t/run.txt $D/run_$t.txt
  cp $(find gpurun_out/prof_wide_$t/stats -name "*kernel_stats.csv" | head -1) $D/kernel_stats_$t.csv
done
cp gpurun_out/prof_mixed_r06/summary.txt $D/rocprofv3_summary_mixed.txt
cp gpurun_out/prof_mixed_r06/timeline_last_run.txt $D/mixed_timeline_last_run.txt
cp gpurun_out/prof_mixed_r06/run.log $D/mixed_resident_runs.txt
cp $(find gpurun_out/prof_mixed_r06/stats -name "*kernel_stats.csv" | head -1) $D/kernel_stats_mixed.csv
{ echo "# tools/mixed_by_class.py, product build"; cat gpurun_out/mixed_by_class.txt;
  for s in 0 1 2; do [ -f gpurun_out/mixed_by_class_skip$s.txt ] && { echo; echo "# profiling build (tools/ab_B.so), RC_DEBUG_SKIP=$s (0: everything, 1: without the DP and scan = the simulation alone, 2: without the simulation)"; cat gpurun_out/mixed_by_class_skip$s.txt; }; done; } > $D/mixed_by_class.txt
cp gpurun_out/rows_sweep_after.txt gpurun_out/rows_sweep_after.json gpurun_out/time_wide.txt gpurun_out/bench_default_run.json gpurun_out/bench_default_line.txt $D/
ls $D
