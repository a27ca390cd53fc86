#!/bin/bash
# gpurun_out/ (scratch) -> profiles/r05/ (tracked): the summaries of profiles/collect_r05.sh under the names profiles/r05/README.md lists
cd "$(dirname "$0")/.."
D=profiles/r05
mkdir -p $D
for t in c3 c5 shard; do
  cp gpurun_out/prof_r05_$t/summary.txt $D/rocprofv3_summary_$t.txt
  cp gpurun_out/prof_r05_$t/k_null_pmc.json $D/k_null_${t}_pmc.json
  cp gpurun_out/prof_r05_$t/bench_under_trace.json $D/bench_under_trace_$t.json
  cp $(find gpurun_out/prof_r05_$t/stats -name "*kernel_stats.csv" | head -1) $D/kernel_stats_$t.csv
done
for t in r05:c3 r05m:mixed; do
  src=gpurun_out/prof_tree_${t%%:*}; name=${t##*:}
  cp $src/summary.txt $D/k_tree_fit_${name}_summary.txt
  cp $(find $src/stats -name "*kernel_stats.csv" | head -1) $D/k_tree_fit_${name}_kernel_stats.csv
done
cp gpurun_out/prof_mixed_r05/summary.txt $D/rocprofv3_summary_mixed.txt
cp gpurun_out/prof_mixed_r05/timeline_last_run.txt $D/mixed_timeline_last_run.txt
cp gpurun_out/prof_mixed_r05/run.log $D/mixed_resident_runs.txt
cp $(find gpurun_out/prof_mixed_r05/stats -name "*kernel_stats.csv" | head -1) $D/kernel_stats_mixed.csv
cp gpurun_out/mixed_by_class.txt $D/mixed_by_class.txt
grep -h "rc tree profile" gpurun_out/tree_phases_c3.txt | tail -1 > $D/k_tree_fit_phases.txt
grep -h "rc tree profile" gpurun_out/tree_phases_mixed.txt | tail -1 >> $D/k_tree_fit_phases.txt
ls $D
