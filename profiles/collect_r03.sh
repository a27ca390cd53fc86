#!/bin/bash
# Everything profiles/r03 holds, in one gpurun call: profiles/collect_r03.sh  (then copy the summaries: see profiles/r03/README.md)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
profiles/collect.sh r03_c3 > gpurun_out/collect_r03_c3.log 2>&1
profiles/collect.sh r03_c5 --config c5 > gpurun_out/collect_r03_c5.log 2>&1
profiles/collect.sh r03_shard --blocks 1250 > gpurun_out/collect_r03_shard.log 2>&1
tools/prof_tree.sh r03_c3 > gpurun_out/collect_r03_tree_c3.log 2>&1
tools/prof_tree.sh r03_mixed mixed > gpurun_out/collect_r03_tree_mixed.log 2>&1
tools/prof_wide.sh 100 300 256 1000 > gpurun_out/collect_r03_wide.log 2>&1
# the stream of one rank's shard: kernel timeline of its last pass
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_r03_shard_stream" -- python3 "$REPO/tools/trace_shard.py" 1250 2 > "$REPO/gpurun_out/prof_r03_shard_stream.txt" 2> /dev/null)
python3 tools/kernel_timeline.py "$(find gpurun_out/prof_r03_shard_stream -name '*kernel_trace.csv' | head -1)" 11 > gpurun_out/prof_r03_shard_stream_timeline.txt 2>&1
rm -rf gpurun_out/prof_r03_shard_stream
ls gpurun_out/prof_r03_c3 gpurun_out/prof_r03_c5 gpurun_out/prof_r03_shard
