mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q -x) > gpurun_out/gputest_d.log 2>&1; tail -4 gpurun_out/gputest_d.log
for ts in 1 0; do for nb in 313 625 1250 10000; do RC_TAIL_SHARING=$ts python tools/time_shard.py $nb 40; done; done > gpurun_out/tail_sharing_ab2.txt 2>&1; cat gpurun_out/tail_sharing_ab2.txt
for ts in 1 0; do for plan in "" "160,320,770" "128,256,512,354" "192,384,674" "256,994"; do echo "RC_TAIL_SHARING=$ts plan=$plan"; RC_TAIL_SHARING=$ts python tools/trace_shard.py 1250 2 $plan 2>/dev/null | tail -1; done; done > gpurun_out/shard_plans.txt 2>&1; cat gpurun_out/shard_plans.txt
RC_TRACE=1 python tools/trace_shard.py 1250 2 > gpurun_out/shard_trace2_out.txt 2> gpurun_out/shard_trace2_err.txt
python bench.py --no-cpu-baseline > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err; python tools/bench_line.py < gpurun_out/bench_d.json; python -c "import json;d=json.load(open(\"gpurun_out/bench_d.json\"));print(json.dumps(d[\"shard\"])[:900]);print(json.dumps(d[\"end_to_end\"])[:300]);print(d[\"mixed\"][\"ms_per_step\"],d[\"mixed\"][\"resident\"][\"ms_per_step\"],d[\"c5\"][\"ms_per_step\"])"
