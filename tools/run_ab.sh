# A/B on one box: the round's new paths against their switches (RC_TAIL_SHARING=0 RC_COPY_RESULTS=1 RC_COMPUTE_STREAMS=2 = the old behaviour)
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q -x) > gpurun_out/gputest_e.log 2>&1; tail -3 gpurun_out/gputest_e.log
for rep in 1 2; do
  python bench.py --no-cpu-baseline > gpurun_out/ab_new_$rep.json 2> /dev/null
  RC_COMPUTE_STREAMS=2 python bench.py --no-cpu-baseline > gpurun_out/ab_two_$rep.json 2> /dev/null
  RC_TAIL_SHARING=0 RC_COPY_RESULTS=1 RC_COMPUTE_STREAMS=2 python bench.py --no-cpu-baseline > gpurun_out/ab_old_$rep.json 2> /dev/null
done
python - <<'PY'
import json
for name in ("new_1", "two_1", "old_1", "new_2", "two_2", "old_2"):
    d = json.load(open(f"gpurun_out/ab_{name}.json"))
    e = d["end_to_end"]
    print(f"{name:8s} step {d['ms_per_step']:.2f} stream {d['stream']['ms_per_step']:.2f} cont {d['stream']['continuous']['ms_per_step']:.2f} | shard {d['shard']['ms_per_step']:.3f} / {d['shard']['stream']['ms_per_step']:.3f} | "
          f"c5 {d['c5']['ms_per_step']:.1f} | mixed {d['mixed']['resident']['ms_per_step']:.1f} / {d['mixed']['ms_per_step']:.1f} | e2e {e['wall_s_median']:.3f} ({e['stages_of_median_run']})")
PY
