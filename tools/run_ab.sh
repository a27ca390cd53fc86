# A/B on one box: the round's new paths against their switches (RC_TAIL_SHARING=0 RC_COPY_RESULTS=1 = the old behaviour)
mkdir -p gpurun_out
for rep in 1 2; do
  python bench.py --no-cpu-baseline > gpurun_out/ab_new_$rep.json 2> /dev/null
  RC_TAIL_SHARING=0 RC_COPY_RESULTS=1 python bench.py --no-cpu-baseline > gpurun_out/ab_old_$rep.json 2> /dev/null
  RC_TAIL_SHARING=0 python bench.py --no-cpu-baseline > gpurun_out/ab_nosharing_$rep.json 2> /dev/null
done
python - <<'PY'
import json
for name in ("new_1", "old_1", "nosharing_1", "new_2", "old_2", "nosharing_2"):
    d = json.load(open(f"gpurun_out/ab_{name}.json"))
    e = d["end_to_end"]
    print(f"{name:12s} step {d['ms_per_step']:.2f} stream {d['stream']['ms_per_step']:.2f} cont {d['stream']['continuous']['ms_per_step']:.2f} | shard {d['shard']['ms_per_step']:.3f} / {d['shard']['stream']['ms_per_step']:.3f} | "
          f"c5 {d['c5']['ms_per_step']:.1f} | mixed {d['mixed']['resident']['ms_per_step']:.1f} / {d['mixed']['ms_per_step']:.1f} | e2e {e['wall_s_median']:.3f} ({e['stages_of_median_run']})")
PY
