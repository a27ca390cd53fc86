mkdir -p gpurun_out
(python -m pytest tests -m gpu -q -x) > gpurun_out/gputest_m.log 2>&1; tail -3 gpurun_out/gputest_m.log
for cfg in "" "RC_FAT_STREAM_MIN_BLOCKS=0" "RC_FAT_STREAM_MIN_BLOCKS=100000" "RC_DUAL_ROWS=0"; do
  echo "== $cfg"
  env $cfg python bench.py --no-cpu-baseline > gpurun_out/bench_serial.json 2> /dev/null
  python tools/bench_line.py < gpurun_out/bench_serial.json
  python -c "
import json
d=json.load(open('gpurun_out/bench_serial.json'))
print('cont', d['stream']['continuous']['ms_per_step'], 'shard', d['shard']['ms_per_step'], d['shard']['stream']['ms_per_step'], 'c5', d['c5']['ms_per_step'], 'mixed', d['mixed']['ms_per_step'], d['mixed']['resident']['ms_per_step'], 'e2e', d['end_to_end']['wall_s_median'], d['roofline']['kernel'])"
done
