for cfg in "" "RC_GRID_CAP_PER_CU=15" "RC_GRID_CAP_PER_CU=14" "RC_NATIVE_FIRST=0" "RC_NATIVE_FIRST=0 RC_GRID_CAP_PER_CU=15" "RC_NATIVE_FIRST=0 RC_TAIL_SHARING=0"; do
  echo "== $cfg"
  env $cfg python bench.py --no-cpu-baseline --no-stream --no-extra --steps 10 2>/dev/null | python tools/bench_line.py
done
