mkdir -p gpurun_out
for cfg in "RC_TAIL_SHARING=0" "RC_DUAL_ROWS=0" "RC_NATIVE_FIRST=0" "RC_NATIVE_FIRST=1" "RC_FAT_STREAM_MIN_ITEMS=0" "RC_COPY_RESULTS=1" "RC_COMPUTE_STREAMS=1" "RC_COMPUTE_STREAMS=3" "RC_HOST_THREADS=1 RC_SPLIT_FACTOR=0" "RC_SPLIT_FACTOR=100"; do
  echo "== $cfg: $(env $cfg python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -m gpu -q 2>&1 | tail -1)"
done
