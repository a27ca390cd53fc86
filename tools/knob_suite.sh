mkdir -p gpurun_out
# the parity and stream files under every remaining switch's other settings (tests/test_gpu_soak.py runs a subset of this in pytest)
for cfg in "RC_TAIL_SHARING=0" "RC_FAT_STREAM_MIN_ITEMS=0" "RC_HOST_THREADS=1 RC_SPLIT_FACTOR=0" "RC_SPLIT_FACTOR=100" "RC_HIGH_OCCUPANCY=0" "RC_HIGH_OCCUPANCY=2" "RC_TOGETHER_MAX_MB=0" "RC_LDS_MAX_BYTES=0"; do
  echo "== $cfg: $(env $cfg python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -m gpu -q 2>&1 | tail -1)"
done
