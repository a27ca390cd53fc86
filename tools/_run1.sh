set -x
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tiled" 2>&1 | tail -3
RC_LIB_PATH=$(pwd)/tools/p4.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tiled" 2>&1 | tail -3
tools/ab_rows.sh tools/p0.so tools/p1.so tools/p3.so tools/p4.so -- --rows 34,40,48,64 --cols 150,200 --no-generic > gpurun_out/ab_rows_dual.txt 2>&1
RC_TILED_DUAL=0 tools/ab_rows.sh tools/p3.so -- --rows 34,40,48,64 --cols 150,200 --no-generic > gpurun_out/ab_rows_p3_single.txt 2>&1
cat gpurun_out/ab_rows_dual.txt gpurun_out/ab_rows_p3_single.txt
