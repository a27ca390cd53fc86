#!/bin/bash
# The asm fetches of k_null (code words loaded into registers the compiler does not know to be pending) against the same kernels with plain
# loads: tools/ab_plain_fetch.sh build   (here: hipcc cross-compiles tools/plain_fetch.so, ~3 min)
#        tools/ab_plain_fetch.sh run     (on the GPU box: parity tests and soaks on the plain build, then both builds timed)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
if [ "$1" = build ]; then
  make -C rnacode_amd/csrc OBJDIR=/tmp/plain_fetch_build OUT=$PWD/tools/plain_fetch.so EXTRA=-DRC_PLAIN_FETCH lib -j8 2>&1 | grep -i "error"
  ls -la tools/plain_fetch.so
  exit 0
fi
export RC_LIB_PATH=$PWD/tools/plain_fetch.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q -k "not bench and not ranks and not collective" 2>&1 | tail -2
for m in "" fat; do timeout 300 python tools/soak_parity.py 100 77 $m 2>&1 | grep "MISMATCH\|^soak"; done
RC_LDS_MAX_BYTES=0 timeout 300 python tools/soak_parity.py 100 78 2>&1 | grep "MISMATCH\|^soak"
for lib in plain asm; do
  if [ $lib = asm ]; then unset RC_LIB_PATH; fi
  echo "== $lib fetches"
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-stream --no-extra 2>/dev/null | sed 's/.*"ms_per_step":\([0-9.]*\).*/  headline \1 ms per step/'
  python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-extra 2>/dev/null | sed 's/.*"ms_per_step":\([0-9.]*\).*/  c5 \1 ms per step/'
  python tools/time_uniform.py 4000 150 1000 7,10,13 2>&1 | tail -3
done
