#!/bin/bash
# tools/mk_ab.sh: build tools/ab_A.so from HEAD and tools/ab_B.so from the working tree, both with -DRC_PROFILING
# (RC_DEBUG_SKIP works in those builds only), into their own object directories: the product library is not touched.
set -e
cd "$(dirname "$0")/.."
ROOT=$(pwd)
build() { make -s -C rnacode_amd/csrc -j6 OUT=$ROOT/tools/ab_$1.so OBJDIR=$ROOT/tools/build_ab_$1 EXTRA=-DRC_PROFILING lib > /dev/null; }
git stash -q
build A
git stash pop -q
build B
echo built
