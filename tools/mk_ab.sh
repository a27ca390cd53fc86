#!/bin/bash
# tools/mk_ab.sh: build tools/ab_A.so from HEAD and tools/ab_B.so from the working tree (for tools/ab.sh)
set -e
cd "$(dirname "$0")/.."
git stash -q
make -C rnacode_amd/csrc -j6 > /dev/null
cp rnacode_amd/librnacode_hip.so tools/ab_A.so
git stash pop -q
make -C rnacode_amd/csrc -j6 > /dev/null
cp rnacode_amd/librnacode_hip.so tools/ab_B.so
echo built
