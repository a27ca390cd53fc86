#!/bin/bash
# tools/mk_ab.sh [REV]: build tools/ab_A.so from REV (default HEAD) and tools/ab_B.so from the working tree, both with
# -DRC_PROFILING (RC_DEBUG_SKIP works in those builds only), into their own object directories: the product library is not touched.
# A is built from a detached worktree of REV (the working tree, its stash list and its untracked files are left alone).
set -e
cd "$(dirname "$0")/.."
ROOT=$(pwd)
REV=${1:-HEAD}
WT=$(mktemp -d /tmp/rc_ab_A.XXXXXX)
trap 'git -C "$ROOT" worktree remove --force "$WT" 2>/dev/null || rm -rf "$WT"' EXIT
git worktree add -q --detach "$WT" "$REV"
make -s -C "$WT/rnacode_amd/csrc" -j6 OUT="$ROOT/tools/ab_A.so" OBJDIR="$ROOT/tools/build_ab_A" EXTRA=-DRC_PROFILING lib > /dev/null
make -s -C "$ROOT/rnacode_amd/csrc" -j6 OUT="$ROOT/tools/ab_B.so" OBJDIR="$ROOT/tools/build_ab_B" EXTRA=-DRC_PROFILING lib > /dev/null
echo "built: A = $REV, B = working tree"
