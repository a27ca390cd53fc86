#!/bin/bash
# Where a tree fit's cycles go: builds the library with -DRC_TREE_PROFILE into tools/ (the product is not touched) and runs
# tools/treefit_time.py with it; rc_fit_trees_device then prints the phases' shares and the counters.   tools/tree_phases.sh [mixed]
set -e
cd "$(dirname "$0")/.."
[ -f tools/tree_prof.so ] || make -s -C rnacode_amd/csrc -j6 OUT="$PWD/tools/tree_prof.so" OBJDIR="$PWD/tools/build_tree_prof" CLI="$PWD/tools/build_tree_prof/rnacode_hip" EXTRA=-DRC_TREE_PROFILE lib > /dev/null
RC_LIB_PATH="$PWD/tools/tree_prof.so" python3 tools/treefit_time.py "$@"
