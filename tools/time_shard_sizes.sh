#!/bin/bash
# resident step time of 6 x 120 batches of several sizes (what a sub-batch of a stream costs on an otherwise idle GPU)
cd ${GRAFT_REPO_ROOT:-.}
for nb in 40 78 156 313 625 1250 2500; do python tools/time_shard.py $nb 20; done
