#!/bin/bash
# small resident batches under several item-split thresholds (RC_SPLIT_FACTOR: split into strand x frame parts below this fraction of the wavefront slots)
cd ${GRAFT_REPO_ROOT:-.}
for f in 0.5 0.25 0.125 0.0; do
  for nb in 20 40 78 120; do echo -n "split factor $f: "; RC_SPLIT_FACTOR=$f python tools/time_shard.py $nb 20; done
done
echo "genomic-preprocessed.maf at n = 1000 (33 scored blocks):"
for f in 0.5 0.25 0.125; do echo -n "split factor $f: "; RC_SPLIT_FACTOR=$f python tools/run_c2.py 2>&1 | tail -1; done
