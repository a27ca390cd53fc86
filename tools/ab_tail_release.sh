#!/bin/bash
# k_null's tail sharing: the workgroup-scope publish (default) against the documented agent-scope release, and sharing off.
for rep in 1 2; do
for cfg in "RC_TAIL_RELEASE=workgroup" "RC_TAIL_RELEASE=agent" "RC_TAIL_SHARING=0"; do
  echo "== $cfg (run $rep)"
  for nb in 313 625 1250 10000; do env $cfg python tools/time_shard.py $nb 20 | sed 's/, mt_stream.*//'; done
  env $cfg python tools/trace_shard.py 1250 2 2>/dev/null | tail -2
done
done
