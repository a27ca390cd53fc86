#!/usr/bin/env python3
"""The native driver end to end on a 100 000-block MAF (the bench's end_to_end_100k leg alone): tools/e2e_100k.py [repeat] [extra driver args...]"""
import json
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from rnacode_amd.synth import synth_blocks  # noqa: E402

repeat = int(sys.argv[1]) if len(sys.argv) > 1 else 10
blocks = [b.upper() for b in synth_blocks(10000, 6, 120, seed=1)]
r = bench.end_to_end_leg(blocks, 1000, 42, runs=3, repeat=repeat, extra_args=sys.argv[2:])
print(json.dumps({k: r[k] for k in ("value", "wall_s_all", "stages_of_median_run", "blocks") if k in r}, indent=1))
