#!/usr/bin/env python3
"""Compact view of bench.py's JSON line (stdin): tools/bench_line.py [label]"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d.get("stream", {})
print(sys.argv[1] if len(sys.argv) > 1 else "", "step %.3f ms" % d["ms_per_step"], "stream %.3f ms (%.3f)" % (s.get("ms_per_step", 0), s.get("frac_of_resident", 0)),
      "k_null %.3f ms" % d["roofline"]["kernel_ms"], "valu %.3f" % d["roofline"]["valu"]["frac"],
      {k: round(v, 3) for k, v in d["device_ms_per_step"].items()})
