#!/usr/bin/env python3
"""Timeline of one streamed pass from the stderr of `RC_TRACE=1 python tools/trace_shard.py ... 2> err.txt`:
host stamps (submit / prepared / uploaded / queued / wait / done per sub-batch) and the device events of each sub-batch, all on the
host's monotonic clock, printed relative to the start of the pass.   tools/shard_trace_report.py err.txt [pass index, default: last]"""
import re
import sys

path = sys.argv[1]
want = int(sys.argv[2]) if len(sys.argv) > 2 else None
passes = []
cur = None
for ln in open(path):
    m = re.match(r"---- pass (\d+) starts at ([\d.]+) us", ln)
    if m:
        cur = {"idx": int(m.group(1)), "t0": float(m.group(2)), "t1": None, "rows": []}
        passes.append(cur)
        continue
    m = re.match(r"---- pass (\d+) ends at ([\d.]+) us", ln)
    if m and cur:
        cur["t1"] = float(m.group(2))
        continue
    if cur is None:
        continue
    m = re.match(r"\[rc\s+([\d.]+) us\] (\S+)\s+(0x[0-9a-f]+)", ln)
    if m:
        cur["rows"].append((float(m.group(1)), "host", m.group(2), m.group(3)))
        continue
    m = re.match(r"\[rc device (0x[0-9a-f]+) n=(\d+)\] now ([\d.]+) \| (.*) \(us", ln)
    if m:
        obj, n = m.group(1), int(m.group(2))
        parts = m.group(4).split()
        for k, v in zip(parts[0::2], parts[1::2]):
            cur["rows"].append((float(v), "dev", f"{k} (n={n})", obj))
p = passes[want] if want is not None else passes[-1]
print(f"pass {p['idx']}: {(p['t1'] - p['t0']) / 1e3:.3f} ms")
names = {}
for t, kind, what, obj in sorted(p["rows"]):
    if t < p["t0"] - 1e6:
        continue
    tag = names.setdefault(obj, f"B{len(names)}")
    print(f"  {(t - p['t0']) / 1e3:8.3f} ms  {kind:4s} {tag:3s} {what}")
