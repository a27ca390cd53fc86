#!/usr/bin/env python3
"""Timeline of the last `window_ms` of a rocprofv3 --kernel-trace csv: tools/kernel_timeline.py trace.csv [window_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
end = max(r["e"] for r in rows)
sel = sorted((r for r in rows if r["s"] >= end - win * 1e6), key=lambda r: r["s"])
t0 = sel[0]["s"]
for r in sel:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rc::", "")
    if "fillBuffer" in name or "copyBuffer" in name:
        continue
    print(f"{(r['s'] - t0) / 1e6:9.3f} {(r['e'] - t0) / 1e6:9.3f} dur={(r['e'] - r['s']) / 1e6:7.3f} {name:28s} q={r.get('Queue_Id', '?')} grid={r.get('Grid_Size', r.get('Grid_Size_X', '?'))}")
