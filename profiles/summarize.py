#!/usr/bin/env python3
"""Summarise rocprofv3 csv output (kernel stats + PMC passes) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("rc::", "").replace("void ", "")
    return name.split("(")[0][:60]


for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"{short(row['Name']):60s} calls={row['Calls']:>6s} total_ns={row['TotalDurationNs']:>14s} "
                  f"avg_ns={float(row['AverageNs']):14.1f} pct={row['Percentage']}")

for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
        print("== PMC per dispatch (mean over dispatches):", os.path.relpath(f, out))
        for k in agg:
            for c in agg[k]:
                print(f"{k:60s} {c:24s} mean={agg[k][c] / cnt[k][c]:18.1f} dispatches={cnt[k][c]}")

# machine-readable per-dispatch means for bench.py's roofline.traffic: the k_null instantiation that takes the time
# (a batch also launches a near-empty EXACT instantiation per class for blocks flagged on the device)
import json
dominant, best_total = None, -1.0
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "k_null" in row["Name"] and float(row["TotalDurationNs"]) > best_total:
                dominant, best_total = row["Name"], float(row["TotalDurationNs"])
res = {"kernel": dominant}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(float); cnt = defaultdict(int)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if dominant and row["Kernel_Name"].split("(")[0] == dominant.split("(")[0]:
                    agg[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
        for c in agg:
            res[c] = agg[c] / cnt[c]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row["Name"] == dominant:
                res["kernel_avg_ns"] = float(row["AverageNs"]); res["kernel_calls"] = int(row["Calls"])
with open(os.path.join(out, "k_null_pmc.json"), "w") as fh:
    json.dump(res, fh, indent=1, sort_keys=True)
