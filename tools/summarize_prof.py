#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a short text summary."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for row in csv.DictReader(open(f)):
        print("%-90s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (row["Name"][:90], row["Calls"], row["AverageNs"], row["MinNs"], row["MaxNs"], row["Percentage"]))
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== PMC %s ==" % os.path.basename(d))
        for kname, ctrs in acc.items():
            if "lasso" not in kname:
                continue
            print(" kernel:", kname[:100])
            for c, vals in sorted(ctrs.items()):
                print("   %-32s per-dispatch mean=%.6g  (n=%d, min=%.6g max=%.6g)" % (c, sum(vals) / len(vals), len(vals), min(vals), max(vals)))
