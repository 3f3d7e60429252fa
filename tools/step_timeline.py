#!/usr/bin/env python3
"""Kernel timeline (all streams) of the last part of a rocprofv3 --kernel-trace csv: start -> end, duration, queue, name.
usage: step_timeline.py <kernel_trace.csv> <count> [anchor-substring]   (anchor: start the window at the count-th last
dispatch whose name contains it)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
count = int(sys.argv[2])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
ev = ev[-count:]
t0 = ev[0][0]
qs = sorted({e[3] for e in ev})
for s, e, n, q in ev:
    name = n.replace("lasso::", "").replace("(anonymous namespace)::", "").split("(")[0][:52]
    print("%9.1f -> %9.1f  (%7.1f)  q%-2s %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, "    " * qs.index(q), name))
print("span %.1f us" % ((max(e[1] for e in ev) - t0) / 1e3))
