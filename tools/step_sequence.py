#!/usr/bin/env python3
"""The launch sequence of ONE step (solve / EM step) from a rocprofv3 --kernel-trace csv: the last `count` dispatches
in time order with duration and the idle gap in front of each.  usage: step_sequence.py <kernel_trace.csv> <count>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
count = int(sys.argv[2])
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))[-count:]
t0 = ev[0][0]
prev_end = ev[0][0]
for s, e, n in ev:
    name = n.replace("lasso::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("%9.1f us  gap %6.1f  run %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
    prev_end = max(prev_end, e)
print("span %.1f us" % ((ev[-1][1] - t0) / 1e3))
