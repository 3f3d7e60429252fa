#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace csv: for the last `--window-ms` of the run, per-kernel
totals and the idle time between consecutive kernels.  usage: trace_timeline.py <kernel_trace.csv> [window_ms]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else None
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
if win:
    t_end = ev[-1][1]
    ev = [e for e in ev if e[0] >= t_end - win]
busy = defaultdict(lambda: [0, 0])
gap_after = defaultdict(lambda: [0, 0])
idle = 0
for a, b in zip(ev, ev[1:]):
    g = max(0, b[0] - a[1])
    idle += g
    gap_after[a[2]][0] += g; gap_after[a[2]][1] += 1
for s, e, n in ev:
    busy[n][0] += e - s; busy[n][1] += 1
span = ev[-1][1] - ev[0][0]
print("window %.3f ms, %d kernels, busy %.3f ms, idle %.3f ms" % (span / 1e6, len(ev), sum(v[0] for v in busy.values()) / 1e6, idle / 1e6))
for n, (t, c) in sorted(busy.items(), key=lambda kv: -kv[1][0]):
    g, gc = gap_after[n]
    print("%8.1f us total %6d calls %8.2f us avg   gap-after avg %6.2f us   %s" % (t / 1e3, c, t / 1e3 / c, g / 1e3 / max(gc, 1), n[:110]))
