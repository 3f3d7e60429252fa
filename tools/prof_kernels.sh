#!/bin/bash
# usage: tools/prof_kernels.sh <tag> <command ...>   (on the GPU box via gpurun)
# rocprofv3 --kernel-trace --stats of an arbitrary command -> gpurun_out/<tag>/kernel_stats.txt (one line per kernel)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
python - <<PY
import csv, glob, os
rows = []
for f in glob.glob(os.path.join("$OUT", "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["Percentage"]))
with open(os.path.join("$OUT", "kernel_stats.txt"), "w") as o:
    o.write("== kernel stats (rocprofv3 --kernel-trace --stats): $* ==\n")
    for r in rows:
        o.write("%-90s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s\n" % (r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
print(open(os.path.join("$OUT", "kernel_stats.txt")).read())
PY
rm -rf $OUT/trace
