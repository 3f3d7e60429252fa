#!/bin/bash
# usage (GPU box): tools/r6_pipe_trace.sh <tag> [n k]   kernel timeline of the pipelined M-step check
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pipe_$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp/trace -o t -- python $R/tools/check_pipe.py ${2:-8192} ${3:-1024} 6 > $O/check.log 2>&1
f=$(find /tmp/pp -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
# the last pipelined rep: find the last pipe sweep (gated) -> print 40 launches around
idx = [i for i, e in enumerate(ev) if "fold_rows_kernel" in e[2]]
last = idx[-1]
first = max(0, idx[-4] - 3)
t0 = ev[first][0]
for s, e, n, q in ev[first:last + 12]:
    name = n.replace("lasso::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    print("%9.1f -> %9.1f  (%7.1f)  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
PY
tail -8 $O/check.log; cat $O/timeline.txt
