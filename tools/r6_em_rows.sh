#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
run() { python $R/bench.py --workload em --rows $2 --steps 40 --warmup 5 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 rows $2: %.4f ms' % (d['ms_per_step']))"; }
for rows in 4096 16384 32768; do
LASSO_EM_SIDE_STREAM=0 run "one-stream" $rows
run "two-stream" $rows
LASSO_EM_SIDE_STREAM=0 run "one-stream" $rows
run "two-stream" $rows
done
