#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
run() { python $R/bench.py --workload em $2 --steps 30 --warmup 5 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1: %.4f ms  loss %.6f' % (d['ms_per_step'], d['objective_last_step']))"; }
LASSO_EM_SIDE_STREAM=0 run "c4 n=65536 one-stream" ""
run "c4 n=65536 two-stream" ""
LASSO_EM_SIDE_STREAM=0 run "c4 n=65536 one-stream" ""
run "c4 n=65536 two-stream" ""
run "c5 n=65536" "--shape c5"
