#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
run() { python $R/bench.py --workload em --shape c4 --rows 8192 --steps 60 --warmup 10 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1: %.4f ms' % d['ms_per_step'])"; }
run base; run base
LASSO_OBJ_GRID=128 run obj128
LASSO_OBJ_GRID=64 run obj64
LASSO_OBJ_GRID=32 run obj32
LASSO_PIPE_AVAIL=160 run avail160
LASSO_PIPE_AVAIL=128 run avail128
LASSO_PIPE_AVAIL=128 LASSO_OBJ_GRID=64 run avail128_obj64
LASSO_PIPE_AVAIL=96 LASSO_OBJ_GRID=64 run avail96_obj64
