#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
for i in 1 2 3 4 5 6 7 8; do
python $R/bench.py --workload em --rows 8192 --steps 60 --warmup 10 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print('two-stream %d: %.4f ms dev %.4f path %s' % ($i, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['em_path']))"
done
