#!/bin/bash
# run bench.py for each variant lib given, interleaved, 2 rounds
for r in 1 2; do
for v in "$@"; do
  LASSO_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib$v.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-time-to-tol 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'it/s=%.0f'%d['value'], 'TF=%.1f'%d['roofline']['achieved'], 'frac=%.3f'%d['roofline']['frac'], 'obj', d.get('objective_after_100'))"
done; done
