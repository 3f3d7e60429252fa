#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
one() { python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1', round(d['value'],1), round(d['roofline']['frac'],4), d['roofline']['avg_launch_ms'])"; }
for i in 1 2 3; do
  python $R/bench.py --steps 30 --warmup 5 2>/dev/null | grep '^{' | one product
  python $R/tools/bench_variant.py $R/variants/liblasso_noskip.so --steps 30 --warmup 5 2>/dev/null | grep '^{' | one noskip
done
