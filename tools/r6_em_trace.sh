#!/bin/bash
# usage (GPU box): tools/r6_em_trace.sh <tag> <shape> <side> <pipe> [count]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/emtr_$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pe_t
LASSO_EM_SIDE_STREAM=$3 LASSO_EM_PIPELINE=$4 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pe_t/trace -o t -- python $R/bench.py --workload em --shape $2 --rows 8192 --steps 30 --warmup 5 > $O/bench_$2_$3$4.log 2>&1
f=$(find /tmp/pe_t -name '*kernel_trace.csv' | head -1)
python $R/tools/step_timeline.py $f ${5:-60} > $O/timeline_$2_$3$4.txt 2>&1
