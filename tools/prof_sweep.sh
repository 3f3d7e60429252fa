#!/bin/bash
# usage (GPU box): tools/prof_sweep.sh <tag> [--lib variants/liblasso_X.so]   kernel durations of the M-step (bench_sweep.py)
R=$GRAFT_REPO_ROOT; TAG=$1; shift; OUT=$R/gpurun_out/prof_sweep_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/tools/bench_sweep.py "$@" > $OUT/trace.log 2>&1
python $R/tools/summarize_prof.py $OUT | grep -i "sweep\|fixup\|transpose\|gemm_nt\|fillBuffer" | cut -c1-60,90-200 | tee $OUT/summary.txt
