#!/bin/bash
# usage: tools/prof_trace.sh <tag> <command ...>   (on the GPU box via gpurun): kernel trace + stats only
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
python $R/tools/summarize_prof.py $OUT | tee $OUT/summary.txt
