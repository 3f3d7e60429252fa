#!/bin/bash
# usage (GPU box): tools/r6_pipe_trace_lib.sh <tag> <lib.so>   kernel timeline of check_pipe.py on an A/B build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pipe_$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp/trace -o t -- python $R/tools/check_pipe.py 8192 1024 6 --lib $2 > $O/check.log 2>&1
f=$(find /tmp/pp -name '*kernel_trace.csv' | head -1)
python $R/tools/step_timeline.py $f 400 > $O/timeline.txt 2>&1
