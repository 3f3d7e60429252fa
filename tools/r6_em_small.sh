#!/bin/bash
# usage (GPU box): tools/r6_em_small.sh <tag>   the double-buffered EM loop of small dictionaries: its tests,
# then config 5's shape at several row counts, one-stream loop (LASSO_EM_SIDE_STREAM=0) against the default, twice each;
# a step timeline of the shard size; a few other shapes forced onto the form
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/emsmall_$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest tests/test_dict_learning_gpu.py tests/test_parallel_gpu.py -x -q -m gpu \
    -k "out_of_place or double_buffered or two_stream or two_ranks" 2>&1 | tail -15 ) > $O/tests.txt
cat $O/tests.txt
for rows in 8192 65536 2048 32768; do
  for side in 0 1; do
    for rep in 1 2; do
      LASSO_EM_SIDE_STREAM=$side python $R/bench.py --workload em --shape c5 --rows $rows --steps 60 --warmup 10 2>$O/err.txt | grep "^{" > $O/c5_${rows}_${side}_$rep.json
      python -c "import json;d=json.load(open('$O/c5_${rows}_${side}_$rep.json'));print('c5 rows=$rows side=$side rep $rep: ms_per_step %.4f  loss %.6f  path %s' % (d['ms_per_step'], d['objective_last_step'], d['em_path']))" || tail -5 $O/err.txt
    done
  done
done
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --workload em --shape c5 --rows 8192 --steps 12 --warmup 0 > $O/trace.log 2>&1
python $R/tools/step_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 64 > $O/timeline.txt 2>&1
tail -64 $O/timeline.txt
