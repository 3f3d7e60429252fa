#!/bin/bash
# usage (GPU box): tools/r6_em_ab.sh <tag>   EM step at the 8-GPU shard size, configs 4 and 5: one-stream loop / side objective / + pipelined M-step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/emab_$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for s in c4 c5; do
  for mode in "0 0" "1 0" "1 1"; do
    set -- $mode
    for rep in 1 2; do
      LASSO_EM_SIDE_STREAM=$1 LASSO_EM_PIPELINE=$2 python $R/bench.py --workload em --shape $s --rows 8192 --steps 60 --warmup 10 2>$O/err.txt | grep "^{" > $O/${s}_$1$2_$rep.json
      python -c "import json;d=json.load(open('$O/${s}_$1$2_$rep.json'));print('$s side=$1 pipe=$2 rep $rep: ms_per_step %.4f  loss %.6f  path %s' % (d['ms_per_step'], d['objective_last_step'], d['em_path']))" || tail -5 $O/err.txt
    done
  done
done
