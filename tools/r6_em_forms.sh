#!/bin/bash
# usage (GPU box): tools/r6_em_forms.sh <tag>   config 4's EM step by rows per GPU: one-stream loop / pipelined M-step /
# double-buffered dictionary with the objective held back until the sweep starts (LASSO_EM_FORM)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/emforms_$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for rows in 65536 32768 16384 8192; do
  steps=$((20 * 65536 / rows)); [ $steps -gt 60 ] && steps=60
  for form in one pipeline double-buffer; do
    for rep in 1 2; do
      if [ $form = one ]; then export LASSO_EM_SIDE_STREAM=0; unset LASSO_EM_FORM; else unset LASSO_EM_SIDE_STREAM; export LASSO_EM_FORM=$form; fi
      python $R/bench.py --workload em --shape c4 --rows $rows --steps $steps --warmup 5 2>$O/err.txt | grep "^{" > $O/c4_${rows}_${form}_$rep.json
      python -c "import json;d=json.load(open('$O/c4_${rows}_${form}_$rep.json'));print('c4 rows=$rows $form rep $rep: ms_per_step %.4f  loss %.6f  path %s' % (d['ms_per_step'], d['objective_last_step'], d['em_path']))" || tail -5 $O/err.txt
    done
  done
done
