#!/bin/bash
# usage (on the GPU box via gpurun): tools/round_profiles.sh <round tag, e.g. r02>
# Everything the round's numbers in DESIGN.md / profiles/ come from, in one call:
#   prof_<tag>_fista   headline kernel (bench.py N=1): kernel trace + PMC passes
#   prof_<tag>_splitk  the same at 512 rows (split-k kernel)
#   <tag>_*.json(l)    shape matrix, small batches, EM step, config 3, ridge, conv + kernel stats of the EM step,
#                      config 3 and the conv solver
T=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/prof_pmc.sh ${T}_fista > $O/${T}_fista_summary.txt 2>&1
bash $R/tools/prof_pmc.sh ${T}_splitk --rows 512 > $O/${T}_splitk_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_matrix.py 2>/dev/null | tail -1 > $O/${T}_fista_shapes.json
python $R/tools/bench_small.py 2>/dev/null | grep "^{" > $O/${T}_small_batches.jsonl
python $R/tools/bench_em.py 2>/dev/null | grep '^{' > $O/${T}_em.jsonl
python $R/tools/bench_c3.py 2>/dev/null | grep '^{' > $O/${T}_c3.jsonl
python $R/tools/bench_c5.py 2>/dev/null | grep '^{' > $O/${T}_c5.jsonl
python $R/tools/bench_ridge.py 2>/dev/null | grep '^{' > $O/${T}_ridge.jsonl
python $R/tools/bench_conv.py --no-cpu 2>/dev/null | tail -1 > $O/${T}_conv.json
for job in "em bench_em.py --n 8192" "c3 bench_c3.py" "conv bench_conv.py --no-cpu"; do
  set -- $job; tag=$1; shift
  rm -rf /tmp/pp_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$tag/trace -o t -- python $R/tools/$* > /dev/null 2>&1
  python $R/tools/summarize_prof.py /tmp/pp_$tag > $O/${T}_${tag}_kernel_stats.txt 2>&1
done
ls -la $O | tail -20
