#!/bin/bash
# usage (GPU box): tools/r6_em_probe.sh <tag> [count_c4 count_c5]
# bench line + kernel trace + launch sequence of the last EM steps at the 8-GPU shard size, configs 4 and 5
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/emp_$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for s in c4 c5; do
  python $R/bench.py --workload em --shape $s --rows 8192 --steps 60 --warmup 10 2>$O/$s.err | grep "^{" > $O/$s.json
  python $R/bench.py --workload em --shape $s --rows 8192 --steps 60 --warmup 10 2>>$O/$s.err | grep "^{" > $O/${s}_b.json
  rm -rf /tmp/pe_$s
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$s/trace -o t -- python $R/bench.py --workload em --shape $s --rows 8192 --steps 40 --warmup 5 > /dev/null 2>&1
  python $R/tools/summarize_prof.py /tmp/pe_$s > $O/${s}_kernel_stats.txt 2>&1
  f=$(find /tmp/pe_$s -name '*kernel_trace.csv' | head -1)
  python $R/tools/step_sequence.py $f ${2:-70} > $O/${s}_sequence.txt 2>&1
  python -c "import json;d=json.load(open('$O/$s.json'));e=json.load(open('$O/${s}_b.json'));print('$s ms_per_step', d['ms_per_step'], e['ms_per_step'], d['objective_last_step'])"
done
