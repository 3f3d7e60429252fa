#!/bin/bash
# usage (GPU box): tools/prof_em_shards.sh <tag>   kernel stats of one EM step at the 8-GPU shard size, configs 4 and 5
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/em_$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for s in c4 c5; do
  python $R/bench.py --workload em --shape $s --rows 8192 --steps 60 --warmup 10 2>/dev/null | grep "^{" > $O/$s.json
  rm -rf /tmp/pe_$s
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$s/trace -o t -- python $R/bench.py --workload em --shape $s --rows 8192 --steps 40 --warmup 5 > /dev/null 2>&1
  python $R/tools/summarize_prof.py /tmp/pe_$s > $O/${s}_kernel_stats.txt 2>&1
  python -c "import json;d=json.load(open('$O/$s.json'));print('$s ms_per_step', d['ms_per_step'])"
  cut -c1-70,92-160 $O/${s}_kernel_stats.txt | head -28
done
