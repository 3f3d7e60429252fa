#!/bin/bash
# usage (GPU box, via gpurun): bash tools/round4_profiles.sh
# Round 4: rocprofv3 kernel stats + the MFMA-busy table of every bench.py workload the driver (or the judge) may run,
# plus the PMC passes of the headline kernel (tools/prof_pmc.sh) -> gpurun_out/r04_*; copy what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/prof_pmc.sh r04_fista > $O/r04_fista_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for job in "c3_bf16 --workload c3 --dtype bf16 --steps 20" "c3_f32 --workload c3 --dtype f32 --steps 20" \
           "em_c4 --workload em --steps 20" "em_c4_shard --workload em --rows 8192 --steps 40" \
           "em_c5 --workload em --shape c5 --steps 40" "em_c5_shard --workload em --shape c5 --rows 8192 --steps 40"; do
  set -- $job; tag=$1; shift
  mkdir -p $O/r04_$tag
  python $R/bench.py "$@" > $O/r04_$tag/bench.json 2> $O/r04_$tag/bench.err
  rm -rf /tmp/pp_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$tag/trace -o t -- python $R/bench.py "$@" > /dev/null 2>&1
  python $R/tools/summarize_prof.py /tmp/pp_$tag > $O/r04_$tag/kernel_stats.txt 2>&1
  bash $R/tools/pmc_mfma_busy.sh $O/r04_$tag/mfma_busy.txt $R/bench.py "$@" > /dev/null 2>&1
done
ls -la $O | tail -12
