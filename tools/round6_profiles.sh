#!/bin/bash
# usage (GPU box, via gpurun): bash tools/round6_profiles.sh [tags...]     (default: every workload)
# Round 6: for every bench.py workload the driver (or the judge) may run: the plain bench line, rocprofv3 kernel stats,
# the MFMA-busy table and the HBM traffic (tools/prof_hbm.sh) -> gpurun_out/r06_<tag>/; copy what is to be judged into
# profiles/.  The steps divisor of the traffic pass = the solves / EM steps the command executes (--warmup 0).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ALL="fista c3_bf16 c3_f32 em_c4 em_c4_shard em_c5 em_c5_shard conv_gray conv_rgb conv_c16 cd"
TAGS=${@:-$ALL}
cd /tmp && export TMPDIR=/tmp
for tag in $TAGS; do
  case $tag in
    fista)       ARGS="--workload fista --steps 30 --warmup 5 --no-cpu-baseline --no-time-to-tol --no-shards --no-extras"; PARGS="--workload fista --steps 20 --warmup 0 --no-cpu-baseline --no-time-to-tol --no-shards --no-extras"; DIV=20 ;;
    c3_bf16)     ARGS="--workload c3 --dtype bf16 --steps 20"; PARGS="--workload c3 --dtype bf16 --steps 20 --warmup 0"; DIV=21 ;;
    c3_f32)      ARGS="--workload c3 --dtype f32 --steps 20"; PARGS="--workload c3 --dtype f32 --steps 20 --warmup 0"; DIV=21 ;;
    em_c4)       ARGS="--workload em --steps 20"; PARGS="--workload em --steps 20 --warmup 0"; DIV=20 ;;
    em_c4_shard) ARGS="--workload em --rows 8192 --steps 40"; PARGS="--workload em --rows 8192 --steps 40 --warmup 0"; DIV=40 ;;
    # (0.2 ms steps: enough of them that the loop's start and end -- a last objective, the last sweep's count, the copy
    # into the caller's tensor: ~80 us once -- do not show in the per-step time; 40 steps read 0.200-0.207 ms, 200 steps 0.195-0.196)
    em_c5)       ARGS="--workload em --shape c5 --steps 100"; PARGS="--workload em --shape c5 --steps 100 --warmup 0"; DIV=100 ;;
    em_c5_shard) ARGS="--workload em --shape c5 --rows 8192 --steps 200"; PARGS="--workload em --shape c5 --rows 8192 --steps 200 --warmup 0"; DIV=200 ;;
    conv_gray)   ARGS="--workload conv --conv-case gray --steps 40"; PARGS="--workload conv --conv-case gray --steps 40 --warmup 0"; DIV=41 ;;
    conv_rgb)    ARGS="--workload conv --conv-case rgb --steps 20"; PARGS="--workload conv --conv-case rgb --steps 20 --warmup 0"; DIV=21 ;;
    conv_c16)    ARGS="--workload conv --conv-case c16 --steps 20"; PARGS="--workload conv --conv-case c16 --steps 20 --warmup 0"; DIV=21 ;;
    cd)          ARGS="--workload cd --steps 20 --no-cpu-baseline"; PARGS="--workload cd --steps 20 --warmup 0 --no-cpu-baseline"; DIV=21 ;;
    *) echo "unknown tag $tag"; continue ;;
  esac
  D=$O/r06_$tag
  mkdir -p $D
  rm -rf /tmp/pp_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$tag/trace -o t -- python $R/bench.py $ARGS > /dev/null 2>&1
  python $R/tools/summarize_prof.py /tmp/pp_$tag > $D/kernel_stats.txt 2>&1
  # EM workloads: the launch timeline of the last steps, both streams (the overlap of the two-stream loop)
  case $tag in em_*) f=$(find /tmp/pp_$tag -name '*kernel_trace.csv' | head -1); python $R/tools/step_timeline.py $f 70 > $D/timeline.txt 2>&1 ;; esac
  # PMC passes serialise the kernels of a process: the two-stream EM loop cannot run under them (the sweep and the side
  # stream's launches wait for each other on the device) -- the counter passes of the em workloads run the ONE-stream
  # loop (LASSO_EM_SIDE_STREAM=0: Gram product by lasso_gram_accumulate, objective on the main stream) and say so
  case $tag in em_*) export LASSO_EM_SIDE_STREAM=0; export LASSO_HBM_NOTE="counter passes ran the one-stream EM loop (LASSO_EM_SIDE_STREAM=0): PMC collection serialises kernels, under which the two-stream loop's device-side waits cannot be met" ;; *) unset LASSO_EM_SIDE_STREAM; unset LASSO_HBM_NOTE ;; esac
  bash $R/tools/pmc_mfma_busy.sh $D/mfma_busy.txt $R/bench.py $ARGS > /dev/null 2>&1
  bash $R/tools/prof_hbm.sh $D $DIV $PARGS > $D/hbm.log 2>&1
  unset LASSO_EM_SIDE_STREAM
  # the bench line LAST: it then finds this build's hbm_traffic.json beside it (gpurun_out/r06_<tag>/ is looked at first)
  python $R/bench.py $ARGS > $D/bench.json 2> $D/bench.err
  echo "== $tag"; head -c 600 $D/bench.json; echo; head -4 $D/kernel_stats.txt; tail -7 $D/hbm.log
done
