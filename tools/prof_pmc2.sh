#!/bin/bash
# deeper PMC passes for the fista kernel (issue / TA / TCP / icache); usage: tools/prof_pmc2.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-time-to-tol"
i=10
for SET in \
 "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_IFETCH SQ_WAIT_INST_ANY" \
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum" \
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum" \
 "SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS" ; do
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
python $R/tools/summarize_prof.py $OUT > $OUT/summary2.txt 2>&1
cat $OUT/summary2.txt
