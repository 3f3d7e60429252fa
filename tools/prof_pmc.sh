#!/bin/bash
# usage: tools/prof_pmc.sh <tag>   (run on the GPU box via gpurun)
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-time-to-tol"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o p -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o p -- $CMD > $OUT/pmc4.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc5 -o p -- $CMD > $OUT/pmc5.log 2>&1
python $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
