#!/bin/bash
# usage (GPU box, via gpurun): tools/round5_conv_profiles.sh   -> gpurun_out/r05_conv/*
# The records DESIGN.md 3.5 (round 5) quotes for the convolutional solver.  The debug / ablation libraries must have
# been built here first: tools/build_variant.sh cf_t conv_fused.hip -DLASSO_CF_TIMING; tools/ab_conv_fused_phases.sh build
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_conv
mkdir -p $O
cd $R
python tools/bench_conv.py > $O/conv.json 2> $O/conv.err
python tools/bench_conv_fused.py > $O/conv_fused.json 2> $O/conv_fused.err
{ echo "# tools/ab_conv.py env:LASSO_CONV_FUSED=0 (us per iteration at 20 iterations, tol = 0; product = the dispatch, other side = the two-kernel form)";
  python tools/ab_conv.py env:LASSO_CONV_FUSED=0; echo "# AB_CONV_CASES=fused"; AB_CONV_CASES=fused python tools/ab_conv.py env:LASSO_CONV_FUSED=0; } > $O/ab_conv_fused.txt 2>&1
python tools/conv_fused_timeline.py > $O/conv_fused_timeline.txt 2>&1
python tools/ab_conv_small_batches.py 2>&1 | grep -v amdgpu.ids > $O/ab_conv_small_batches.txt
python tools/stress_conv_fused.py 400 2>&1 | grep -v amdgpu.ids > $O/stress_conv_fused.txt
python tools/stress_conv.py 150 2>&1 | grep -v amdgpu.ids > $O/stress_conv.txt
bash tools/ab_conv_fused_phases.sh run > $O/ab_conv_fused_phases.txt 2>&1
rm -f $O/pmc_fused.txt
bash tools/pmc_counters.sh gpurun_out/r05_conv/pmc_fused.txt "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" conv_fused_kernel tools/bench_conv.py --no-cpu --case 0 > /dev/null
bash tools/pmc_counters.sh gpurun_out/r05_conv/pmc_fused.txt "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY" conv_fused_kernel tools/bench_conv.py --no-cpu --case 0 > /dev/null
bash tools/prof_conv_hbm.sh gpurun_out/r05_conv/hbm_fused.txt --case 0 > /dev/null
LASSO_CONV_FUSED=0 bash tools/prof_conv_hbm.sh gpurun_out/r05_conv/hbm_two_kernel.txt --case 0 > /dev/null
bash tools/prof_conv_hbm.sh gpurun_out/r05_conv/hbm_c3.txt --case 1 > /dev/null
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ckt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ckt -o t -- python $R/tools/bench_conv.py --no-cpu > /tmp/ckt.log 2>&1
python - > $O/conv_kernel_stats.txt <<'PY'
import csv, glob
f = glob.glob('/tmp/ckt/**/*kernel_stats.csv', recursive=True)
print("== kernel stats (rocprofv3 --kernel-trace --stats) of tools/bench_conv.py --no-cpu ==")
for r in sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%-100s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (r["Name"][:100], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
PY
ls -la $O
