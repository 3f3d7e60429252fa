#!/bin/bash
# usage: tools/prof_pmc.sh <tag> [extra bench.py args, e.g. --rows 512]   (run on the GPU box via gpurun)
TAG=${1:-r01}
shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-time-to-tol --no-shards --no-extras $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o p -- $CMD > $OUT/pmc3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o p -- $CMD > $OUT/pmc4.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc5 -o p -- $CMD > $OUT/pmc5.log 2>&1
python $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
python - <<PY
import csv, glob, json, os
out = "$OUT"
kernels = set()
def mean(counter, d):
    vals = []
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "fista" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
                kernels.add(r["Kernel_Name"].split("(")[0].replace("void ", "").strip())
    return sum(vals) / len(vals) if vals else None
fetch, write = mean("FETCH_SIZE", "pmc3"), mean("WRITE_SIZE", "pmc4")
if fetch is not None and write is not None:
    # FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-reports wide reads by 2x
    json.dump({"fetch_size_kib": fetch, "write_size_kib": write,
               "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
               "kernel": sorted(kernels)[0] if len(kernels) == 1 else sorted(kernels),   # bench.py reports the figure only for this kernel
               "note": "FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, "
                       "separate --pmc passes, per 100-iteration launch"},
              open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
PY
