#!/bin/bash
# cycle-level A/B (clock independent): SQ_BUSY_CYCLES/32 per dispatch for each variant lib
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/abc_$v
  LASSO_HIP_LIB=$GRAFT_REPO_ROOT/variants/lib$v.so timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/abc_$v -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-time-to-tol > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/abc_$v/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "fista" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in acc.items()}
cyc=m["SQ_BUSY_CYCLES"]/32
print("$v cycles/dispatch=%.4g  mfma_busy_frac=%.3f  wait_any_frac=%.3f"%(cyc, m["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/cyc, m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
done
