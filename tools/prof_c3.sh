#!/bin/bash
# usage: tools/prof_c3.sh   (on the GPU box via gpurun) -- config-3 line search, bf16 kernels:
# kernel trace + HBM bytes (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) + MFMA busy.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_c3pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_c3.py --bf16-only --reps 3"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o p -- $CMD > $OUT/pmc_mfma.log 2>&1
python - <<PY
import csv, glob, os, json
out = "$OUT"
def per_kernel(counter, d):
    acc = {}
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc.setdefault(r["Kernel_Name"].split("(")[0][-60:], []).append(float(r["Counter_Value"]))
    return acc
dur = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0][-60:]] = (float(r["AverageNs"]), int(r["Calls"]), float(r["MaxNs"]))
fetch, write = per_kernel("FETCH_SIZE", "pmc_fetch"), per_kernel("WRITE_SIZE", "pmc_write")
mf, gui = per_kernel("SQ_VALU_MFMA_BUSY_CYCLES", "pmc_mfma"), per_kernel("GRBM_GUI_ACTIVE", "pmc_mfma")
rows = []
for name in dur:
    if "bt16" not in name and "bt_finish" not in name: continue
    fv, wv = fetch.get(name, []), write.get(name, [])
    # executed launches only (trial kernels that exit at once move ~nothing)
    big = [i for i, v in enumerate(fv) if v > 1024] or list(range(len(fv)))
    fk = sum(fv[i] for i in big) / max(len(big), 1) if fv else None
    wk = sum(wv[i] for i in big if i < len(wv)) / max(len(big), 1) if wv else None
    rows.append({"kernel": name, "avg_ns_all_launches": dur[name][0], "max_ns": dur[name][2], "launches": dur[name][1],
                 "fetch_kib_executed": fk, "write_kib_executed": wk,
                 "hbm_bytes_executed": None if fk is None or wk is None else (2 * fk + wk) * 1024,
                 "mfma_busy_cycles_mean": (sum(mf[name]) / len(mf[name])) if name in mf else None,
                 "gui_active_mean": (sum(gui[name]) / len(gui[name])) if name in gui else None})
json.dump(rows, open(os.path.join(out, "c3_kernels.json"), "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
