#!/bin/bash
# usage (GPU box, via gpurun): tools/prof_hbm.sh <out_dir> <steps_divisor> <bench.py args ...>
# HBM traffic of ONE bench.py workload: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, as
# MI355X_MICROARCH.md prescribes; KiB units; FETCH_SIZE x2 on gfx950) over the same command, summed over EVERY dispatch
# of the process and divided by the number of steps (solves / EM steps) the command executes -> <out_dir>/hbm_traffic.json
# with the per-kernel split, the kernel bench.py names for the workload and the digest of the kernel sources (bench.py
# reports the figure only while the sources are the ones that were measured).
OUT=$1; DIV=$2; shift 2
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hbm_f /tmp/hbm_w
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/hbm_f -o p -- python $R/bench.py "$@" > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/hbm_w -o p -- python $R/bench.py "$@" > $OUT/pmc_write.log 2>&1
python - "$OUT" "$DIV" "$R" "$@" <<'PY'
import csv, glob, json, os, sys, collections
out, div, root, args = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4:]
sys.path.insert(0, root)
def collect(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
                acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc
fetch, write = collect("/tmp/hbm_f", "FETCH_SIZE"), collect("/tmp/hbm_w", "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, [0.0, 0]); w, nw = write.get(k, [0.0, 0])
    rows.append({"kernel": k, "dispatches": max(nf, nw), "fetch_kib_total": f, "write_kib_total": w,
                 "hbm_bytes_per_step": (2 * f + w) * 1024 / div})
rows.sort(key=lambda r: -r["hbm_bytes_per_step"])
total = sum(r["hbm_bytes_per_step"] for r in rows)
import bench
rec = {"hbm_bytes_per_step": total, "steps_divisor": div, "command": "bench.py " + " ".join(args),
       "conditions": os.environ.get("LASSO_HBM_NOTE", ""),
       "kernels": rows[:12], "sources_digest": bench.sources_digest(),
       "note": "sum over every dispatch of the process of FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM "
               "section) + WRITE_SIZE, KiB -> bytes, separate --pmc passes, divided by the steps the command executes"}
json.dump(rec, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("hbm_bytes_per_step", "steps_divisor", "command")}))
for r in rows[:6]:
    print("  %-70s n=%-5d %10.1f MB/step" % (r["kernel"][:70], r["dispatches"], r["hbm_bytes_per_step"] / 1e6))
PY
