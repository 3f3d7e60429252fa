#!/bin/bash
# usage (GPU box, via gpurun): tools/prof_conv_hbm.sh <out_file> [bench_conv.py args ...]
# HBM traffic per dispatch of the convolutional solver's kernels: two rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950,
# WRITE_SIZE; KiB; separate runs) over tools/bench_conv.py --no-cpu.
OUT=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/chbm_f /tmp/chbm_w
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/chbm_f -o p -- python $R/tools/bench_conv.py --no-cpu "$@" > /tmp/chbm_f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/chbm_w -o p -- python $R/tools/bench_conv.py --no-cpu "$@" > /tmp/chbm_w.log 2>&1
python - "$R/$OUT" <<'PY'
import csv, glob, os, sys, collections
def collect(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
                acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc
fetch, write = collect("/tmp/chbm_f", "FETCH_SIZE"), collect("/tmp/chbm_w", "WRITE_SIZE")
lines = ["# per dispatch: FETCH_SIZE x 2 (gfx950) and WRITE_SIZE, KiB -> MB; tools/prof_conv_hbm.sh"]
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, [0, 1])[0] + write.get(k, [0, 1])[0])):
    f, nf = fetch.get(k, [0.0, 1]); w, nw = write.get(k, [0.0, 1])
    lines.append("%-70s dispatches=%5d read %8.2f MB  written %8.2f MB per dispatch" % (k[:70], max(nf, nw), 2 * f * 1024 / max(nf, 1) / 1e6, w * 1024 / max(nw, 1) / 1e6))
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
PY
