#!/bin/bash
# usage (GPU box): tools/pmc_counters.sh <out.txt> "<COUNTER ...>" <kernel name substring> <python script (repo-relative)> [args...]; out.txt repo-relative
# Per-dispatch averages of the given SQ / GRBM counters for the kernels whose name holds the substring (one rocprofv3
# --pmc pass; no tracing options beside it).
OUT=$GRAFT_REPO_ROOT/$1; CTR=$2; KEY=$3; shift 3; SCRIPT=$GRAFT_REPO_ROOT/$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmcc
timeout 600 rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmcc -o p -- python $SCRIPT "$@" > /tmp/pmcc.log 2>&1 || tail -5 /tmp/pmcc.log
python - "$KEY" >> $OUT <<'PY'
import csv, glob, collections, sys
key = sys.argv[1]
fs = glob.glob('/tmp/pmcc/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        if key in r['Kernel_Name']:
            agg[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        print('   %-32s %16.1f per dispatch (%d dispatches)' % (c, sum(vals) / len(vals), len(vals)))
PY
cat $OUT
