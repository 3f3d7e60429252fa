#!/bin/bash
# usage (GPU box): bash tools/round6_final.sh    the records of the final tree: GPU tests, smoke, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final; mkdir -p $O; cd $R
python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1
tail -3 $O/gpu_tests.txt
cp $R/gpurun_out/parity_margins.json $O/parity_margins.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 400 $O/bench_default.json; echo
python -c "import sys; sys.path.insert(0, '.'); import bench; print(bench.sources_digest())" > $O/sources_digest.txt
