cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd $R && timeout 300 python -m pytest tests/test_dict_learning_gpu.py -x -q -m gpu -k ridge 2>&1 | tail -15 )
for a in "1024 256 8192" "1500 300 3000" "2048 256 8192" "256 64 4096"; do timeout 120 python $R/tools/bench_ridge.py $a 2>&1 | grep -v amdgpu.ids; done | tee $R/gpurun_out/r2_ridge_bench.log
rm -rf $R/gpurun_out/prof_ridge
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ridge/trace -o t -- python $R/tools/bench_ridge.py > /dev/null 2>&1
python $R/tools/summarize_prof.py $R/gpurun_out/prof_ridge > $R/gpurun_out/r2_ridge_prof.txt 2>&1
grep -i "chol\|ridge" $R/gpurun_out/r2_ridge_prof.txt
