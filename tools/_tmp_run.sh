cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
cd /tmp
for sh in "c5 8192" "c4 8192"; do set -- $sh; for rep in 1 2; do
python $GRAFT_REPO_ROOT/bench.py --workload em --shape $1 --rows $2 --steps 60 --warmup 10 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1 $2', d['ms_per_step'], d['objective_last_step'])"
done; done
python $GRAFT_REPO_ROOT/bench.py 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print('headline', d['value'], d['roofline']['frac'])"
