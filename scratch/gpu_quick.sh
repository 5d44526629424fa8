#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > $OUT/pytest_q.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_q.log
tail -30 $OUT/pytest_q.log
timeout 600 python bench.py --steps 100 --warmup 150 --cpu-leapfrogs 3 > $OUT/bench_q.json 2> $OUT/bench_q.err; echo "bench rc=$?"
cat $OUT/bench_q.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','leapfrog_steps_per_sec','mean_tree_size']}); print(d['roofline'])"
tail -3 $OUT/bench_q.err
