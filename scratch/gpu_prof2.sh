#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_q.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_q.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_q.log | tail -15
for side in 1 0; do
  NUTS_SIDE_STREAM=$side timeout 600 python bench.py --steps 40 --warmup 60 --cpu-leapfrogs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=$side', {k:d[k] for k in ['ms_per_step','leapfrog_steps_per_sec','mean_tree_size']}, d['roofline']['avg_launch_ms'])"
done
NUTS_SIDE_STREAM=1 timeout 600 python bench.py --steps 300 --warmup 700 --cpu-leapfrogs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=1 long', {k:d[k] for k in ['ms_per_step','leapfrog_steps_per_sec','mean_tree_size']}, d['roofline']['avg_launch_ms'])"
NUTS_SIDE_STREAM=0 timeout 600 python bench.py --steps 300 --warmup 700 --cpu-leapfrogs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=0 long', {k:d[k] for k in ['ms_per_step','leapfrog_steps_per_sec','mean_tree_size']}, d['roofline']['avg_launch_ms'])"
