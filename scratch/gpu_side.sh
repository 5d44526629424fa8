#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
for cfg in "1 16" "1 32" "1 64" "0 32" "0 64"; do
  set -- $cfg
  NUTS_SIDE_STREAM=$1 NUTS_ROWS_WAVES_PER_CU=$2 timeout 600 python bench.py --steps 40 --warmup 60 --cpu-leapfrogs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('side=$1 wpc=$2', {k:round(d[k],1) for k in ['ms_per_step','leapfrog_steps_per_sec','mean_tree_size']}, round(d['roofline']['avg_launch_ms']*1e3,1))"
done
