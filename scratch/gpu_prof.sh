#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_q.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_q.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_q.log | tail -15
cd /tmp
rm -rf $OUT/prof_q
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_q -o trace -- python $R/bench.py --steps 20 --warmup 40 --cpu-leapfrogs 0 > $OUT/prof_q.log 2>&1; echo "prof rc=$?"
grep -E '^\{' $OUT/prof_q.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','leapfrog_steps_per_sec','mean_tree_size']}); print(d['roofline'])"
python $R/tools/rocpd_summary.py $OUT/prof_q/trace_results.db | head -40
