#!/bin/bash
# Round 4, second GPU call: the generalised one-launch row passes -- new tests, the regression suites of the row passes, and a
# same-box A/B of the C2-L / C2-S bench lines between this build and the previous commit's library (pymc_amd/libnuts_prev.so).
export PYMC_AMD_HONOUR_NUTS_ENV=1
TAG=${1:-r04b}
STAGES=${2:-tra}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
if [[ $STAGES == *t* ]]; then
  timeout 1500 python -m pytest tests/test_gpu_rows_generalised.py -m gpu -q --timeout 900 -x -rA 2>&1 | grep -v "^PASSED\|^$" > $OUT/pytest_rows_$TAG.log; echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_rows_$TAG.log
  tail -25 $OUT/pytest_rows_$TAG.log
fi
if [[ $STAGES == *r* ]]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchmark_shapes.py -m gpu -q --timeout 900 -rA 2>&1 | grep -v "^PASSED\|^$" > $OUT/pytest_regr_$TAG.log; echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_regr_$TAG.log
  tail -8 $OUT/pytest_regr_$TAG.log
fi
if [[ $STAGES == *a* ]]; then
  B="python bench.py --steps 300 --warmup 300 --cpu-leapfrogs 0 --ess-tune 0"
  pick() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], 'leapfrog/s', round(j['leapfrog_steps_per_sec']), 'tree', round(j['mean_tree_size'],1), 'launch_us', round(1e3*j['roofline']['avg_launch_ms'],2), 'frac', round(j['roofline']['frac'],4))" $1 "$2"; }
  {
  echo "# same-box A/B (tag $TAG): python bench.py --steps 300 --warmup 300 --cpu-leapfrogs 0 --ess-tune 0 [--rows-per-group 80]; prev = the previous commit's library"
  for rep in 1 2; do
    PYMC_AMD_LIB=$R/pymc_amd/libnuts_prev.so $B > $OUT/ab_prev_$TAG.json 2> $OUT/ab_prev_$TAG.err; pick $OUT/ab_prev_$TAG.json "C2-L prev  "
    $B > $OUT/ab_new_$TAG.json 2> $OUT/ab_new_$TAG.err; pick $OUT/ab_new_$TAG.json "C2-L new   "
  done
  PYMC_AMD_LIB=$R/pymc_amd/libnuts_prev.so $B --rows-per-group 80 > $OUT/ab_prev_c2s_$TAG.json 2> $OUT/ab_prev_c2s_$TAG.err; pick $OUT/ab_prev_c2s_$TAG.json "C2-S prev  "
  $B --rows-per-group 80 > $OUT/ab_new_c2s_$TAG.json 2> $OUT/ab_new_c2s_$TAG.err; pick $OUT/ab_new_c2s_$TAG.json "C2-S new   "
  } 2>&1 | tee $OUT/ab_$TAG.txt
fi
