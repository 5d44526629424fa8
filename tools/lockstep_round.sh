#!/bin/bash
# one GPU call: the chain-group tests, then the C3 lockstep bench (and optionally a kernel trace of the group mode)
TAG=${1:-x}; WHAT=${2:-tb}
OUT=gpurun_out; mkdir -p $OUT
if [[ $WHAT == *t* ]]; then timeout 900 python -m pytest tests/test_gpu_chain_group.py -m gpu -x -q 2>&1 | tail -15 > $OUT/lockstep_tests_$TAG.txt; cat $OUT/lockstep_tests_$TAG.txt; fi
if [[ $WHAT == *b* ]]; then timeout 600 python tools/lockstep_bench.py --tune ${TUNE:-300} --draws ${DRAWS:-300} > $OUT/lockstep_bench_$TAG.json 2> $OUT/lockstep_bench_$TAG.err; cat $OUT/lockstep_bench_$TAG.json; tail -3 $OUT/lockstep_bench_$TAG.err; fi
if [[ $WHAT == *p* ]]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_lockstep_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/lockstep_bench.py --tune 200 --draws 200 --modes group > $GRAFT_REPO_ROOT/$OUT/lockstep_prof_$TAG.json 2> $GRAFT_REPO_ROOT/$OUT/lockstep_prof_$TAG.err
  cd $GRAFT_REPO_ROOT
  DB=$(find $OUT/prof_lockstep_$TAG -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/lockstep_profile_$TAG.txt 2>&1; head -30 $OUT/lockstep_profile_$TAG.txt
fi
