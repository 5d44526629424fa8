#!/bin/bash
# GPU-box round script: parity tests, bench, rocprofv3 kernel trace + PMC passes, summarised into gpurun_out/.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
nproc > $OUT/nproc_$TAG.txt
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_$TAG.log
timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?" >> $OUT/bench_$TAG.err
cd /tmp
PROF_ARGS="--steps 30 --warmup 60 --cpu-leapfrogs 0"
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o trace -- python $R/bench.py $PROF_ARGS > $OUT/prof_$TAG.log 2>&1; echo "prof rc=$?" >> $OUT/prof_$TAG.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 4 --warmup 8 --cpu-leapfrogs 0 > $OUT/pmc_fetch_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 4 --warmup 8 --cpu-leapfrogs 0 > $OUT/pmc_write_$TAG.log 2>&1
{
  echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py $PROF_ARGS   (tag $TAG)"
  grep -E '^\{' $OUT/prof_$TAG.log | head -1
  python $R/tools/rocpd_summary.py $OUT/prof_$TAG/trace_results.db --pmc $OUT/pmc_fetch_$TAG/pmc_results.db $OUT/pmc_write_$TAG/pmc_results.db
} > $OUT/profile_$TAG.txt 2>&1
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG   # raw rocpd databases are large; the summary is what is kept
tail -3 $OUT/pytest_gpu_$TAG.log; cat $OUT/bench_$TAG.json | head -c 600; echo; head -20 $OUT/profile_$TAG.txt
