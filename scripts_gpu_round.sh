#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts_gpu_round.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_$TAG.log
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?" >> $OUT/bench_$TAG.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o trace -- python $R/bench.py --steps 30 --warmup 60 --cpu-leapfrogs 0 > $OUT/prof_$TAG.log 2>&1; echo "prof rc=$?" >> $OUT/prof_$TAG.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 4 --warmup 8 --cpu-leapfrogs 0 > $OUT/pmc_fetch_$TAG.log 2>&1; echo "pmc rc=$?" >> $OUT/pmc_fetch_$TAG.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 4 --warmup 8 --cpu-leapfrogs 0 > $OUT/pmc_write_$TAG.log 2>&1; echo "pmc rc=$?" >> $OUT/pmc_write_$TAG.log
# keep only the small summaries (the raw kernel trace can be 100s of MB)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -laR $OUT | head -80 > $OUT/ls_$TAG.txt
