#!/bin/bash
# where a draw's time goes outside the row passes: kernel trace of a bench run whose second half is the timed region
# usage: bash tools/draw_anatomy.sh <tag> [bench args]
TAG=${1:-lab}; shift; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; rm -rf $OUT/prof_da_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_da_$TAG -o trace -- python $R/bench.py --cpu-leapfrogs 0 --ess-tune 0 "$@" > $OUT/prof_da_$TAG.log 2>&1
{ echo "# bench.py --cpu-leapfrogs 0 --ess-tune 0 $@"; python $R/tools/rocpd_summary.py $OUT/prof_da_$TAG/trace_results.db; } > $OUT/draw_anatomy_$TAG.txt 2>&1
rm -rf $OUT/prof_da_$TAG
tail -50 $OUT/draw_anatomy_$TAG.txt
