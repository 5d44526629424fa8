#!/bin/bash
# Round 4, first GPU call: configs[3] parity at its literal shape, C4 / C5 re-measured on the current sources (+ rocprof summaries),
# the idle-gap lab, and a kernel trace of the C2-L bench with the gap quantiles.   usage: bash tools/r04_round1.sh <tag>
export PYMC_AMD_HONOUR_NUTS_ENV=1
TAG=${1:-r04a}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
nproc > $OUT/nproc_$TAG.txt
timeout 900 python -m pytest tests/test_advi.py -m gpu -q --timeout 800 -rA 2>&1 | grep -v "^PASSED\|^$" > $OUT/pytest_advi_$TAG.log; echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_advi_$TAG.log
cd /tmp
# C4 / C5: the line, then the same command under rocprofv3 (shorter)
timeout 600 python $R/tools/aux_bench.py --workload c4 > $OUT/aux_c4_$TAG.json 2> $OUT/aux_c4_$TAG.err
rm -rf $OUT/prof_c4_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_c4_$TAG -o trace -- python $R/tools/aux_bench.py --workload c4 --steps 1000 --cpu-steps 0 > $OUT/prof_c4_$TAG.log 2>&1
{ echo "# command: rocprofv3 --kernel-trace --stats -- python tools/aux_bench.py --workload c4 --steps 1000 --cpu-steps 0 (tag $TAG)"; grep -E '^\{' $OUT/prof_c4_$TAG.log | head -1
  python $R/tools/rocpd_summary.py $OUT/prof_c4_$TAG/trace_results.db; } > $OUT/profile_c4_$TAG.txt 2>&1
rm -rf $OUT/prof_c4_$TAG
timeout 600 python $R/tools/aux_bench.py --workload c5 > $OUT/aux_c5_$TAG.json 2> $OUT/aux_c5_$TAG.err
rm -rf $OUT/prof_c5_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_c5_$TAG -o trace -- python $R/tools/aux_bench.py --workload c5 --steps 300 --cpu-steps 0 > $OUT/prof_c5_$TAG.log 2>&1
{ echo "# command: rocprofv3 --kernel-trace --stats -- python tools/aux_bench.py --workload c5 --steps 300 --cpu-steps 0 (tag $TAG)"; grep -E '^\{' $OUT/prof_c5_$TAG.log | head -1
  python $R/tools/rocpd_summary.py $OUT/prof_c5_$TAG/trace_results.db; } > $OUT/profile_c5_$TAG.txt 2>&1
rm -rf $OUT/prof_c5_$TAG
# idle-gap lab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/gap_lab.hip -o /tmp/gap_lab > $OUT/gap_lab_$TAG.txt 2>&1
rm -rf $OUT/prof_gap_$TAG
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_gap_$TAG -o gap -- /tmp/gap_lab >> $OUT/gap_lab_$TAG.txt 2>&1
python $R/tools/rocpd_summary.py $OUT/prof_gap_$TAG/gap_results.db >> $OUT/gap_lab_$TAG.txt 2>&1
rm -rf $OUT/prof_gap_$TAG
# C2-L kernel trace with gap quantiles
rm -rf $OUT/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o trace -- python $R/bench.py --steps 30 --warmup 60 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/prof_$TAG.log 2>&1; echo "prof rc=$?" >> $OUT/prof_$TAG.log
{ echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 60 --cpu-leapfrogs 0 --ess-tune 0 (tag $TAG)"; grep -E '^\{' $OUT/prof_$TAG.log | head -1
  python $R/tools/rocpd_summary.py $OUT/prof_$TAG/trace_results.db; } > $OUT/profile_$TAG.txt 2>&1
rm -rf $OUT/prof_$TAG
cd $R
tail -3 $OUT/pytest_advi_$TAG.log; head -c 700 $OUT/aux_c4_$TAG.json; echo; head -c 700 $OUT/aux_c5_$TAG.json; echo; grep "gap_kernel" $OUT/gap_lab_$TAG.txt | head -40; grep -A3 "idle gap between" $OUT/profile_$TAG.txt
