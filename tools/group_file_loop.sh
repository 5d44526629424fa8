#!/bin/bash
# File-order replay of tests/test_gpu_chain_group.py, N times in fresh processes (the rows group's bitwise test differed in
# about one run in ten of its FILE in round 5, never in a process of its own).  usage: bash tools/group_file_loop.sh <tag> <N> [pytest -k expression]
TAG=${1:-r06}; N=${2:-40}; K=${3:-"not oracle_chains"}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
: > $OUT/group_file_loop_$TAG.log
for i in $(seq 1 $N); do
  timeout 400 python -m pytest tests/test_gpu_chain_group.py -q --timeout 300 -rA -p no:cacheprovider -k "$K" 2>&1 | grep -i "mismatch\|sampled again\|passed\|failed\|^FAILED\|Error" | sed "s/^/run $i: /" >> $OUT/group_file_loop_$TAG.log
done
echo "runs: $N; runs with a failure: $(grep -c 'failed' $OUT/group_file_loop_$TAG.log); mismatch lines: $(grep -ci 'mismatch' $OUT/group_file_loop_$TAG.log)" | tee -a $OUT/group_file_loop_$TAG.log
