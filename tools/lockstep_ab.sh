#!/bin/bash
# same-box A/B of library builds on the C3 chain-group bench: kernel trace of the group mode per library
# usage: bash tools/lockstep_ab.sh <tag> <lib1> [<lib2> ...]     (lib = path of a libnuts build, or "default")
TAG=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename $L .so)
  if [ "$L" != default ]; then export PYMC_AMD_LIB=$R/$L; else unset PYMC_AMD_LIB; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_ab_${N}_$TAG -o trace -- python $R/tools/lockstep_bench.py --tune 200 --draws 200 --modes ${MODES:-group} > $OUT/lockstep_ab_${N}_$TAG.json 2> $OUT/lockstep_ab_${N}_$TAG.err
  DB=$(find $OUT/prof_ab_${N}_$TAG -name "*.db" | head -1)
  echo "== $N"; python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        j=json.loads(l); print(j['mode'], round(j['leapfrog_steps_per_sec']), j['launches_by_chains_carried'], j.get('draws_bitwise_equal_to_first_multi_chain_mode'))
" $OUT/lockstep_ab_${N}_$TAG.json
  python $R/tools/rocpd_summary.py $DB 2>/dev/null | grep -E "k_mvn_aligned|k_mva_control|GPU busy" | head -8
  rm -rf $OUT/prof_ab_${N}_$TAG
done 2>&1 | tee $OUT/lockstep_ab_$TAG.txt
