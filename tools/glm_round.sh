#!/bin/bash
# GLM node under NUTS (configs[3]'s model: 1 M rows x 512 covariates): bench line, rocprofv3 kernel trace, PMC passes.
# usage: bash tools/glm_round.sh <tag> [stages: b(ench) p(rofile) s(weep of workgroups per CU)]
export PYMC_AMD_HONOUR_NUTS_ENV=1
TAG=${1:-glm}
STAGES=${2:-bp}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
HASH=$(python -c "import bench; print(bench.kernel_source_hash())")
if [[ $STAGES == *b* ]]; then
  timeout 900 python bench.py --workload glm --steps 200 --warmup 300 --ess-tune 0 > $OUT/bench_glm_$TAG.json 2> $OUT/bench_glm_$TAG.err; echo "bench rc=$?" >> $OUT/bench_glm_$TAG.err
  head -c 1800 $OUT/bench_glm_$TAG.json; echo
fi
if [[ $STAGES == *s* ]]; then
  for W in 4 8 12 16; do
    NUTS_GLM_WG_PER_CU=$W timeout 600 python bench.py --workload glm --steps 60 --warmup 100 --ess-tune 0 --cpu-leapfrogs 0 > $OUT/bench_glm_w${W}_$TAG.json 2> /dev/null
    python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print('workgroups per CU', sys.argv[2], 'leapfrog/s', round(j['leapfrog_steps_per_sec'],1), 'launch_us', round(1e3*j['roofline']['avg_launch_ms'],1), 'frac', round(j['roofline']['frac'],4))" $OUT/bench_glm_w${W}_$TAG.json $W
  done | tee $OUT/glm_sweep_$TAG.txt
fi
if [[ $STAGES == *p* ]]; then
  cd /tmp
  PA="--workload glm --steps 30 --warmup 60 --cpu-leapfrogs 0 --ess-tune 0"
  rm -rf $OUT/prof_glm_$TAG $OUT/pmc_glm_f_$TAG $OUT/pmc_glm_w_$TAG
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_glm_$TAG -o trace -- python $R/bench.py $PA > $OUT/prof_glm_$TAG.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_glm_f_$TAG -o pmc -- python $R/bench.py --workload glm --steps 4 --warmup 8 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/pmc_glm_f_$TAG.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_glm_w_$TAG -o pmc -- python $R/bench.py --workload glm --steps 4 --warmup 8 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/pmc_glm_w_$TAG.log 2>&1
  { echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py $PA   (tag $TAG, kernel source hash $HASH)"
    grep -E '^\{' $OUT/prof_glm_$TAG.log | head -1
    python $R/tools/rocpd_summary.py $OUT/prof_glm_$TAG/trace_results.db --pmc $OUT/pmc_glm_f_$TAG/pmc_results.db $OUT/pmc_glm_w_$TAG/pmc_results.db
  } > $OUT/profile_glm_$TAG.txt 2>&1
  python $R/tools/rocpd_summary.py --traffic $OUT/pmc_glm_f_$TAG/pmc_results.db $OUT/pmc_glm_w_$TAG/pmc_results.db $OUT/traffic_glm_$TAG.json k_glm_rows $TAG $HASH k_glm_rows_bytes_per_launch
  python $R/tools/rocpd_summary.py --launch-time $OUT/prof_glm_$TAG/trace_results.db $OUT/launch_time_glm_$TAG.json "k_glm_rows<" $TAG $HASH "python bench.py $PA"
  rm -rf $OUT/prof_glm_$TAG $OUT/pmc_glm_f_$TAG $OUT/pmc_glm_w_$TAG
  cd $R
  head -14 $OUT/profile_glm_$TAG.txt | cut -c1-200; cat $OUT/traffic_glm_$TAG.json | head -12
fi
