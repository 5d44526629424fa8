#!/bin/bash
# rocprofv3 kernel trace + PMC of the C2-S bench (cache-resident size): summary into gpurun_out/profile_c2s_<tag>.txt
TAG=${1:-r03}; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--rows-per-group 80 --steps 300 --warmup 300 --cpu-leapfrogs 0 --ess-tune 0"
python bench.py $ARGS > $OUT/bench_c2s_$TAG.json 2> $OUT/bench_c2s_$TAG.err
cd /tmp; rm -rf $OUT/prof_c2s_$TAG $OUT/pmc_c2s_f_$TAG $OUT/pmc_c2s_w_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_c2s_$TAG -o trace -- python $R/bench.py $ARGS > $OUT/prof_c2s_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_c2s_f_$TAG -o pmc -- python $R/bench.py --rows-per-group 80 --steps 40 --warmup 40 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/pmc_c2s_f_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_c2s_w_$TAG -o pmc -- python $R/bench.py --rows-per-group 80 --steps 40 --warmup 40 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/pmc_c2s_w_$TAG.log 2>&1
{
  echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py $ARGS   (tag $TAG, kernel source hash $(cd $R; python -c 'import bench; print(bench.kernel_source_hash())'))"
  grep -E '^\{' $OUT/bench_c2s_$TAG.json | head -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('# un-profiled bench line: leapfrog/s', round(j['leapfrog_steps_per_sec']), 'mean tree', j['mean_tree_size'], 'HIP-event launch us', round(1e3*j['roofline']['avg_launch_ms'],2), 'leapfrog_frac', round(j['roofline']['leapfrog_frac'],4), '|', j['schedule'])"
  grep -E '^\{' $OUT/prof_c2s_$TAG.log | head -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('# profiled run: leapfrog/s', round(j['leapfrog_steps_per_sec']))"
  python $R/tools/rocpd_summary.py $OUT/prof_c2s_$TAG/trace_results.db --pmc $OUT/pmc_c2s_f_$TAG/pmc_results.db $OUT/pmc_c2s_w_$TAG/pmc_results.db
} > $OUT/profile_c2s_$TAG.txt 2>&1
rm -rf $OUT/prof_c2s_$TAG $OUT/pmc_c2s_f_$TAG $OUT/pmc_c2s_w_$TAG
head -30 $OUT/profile_c2s_$TAG.txt
