#!/bin/bash
# GPU-box round script: parity tests, bench, rocprofv3 kernel trace + PMC passes, summarised into gpurun_out/.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [stages]   stages: any of t(ests) b(ench) p(rofile) c(3) x (only C3's counter passes) s(C2-S) e(ss study) m (smoke + N = 2 on one GPU)
export PYMC_AMD_HONOUR_NUTS_ENV=1   # the NUTS_* variables below reach the engine as schedule options (nuts_set_option)
TAG=${1:-r02}
STAGES=${2:-tbp}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
nproc > $OUT/nproc_$TAG.txt
HASH=$(python -c "import bench; print(bench.kernel_source_hash())")
if [[ $STAGES == *t* ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rA 2>&1 | grep -v "^PASSED\|^$" > $OUT/pytest_gpu_$TAG.log; echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_gpu_$TAG.log
fi
if [[ $STAGES == *b* ]]; then
  timeout 900 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?" >> $OUT/bench_$TAG.err
fi
if [[ $STAGES == *p* ]]; then
  cd /tmp
  PROF_ARGS="--steps 30 --warmup 60 --cpu-leapfrogs 0 --ess-tune 0"
  rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o trace -- python $R/bench.py $PROF_ARGS > $OUT/prof_$TAG.log 2>&1; echo "prof rc=$?" >> $OUT/prof_$TAG.log
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$TAG -o pmc -- python $R/bench.py --steps 4 --warmup 8 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/pmc_fetch_$TAG.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_$TAG -o pmc -- python $R/bench.py --steps 4 --warmup 8 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/pmc_write_$TAG.log 2>&1
  {
    echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py $PROF_ARGS   (tag $TAG, kernel source hash $HASH)"
    grep -E '^\{' $OUT/prof_$TAG.log | head -1
    python $R/tools/rocpd_summary.py $OUT/prof_$TAG/trace_results.db --pmc $OUT/pmc_fetch_$TAG/pmc_results.db $OUT/pmc_write_$TAG/pmc_results.db
  } > $OUT/profile_$TAG.txt 2>&1
  python $R/tools/rocpd_summary.py --launch-time $OUT/prof_$TAG/trace_results.db $OUT/launch_time_$TAG.json "k_rows_ga<" $TAG $HASH "python bench.py $PROF_ARGS"
  python $R/tools/rocpd_summary.py --traffic $OUT/pmc_fetch_$TAG/pmc_results.db $OUT/pmc_write_$TAG/pmc_results.db $OUT/traffic_$TAG.json k_rows $TAG $HASH
  rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG   # raw rocpd databases are large; the summary is what is kept
  cd $R
fi
if [[ $STAGES == *c* ]]; then
  timeout 600 python bench.py --workload c3 > $OUT/bench_c3_$TAG.json 2> $OUT/bench_c3_$TAG.err; echo "bench rc=$?" >> $OUT/bench_c3_$TAG.err
  cd /tmp; rm -rf $OUT/prof_c3_$TAG
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_c3_$TAG -o trace -- python $R/bench.py --workload c3 --steps 100 --warmup 200 --cpu-leapfrogs 0 > $OUT/prof_c3_$TAG.log 2>&1
  { echo "# command: rocprofv3 --kernel-trace --stats -- python bench.py --workload c3 --steps 100 --warmup 200 --cpu-leapfrogs 0 (tag $TAG)"; grep -E '^\{' $OUT/prof_c3_$TAG.log | head -1
    python $R/tools/rocpd_summary.py $OUT/prof_c3_$TAG/trace_results.db; } > $OUT/profile_c3_$TAG.txt 2>&1
  python $R/tools/rocpd_summary.py --launch-time $OUT/prof_c3_$TAG/trace_results.db $OUT/launch_time_c3_$TAG.json "k_mvn_aligned<" $TAG $HASH "python bench.py --workload c3 --steps 100 --warmup 200 --cpu-leapfrogs 0"
  rm -rf $OUT/prof_c3_$TAG; cd $R
fi
if [[ $STAGES == *c* || $STAGES == *x* ]]; then   # (x: only these) counter passes of the C3 launch -- what crosses the fabric into the XCDs' L2s
  cd /tmp; rm -rf $OUT/pmc_c3_f_$TAG $OUT/pmc_c3_w_$TAG
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_c3_f_$TAG -o pmc -- python $R/bench.py --workload c3 --steps 30 --warmup 60 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/pmc_c3_f_$TAG.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_c3_w_$TAG -o pmc -- python $R/bench.py --workload c3 --steps 30 --warmup 60 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/pmc_c3_w_$TAG.log 2>&1
  python $R/tools/rocpd_summary.py --traffic $OUT/pmc_c3_f_$TAG/pmc_results.db $OUT/pmc_c3_w_$TAG/pmc_results.db $OUT/traffic_c3_$TAG.json k_mvn_aligned $TAG $HASH k_mvn_aligned_bytes_per_launch ". What the counter sees crosses the fabric into the XCDs' L2s: with P (33.5 MB) resident in the 256 MiB Infinity Cache it is not HBM traffic"
  { echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE -- python bench.py --workload c3 --steps 30 --warmup 60 --cpu-leapfrogs 0 --ess-tune 0 (tag $TAG, kernel source hash $HASH)"
    python $R/tools/rocpd_summary.py $OUT/pmc_c3_f_$TAG/pmc_results.db --pmc $OUT/pmc_c3_f_$TAG/pmc_results.db $OUT/pmc_c3_w_$TAG/pmc_results.db; } > $OUT/pmc_c3_$TAG.txt 2>&1
  rm -rf $OUT/pmc_c3_f_$TAG $OUT/pmc_c3_w_$TAG; cd $R
fi
if [[ $STAGES == *s* ]]; then
  bash tools/c2s_profile.sh $TAG > /dev/null 2>&1
  timeout 600 python bench.py --rows-per-group 80 --cpu-leapfrogs 0 > $OUT/bench_c2s_$TAG.json 2> $OUT/bench_c2s_$TAG.err
fi
if [[ $STAGES == *e* ]]; then
  timeout 900 python tools/ess_study.py > $OUT/ess_study_$TAG.json 2> $OUT/ess_study_$TAG.err
fi
tail -3 $OUT/pytest_gpu_$TAG.log 2>/dev/null; head -c 1500 $OUT/bench_$TAG.json 2>/dev/null; echo; head -24 $OUT/profile_$TAG.txt 2>/dev/null
if [[ $STAGES == *m* ]]; then   # smoke() + the N = 2 launch of bench.py with both ranks on this one GPU (test mode of the distributed path)
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke_$TAG.log
  NUTS_GA_TREE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --share-gpu --steps 30 --warmup 30 --cpu-leapfrogs 0 > $OUT/bench_n2_shared_$TAG.json 2> $OUT/bench_n2_shared_$TAG.err; echo "n2 rc=$?" >> $OUT/bench_n2_shared_$TAG.err
  tail -3 $OUT/smoke_$TAG.log; head -c 600 $OUT/bench_n2_shared_$TAG.json; tail -2 $OUT/bench_n2_shared_$TAG.err
fi
