export PYMC_AMD_HONOUR_NUTS_ENV=1; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd /tmp
for W in 4 12; do
  rm -rf $OUT/pg$W
  NUTS_GLM_WG_PER_CU=$W timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/pg$W -o trace -- python $R/bench.py --workload glm --steps 30 --warmup 60 --cpu-leapfrogs 0 --ess-tune 0 > $OUT/pg$W.log 2>&1
  echo "== workgroups per CU $W"; python $R/tools/rocpd_summary.py $OUT/pg$W/trace_results.db | head -12 | cut -c1-150
  rm -rf $OUT/pg$W
done
