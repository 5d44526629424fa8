#!/bin/bash
# GLM row kernel: row-iterations in flight per wave (1 / 2 / 3) x workgroups per CU (4 / 8), same box.  The variants are builds of the
# same sources with -DGLM_PF=n (pymc_amd/libnuts_pf<n>.so; libnuts_mi355.so = the committed default).
export PYMC_AMD_HONOUR_NUTS_ENV=1
TAG=${1:-pf}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -m pytest tests/test_glm_node.py -m gpu -q -x 2>&1 | tail -2
for LIB in pf1 mi355 pf3; do
  [ -f pymc_amd/libnuts_$LIB.so ] || continue
  for W in 4 8; do
    PYMC_AMD_LIB=$R/pymc_amd/libnuts_$LIB.so NUTS_GLM_WG_PER_CU=$W timeout 600 python bench.py --workload glm --steps 60 --warmup 100 --ess-tune 0 --cpu-leapfrogs 0 > $OUT/glm_${LIB}_w${W}_$TAG.json 2> /dev/null
    python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], 'workgroups per CU', sys.argv[3], 'leapfrog/s', round(j['leapfrog_steps_per_sec'],1), 'launch_us', round(1e3*j['roofline']['avg_launch_ms'],1), 'frac', round(j['roofline']['frac'],4))" $OUT/glm_${LIB}_w${W}_$TAG.json "lib $LIB (mi355 = default build, GLM_PF=2)" $W
  done
done | tee $OUT/glm_pf_ab_$TAG.txt
