#!/bin/bash
# Same-box A/B of the generalised one-launch row pass: the benchmark's own model against the same rows under
# other hyper-priors / with further variables, at C2-L (group-aligned pass) and C2-S (group-block pass), and -- for the price of
# falling off the one-launch pass -- the same variants with the auxiliary workgroups switched off (NUTS_GA_AUX=0: general path).
# usage: bash tools/variants_ab.sh <tag>
export PYMC_AMD_HONOUR_NUTS_ENV=1
TAG=${1:-ab}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
B="python bench.py --steps 300 --warmup 300 --cpu-leapfrogs 0 --ess-tune 0"
pick() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], 'leapfrog/s', round(j['leapfrog_steps_per_sec']), 'tree', round(j['mean_tree_size'],1), 'launch_us', round(1e3*j['roofline']['avg_launch_ms'],2), 'frac', round(j['roofline']['frac'],4), '|', j['schedule'][:38])" $1 "$2"; }
{
echo "# $B [--rows-per-group 80] [--variant V]   (tag $TAG; one box, one after the other)"
for RPG in 4000 80; do
  $B --rows-per-group $RPG > $OUT/var_base_$TAG.json 2> $OUT/var_base_$TAG.err; pick $OUT/var_base_$TAG.json "rows/group $RPG  benchmark model        "
  for V in ${VLIST:-halfcauchy exponential lognormal gamma datapriors extra}; do
    $B --rows-per-group $RPG --variant $V > $OUT/var_${V}_$TAG.json 2> $OUT/var_${V}_$TAG.err; pick $OUT/var_${V}_$TAG.json "rows/group $RPG  $(printf %-12s $V) one-launch "
  done
  for V in ${GLIST-halfcauchy extra}; do
    NUTS_GA_AUX=0 $B --rows-per-group $RPG --variant $V > $OUT/var_${V}_gen_$TAG.json 2> $OUT/var_${V}_gen_$TAG.err; pick $OUT/var_${V}_gen_$TAG.json "rows/group $RPG  $(printf %-12s $V) round-3 path"
  done
  $B --rows-per-group $RPG > $OUT/var_base2_$TAG.json 2> $OUT/var_base2_$TAG.err; pick $OUT/var_base2_$TAG.json "rows/group $RPG  benchmark model (again)"
done
} 2>&1 | tee $OUT/variants_ab_$TAG.txt
