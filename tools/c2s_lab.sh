#!/bin/bash
# C2-S (cache-resident size) under the schedules the engine has: group-block pass (default for small groups), general path,
# group-aligned pass forced.  usage: bash tools/c2s_lab.sh [tag]
export PYMC_AMD_HONOUR_NUTS_ENV=1   # the NUTS_* variables below reach the engine as schedule options (nuts_set_option)
TAG=${1:-lab}
OUT=gpurun_out; mkdir -p $OUT
B="python bench.py --rows-per-group 80 --steps 400 --warmup 400 --cpu-leapfrogs 0 --ess-tune 0"
pick() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], 'lps', round(j['leapfrog_steps_per_sec']), 'tree', round(j['mean_tree_size'],1), 'launch_ms', round(j['roofline']['avg_launch_ms'],5), j['schedule'][:60])" $1 "$2"; }
{
$B > $OUT/c2s_gb_$TAG.json 2>$OUT/c2s_gb_$TAG.err; pick $OUT/c2s_gb_$TAG.json "default (group-block)"
for G in 8 16; do NUTS_ROWS_GPW=$G $B > $OUT/c2s_gb${G}_$TAG.json 2>$OUT/c2s_gb${G}_$TAG.err; pick $OUT/c2s_gb${G}_$TAG.json "group-block GPW=$G"; done
NUTS_ROWS_GB=0 $B > $OUT/c2s_general_$TAG.json 2>$OUT/c2s_general_$TAG.err; pick $OUT/c2s_general_$TAG.json "general path"
NUTS_ROWS_GB=0 NUTS_ROWS_GA=2 NUTS_ROWS_GA_W=1 $B > $OUT/c2s_ga_w1_$TAG.json 2>$OUT/c2s_ga_w1_$TAG.err; pick $OUT/c2s_ga_w1_$TAG.json "group-aligned W=1"
} 2>&1 | tee $OUT/c2s_lab_$TAG.txt
