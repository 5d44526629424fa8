#!/bin/bash
# one measurement round of the group-block pass: C2-S bench (default build) + phase stamps of leaves j = 0, 1, 3, 7 (ticks build)
export PYMC_AMD_HONOUR_NUTS_ENV=1   # the NUTS_* variables below reach the engine as schedule options (nuts_set_option)
TAG=${1:-lab}; OUT=gpurun_out; mkdir -p $OUT
B="python bench.py --rows-per-group 80 --steps 400 --warmup 400 --cpu-leapfrogs 0 --ess-tune 0"
{
$B > $OUT/c2s_gb_$TAG.json 2>$OUT/c2s_gb_$TAG.err
python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print('C2-S lps', round(j['leapfrog_steps_per_sec']), 'tree', round(j['mean_tree_size'],1), 'launch_us', round(1e3*j['roofline']['avg_launch_ms'],2))" $OUT/c2s_gb_$TAG.json
for J in 0 1 3 7; do echo "leaf j=$J"; NUTS_TICK_J=$J python tools/gb_ticks.py 80 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); print('   rows    cycles/100:', j['phases_us'], 'total', j['top_to_end_us']); print('   control cycles/100:', j['control'], 'total', j['control_total'])"; done
} 2>&1 | tee $OUT/gb_round_$TAG.txt
