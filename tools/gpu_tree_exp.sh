#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export NUTS_GA_TREE_TIMEOUT_MS=20
timeout 900 python -m pytest tests/test_gpu_benchmark_shapes.py -m gpu -q -x --timeout 600 -rA 2>&1 | grep -v "^PASSED" | grep "Error\|assert\|^E \|passed\|failed\|identical\|worst" | cut -c1-400 | head -20
B="--steps 30 --warmup 60 --cpu-leapfrogs 0"
NUTS_GA_ONES0=0 timeout 300 python bench.py $B > $OUT/n0.json 2> $OUT/n0.err
timeout 300 python bench.py $B > $OUT/n1.json 2> $OUT/n1.err
timeout 400 python bench.py --cpu-leapfrogs 0 > $OUT/n1_full.json 2> $OUT/n1_full.err
NUTS_GA_VARIANT=32 NUTS_GA_TREE_OPTS=1 timeout 400 python bench.py --cpu-leapfrogs 0 > $OUT/n1t_full.json 2> $OUT/n1t_full.err
for f in n0 n1 n1_full n1t_full; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$f", "lf/s %.0f"%d["leapfrog_steps_per_sec"], "ms/step %.3f"%d["ms_per_step"], "tree", d["mean_tree_size"], "avg_launch_ms %.5f"%r["avg_launch_ms"], "frac %.3f"%r["frac"], "leapfrog_frac %.3f"%r["leapfrog_frac"], "ess/s", d.get("ess_per_sec"))
except Exception as e:
    print("$f failed", e); print(open("$OUT/$f.err").read()[-1500:])
PY
done
