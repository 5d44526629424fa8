#!/bin/bash
# first GPU contact of the tree kernel: parity vs per-leaf (bounded), then per-leaf 42 vs 32 vs tree benches (deep trees)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export NUTS_GA_TREE_TIMEOUT_MS=20
timeout 600 python -m pytest tests/test_gpu_benchmark_shapes.py -m gpu -q -x --timeout 300 -k "tree_kernel" -rA 2>&1 | tail -40 > $OUT/tree_test.log
B="--steps 30 --warmup 60 --cpu-leapfrogs 0"
true
true
NUTS_GA_VARIANT=32 NUTS_GA_TREE=1 timeout 300 python bench.py $B > $OUT/b32t.json 2> $OUT/b32t.err
NUTS_GA_VARIANT=32 NUTS_GA_TREE=1 timeout 400 python bench.py --cpu-leapfrogs 0 > $OUT/b32t_full.json 2> $OUT/b32t_full.err
cat $OUT/tree_test.log | tail -15
for f in b42 b32 b32t b32t_full; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$f", "lf/s %.0f"%d["leapfrog_steps_per_sec"], "ms/step %.3f"%d["ms_per_step"], "tree", d["mean_tree_size"], "avg_launch_ms %.5f"%r["avg_launch_ms"], "frac %.3f"%r["frac"], "leapfrog_frac %.3f"%r["leapfrog_frac"])
except Exception as e:
    print("$f failed", e); print(open("$OUT/$f.err").read()[-1500:])
PY
done
