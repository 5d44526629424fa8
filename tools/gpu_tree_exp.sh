#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_dense_adapt.py -m gpu -q --timeout 300 2>&1 | grep -v "^PASSED" | cut -c1-300 | grep "Error\|assert\|^E \|passed\|failed" | head -30
timeout 600 python tools/fa_bench.py 2048 20 > $OUT/fa_bench.json 2> $OUT/fa_bench.err
python -c "
import json; [print({k:(round(v,2) if isinstance(v,float) else v) for k,v in r.items()}) for r in json.load(open('$OUT/fa_bench.json'))]"
