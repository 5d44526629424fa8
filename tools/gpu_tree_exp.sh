#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_benchmark_shapes.py -m gpu -q -x --timeout 600 -rA -k "cross_doubling" 2>&1 | grep -v "^PASSED" | grep "Error\|assert\|^E \|passed\|failed" | cut -c1-600 | head -20
