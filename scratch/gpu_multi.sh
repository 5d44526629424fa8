#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "large_n" 2>&1 | tail -15
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 30 --warmup 40 --cpu-leapfrogs 0 --backend gloo --share-gpu --rows-per-group 400 2>&1 | tail -5
