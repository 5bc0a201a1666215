#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
DPR_NJ_EPOCH_LOG=1 timeout -k 10 300 python profiles/nj_big.py 30000 10000 2 2>&1 | grep -v "graph capture" | grep "n=24000\|n=19200\|wall_s"
timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tail -2
timeout -k 10 600 python -m pytest tests/test_gpu_nj.py -x -q -m gpu -k "large_shape" 2>&1 | tail -2
