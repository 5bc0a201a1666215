#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout -k 10 300 python profiles/nj_big.py 30000 10000 3 2>&1 | tail -2 | cut -c1-150
timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tail -1 | cut -c1-150
timeout -k 10 900 python -m pytest tests/test_gpu_nj.py -x -q -m gpu 2>&1 | tail -3
