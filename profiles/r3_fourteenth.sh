#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for g in 128 192 256 384; do echo "30k DPR_NJP_GRID=$g"; DPR_NJP_GRID=$g timeout -k 10 300 python profiles/nj_big.py 30000 10000 3 2>&1 | tail -2 | cut -c1-110; done
