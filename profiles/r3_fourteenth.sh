#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for g in 256 384 512 512; do echo "DPR_NJP_GRID=$g"; DPR_NJP_GRID=$g timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tail -1 | cut -c1-120; done
