#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for f in 1 0; do echo "FLAGS=$f"; DPR_NJP_FLAGS=$f timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tail -1; done
echo "30k post2=0 (old kernels everywhere)"; DPR_NJP_POST2=0 timeout -k 10 300 python profiles/nj_big.py 30000 10000 2 2>&1 | tail -1
