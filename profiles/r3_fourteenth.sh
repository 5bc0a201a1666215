#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for bp in 1 8000 40000; do echo "30k tips BIG_P=$bp"; DPR_NJ_BIG_P=$bp timeout -k 10 300 python profiles/nj_big.py 30000 10000 3 2>&1 | tail -2; done
echo "100k BIG_P=1"; DPR_NJ_BIG_P=1 timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tail -1
