#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for bp in 16000 24000 32000; do echo "BIG_P=$bp"; DPR_NJ_BIG_P=$bp timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tail -1 | cut -c1-120; done
echo "graph 64"; DPR_NJ_GRAPH_ITERS=64 timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tail -1 | cut -c1-120
echo "epoch pct 85"; DPR_NJ_EPOCH_PCT=85 timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tail -1 | cut -c1-120
echo "epoch pct 75"; DPR_NJ_EPOCH_PCT=75 timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tail -1 | cut -c1-120
