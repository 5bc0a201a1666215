#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
for v in 1 0; do
  DPR_NJP_POST2=$v timeout -k 10 300 python profiles/nj_kt.py 100000 10000 20000 10 2>&1 | tail -1 | tee -a $OUT/nj_kt_100k.txt
done
