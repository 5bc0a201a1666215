#!/bin/bash
# round 3: the 100 000-tip records with the final njp_post2_kernel (and the fused kernel beside it, same box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
timeout -k 10 300 python profiles/nj_phases2.py 100000 10000 5000 > $OUT/nj_phases2_100k.txt 2>&1; tail -4 $OUT/nj_phases2_100k.txt
timeout -k 10 300 python profiles/nj_big.py 100000 10000 3 > $OUT/nj100k_post2.txt 2>&1; tail -3 $OUT/nj100k_post2.txt
DPR_NJP_POST2=0 timeout -k 10 300 python profiles/nj_big.py 100000 10000 3 > $OUT/nj100k_fused.txt 2>&1; tail -3 $OUT/nj100k_fused.txt
rm -f $OUT/nj_kt_100k.txt
for v in 1 0; do DPR_NJP_POST2=$v timeout -k 10 300 python profiles/nj_kt.py 100000 10000 20000 10 2>&1 | tail -1 >> $OUT/nj_kt_100k.txt; done
cut -c1-330 $OUT/nj_kt_100k.txt
