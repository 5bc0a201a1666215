#!/bin/bash
# round 3: njp_post2_kernel -- parity in the forced large shape, then NJ at 100 000 tips with both post kernels
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests/test_gpu_nj.py -x -q -m gpu -k "large_shape" 2>&1 | tail -5 || exit 1
rm -f $OUT/nj_kt_100k.txt
for v in 1 0; do
  DPR_NJP_POST2=$v timeout -k 10 300 python profiles/nj_kt.py 100000 10000 20000 10 2>&1 | tail -1 | tee -a $OUT/nj_kt_100k.txt | cut -c1-330
done
echo "== post2 (default)"
timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | tee $OUT/nj100k_post2.txt | tail -2 || exit 1
