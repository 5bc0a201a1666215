#!/bin/bash
# round 3: PMC rows of the post kernels in the LARGE shape (100 000 tips, first 400 iterations, eager): njp_post2_kernel and, with DPR_NJP_POST2=0, the fused kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
: > $OUT/pmc_post_kernels_100k.csv
for v in 1 0; do
  export DPR_NJP_POST2=$v
  timeout -k 10 250 bash profiles/pmc_njp.sh sq_p2_$v "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" 400 100000 2>&1 | grep "^sq_p2" >> $OUT/pmc_post_kernels_100k.csv || exit 1
  timeout -k 10 250 bash profiles/pmc_njp.sh tcc_p2_$v "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" 400 100000 2>&1 | grep "^tcc_p2" >> $OUT/pmc_post_kernels_100k.csv || exit 1
  timeout -k 10 250 bash profiles/pmc_njp.sh fetch_p2_$v "FETCH_SIZE" 400 100000 2>&1 | grep "^fetch_p2" >> $OUT/pmc_post_kernels_100k.csv || exit 1
done
grep "post" $OUT/pmc_post_kernels_100k.csv
