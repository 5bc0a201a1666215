#!/bin/bash
# second GPU call of round 3: exchange overhead with process ranks on one GPU, tie-heavy rows of the worst-case table,
# PMC rows of the timed pruned-NJ kernels
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
step() {
    local lim=$1 name=$2; shift 2
    echo "=== $name" | tee -a $OUT/steps2.log
    timeout -k 10 $lim "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "rc=$rc" | tee -a $OUT/steps2.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name killed at its limit: stopping" | tee -a $OUT/steps2.log; exit 1; fi
}
step 300 njs_procs_30k python profiles/njs_procs_bench.py 30000 10000 256 1,2,4
cat $OUT/njs_procs_30k.out
step 200 njs_procs_4k python profiles/njs_procs_bench.py 4096 2000 2048 1,2,4
cat $OUT/njs_procs_4k.out
step 300 nj_worstcase_ties python profiles/nj_worstcase.py 30000 10000 const,ints
cat $OUT/nj_worstcase_ties.out
step 200 pmc_sq bash profiles/pmc_njp.sh sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE"
cat $OUT/pmc_sq.out
step 200 pmc_tcc bash profiles/pmc_njp.sh tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
cat $OUT/pmc_tcc.out
step 200 pmc_fetch bash profiles/pmc_njp.sh fetch "FETCH_SIZE"
cat $OUT/pmc_fetch.out
step 200 pmc_write bash profiles/pmc_njp.sh write "WRITE_SIZE"
cat $OUT/pmc_write.out
