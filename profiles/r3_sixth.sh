#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
step() {
    local lim=$1 name=$2; shift 2
    echo "=== $name" | tee -a $OUT/steps6.log
    local t0=$(date +%s)
    timeout -k 10 $lim "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "rc=$rc wall=$(( $(date +%s) - t0 ))s" | tee -a $OUT/steps6.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name killed at its limit: stopping" | tee -a $OUT/steps6.log; exit 1; fi
}
step 300 tests_exact python -m pytest tests/test_gpu_exact.py -m gpu -x -q
tail -15 $OUT/tests_exact.out
step 200 exact_10k python profiles/exact_bench.py 10000 2000
cat $OUT/exact_10k.out
step 300 exact_30k python profiles/exact_bench.py 30000 2000
cat $OUT/exact_30k.out
step 200 nj_worstcase4 python profiles/nj_worstcase.py 30000 10000 const,ints,random
cat $OUT/nj_worstcase4.out
for g in 2048 1024 512; do
  step 100 vw8_grid$g env DPR_NJS_GRID=$g python profiles/njs_vworld_stats.py 30000 10000 256 8 2
  cat $OUT/vw8_grid$g.out
done
step 600 tests_all python -m pytest tests -m gpu -x -q
tail -4 $OUT/tests_all.out
