#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
step() {
    local lim=$1 name=$2; shift 2
    echo "=== $name" | tee -a $OUT/steps4.log
    local t0=$(date +%s)
    timeout -k 10 $lim "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "rc=$rc wall=$(( $(date +%s) - t0 ))s" | tee -a $OUT/steps4.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name killed at its limit: stopping" | tee -a $OUT/steps4.log; exit 1; fi
}
step 300 tests_sharded python -m pytest tests/test_gpu_sharded.py tests/test_gpu_multiproc.py -m gpu -x -q
tail -3 $OUT/tests_sharded.out
step 400 bench_rehearsal env DPR_BENCH_CHECK=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 3 --warmup 1
tail -c 3000 $OUT/bench_rehearsal.out
step 500 bench_default python bench.py
tail -c 1500 $OUT/bench_default.out
export DPR_NRF_RECORD=$OUT/nrf_measured.jsonl
rm -f $DPR_NRF_RECORD
step 500 tests_accuracy python -m pytest tests/test_gpu_accuracy.py -m gpu -x -q
tail -5 $OUT/tests_accuracy.out
cat $DPR_NRF_RECORD
step 500 tests_fullsize python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q
tail -5 $OUT/tests_fullsize.out
cd /tmp && export TMPDIR=/tmp
step 150 vworld8_mailbox rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vw8b -o v -- python3 $REPO/profiles/njs_vworld_stats.py 30000 10000 256 8 2
cat $OUT/vworld8_mailbox.out
find $OUT/vw8b -name "*kernel_trace.csv" -delete
