#!/bin/bash
# first GPU call of round 3 (run through gpurun from the repo root): GPU suite, IPC probe, worst-case table, PMC rows of the
# timed NJ kernels.  A step that is KILLED at its time limit ends the script (no further GPU step after a kill).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
step() {   # step <seconds> <name> <cmd...>
    local lim=$1 name=$2; shift 2
    echo "=== $name" | tee -a $OUT/steps.log
    timeout -k 10 $lim "$@" > $OUT/$name.out 2> $OUT/$name.err
    local rc=$?
    echo "rc=$rc" | tee -a $OUT/steps.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name killed at its limit: stopping" | tee -a $OUT/steps.log; exit 1; fi
}
step 400 tests_sharded python -m pytest tests/test_gpu_sharded.py tests/test_gpu_multiproc.py -m gpu -x -q
tail -15 $OUT/tests_sharded.out
step 500 tests python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_sharded.py --deselect tests/test_gpu_multiproc.py
tail -3 $OUT/tests.out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/bin/ipc_probe tools/ipc_probe.hip -ldl 2> $OUT/ipc_build.err
step 150 ipc_probe tools/bin/ipc_probe
cat $OUT/ipc_probe.out
step 500 nj_worstcase python profiles/nj_worstcase.py 30000 10000
cat $OUT/nj_worstcase.out
