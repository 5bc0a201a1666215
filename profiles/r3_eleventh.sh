#!/bin/bash
# round 3: (a) hipIpcOpenMemHandle under the two HIP runtimes, (b) bench.py rehearsals: one rank under torchrun with RCCL
# (DPR_BENCH_CHECK=1: every multi-GPU leg incl. the child legs) and two process ranks on one GPU (DPR_BENCH_ONE_GPU=1)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
bash profiles/r3_probe2.sh > $OUT/ipc_runtime_probe.txt 2>&1
cat $OUT/ipc_runtime_probe.txt
t0=$(date +%s)
DPR_BENCH_CHECK=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 3 --warmup 1 2>$OUT/bench_check.err >$OUT/bench_check.out
echo "check rc=$? wall=$(( $(date +%s) - t0 ))s"
grep "row_sharded\|error\|Error" $OUT/bench_check.err | cut -c1-700
python3 - <<PY
import json
d = json.loads(open("$OUT/bench_check.out").read().strip().splitlines()[-1])
print("value", d["value"], "wall", d.get("bench_wall_s"))
print("nj_scaling", json.dumps(d.get("nj_scaling", {}).get("row_sharded"))[:1500])
s = d.get("sharded_100k", {})
print("sharded unit", json.dumps(s.get("unit_sharded_plan"))[:400])
print("sharded rows", json.dumps(s.get("nj_scaling", {}).get("row_sharded"))[:1500])
print("dc", json.dumps(s.get("dc_1m"))[:400])
PY
