#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
t0=$(date +%s)
DPR_BENCH_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29677 bench.py --gpus 2 --steps 2 --warmup 1 --sharded-tips 50000 --deadline-s 330 2>&1 >$OUT/bench_2proc.out | tee $OUT/bench_2proc.err | grep --line-buffered -v "^W1\|amdgpu.ids\|c10d" || true
echo "rc=$? wall=$(( $(date +%s) - t0 ))s"
tail -c 600 $OUT/bench_2proc.err
python3 - <<PY
import json
txt = open("$OUT/bench_2proc.out").read().strip().splitlines()
d = json.loads(txt[-1])
print("value", d["value"], "n_gpus", d["n_gpus"], "ms/step", d["ms_per_step"], "wall", d.get("bench_wall_s"))
print("multi_gpu_check", d.get("multi_gpu_check"), "staging", d.get("input_staging"))
print("nj_scaling", json.dumps(d.get("nj_scaling"))[:2500])
print("sharded", json.dumps(d.get("sharded_100k"))[:2500])
PY
