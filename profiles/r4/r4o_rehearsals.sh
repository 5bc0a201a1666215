# round 4, call o: rehearsals of bench.py's multi-rank control flow on ONE GPU
#  (a) DPR_BENCH_CHECK=1 with one rank under torch.distributed.run: RCCL communicator with one rank, child legs, sharded legs
#  (b) DPR_BENCH_ONE_GPU=1 with two process ranks on one GPU (gloo; ranks joined through hipIpc windows: mailbox plan only)
O=gpurun_out/r4/o; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time DPR_BENCH_CHECK=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 > $O/bench_check_rehearsal.json 2> $O/bench_check_rehearsal.err ) 2> $O/time_a.txt; echo "rehearsal a rc=$?"; tail -3 $O/time_a.txt
python3 - <<PY
import json
d = json.load(open("$O/bench_check_rehearsal.json"))
print("a:", d["value"], d.get("bench_wall_s"), json.dumps(d.get("nj_iteration_scaling"))[:1200])
print("a sharded_100k keys:", list((d.get("sharded_100k") or {}).keys()))
PY
( time DPR_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_2proc_one_gpu.json 2> $O/bench_2proc_one_gpu.err ) 2> $O/time_b.txt; echo "rehearsal b rc=$?"; tail -3 $O/time_b.txt
python3 - <<PY
import json
d = json.load(open("$O/bench_2proc_one_gpu.json"))
print("b:", d["value"], d["n_gpus"], d.get("bench_wall_s"), json.dumps(d.get("nj_iteration_scaling"))[:1500])
PY
