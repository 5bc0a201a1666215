cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3
D=$(mktemp -d)
for r in 0 1; do RANK=$r WORLD_SIZE=2 PROBE_VARIANT=torch PROBE_N=30000 PROBE_DIR=$D timeout -k 5 40 python -X faulthandler profiles/ipc_torch_probe.py > gpurun_out/r3/probe3_r$r.txt 2>&1 & done; wait
tail -n 15 gpurun_out/r3/probe3_r0.txt
