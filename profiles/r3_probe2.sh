cd ${GRAFT_REPO_ROOT:-/root/repo}
run() {   # variant n world
  D=$(mktemp -d); echo "== $1 n=$2 world=$3 (bytes per rank: $(( ($2 / $3) * $2 * 8 )))"
  for r in $(seq 0 $(($3-1))); do RANK=$r WORLD_SIZE=$3 PROBE_VARIANT=$1 PROBE_N=$2 PROBE_DIR=$D DPR_IPC_ANY_SIZE=$4 timeout -k 5 40 python profiles/ipc_torch_probe.py 2>&1 | grep "^\[r\|Error\|error" | tr '\n' ' ' & done; wait; echo; rm -rf $D
}
run none 30000 2 ""
run none 44000 4 ""
run torch 30000 2 ""
run torch 22000 2 ""
