#!/bin/bash
# usage (GPU box): bash profiles/hip_init_env.sh  -> runtime start-up lap of the CLI (dpr_create) under a few environment settings, back to back
python3 - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
from tests import _util
seqs = _util.synth_alignment(np.random.default_rng(1), 64, 500)
_util.write_fasta("/tmp/small.fa", ["T%d" % (i + 1) for i in range(64)], seqs, width=0)
PY
for envs in "X=1" "HSA_ENABLE_SDMA=0" "ROCR_VISIBLE_DEVICES=0" "HIP_VISIBLE_DEVICES=0" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=1" "HSA_DISABLE_CACHE=0" "X=1"; do
  for i in 1 2 3 4; do
    env $envs DPR_CLI_TIMING=1 ./dipper_amd/bin/dipper -i m -I /tmp/small.fa -O /tmp/o.nwk -m 2 -d 2 2>&1 | grep -E "runtime start-up|stream \+ events" | sed -e 's/.*at //' | tr '\n' ' '
    echo -n "| "
  done
  echo " <- $envs"
done
