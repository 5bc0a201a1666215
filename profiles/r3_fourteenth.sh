#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r3
mkdir -p $OUT
cd $REPO
python3 - <<'PY'
import numpy as np, dipper_amd
from dipper_amd import capi
from tests import _util
rng = np.random.default_rng(3)
seqs = _util.synth_alignment(rng, 3000, 700, mean_bl=2e-3, lo=2e-4, hi=2e-2)
d = dipper_amd.Dipper(0); d.set_nj_mode(0); d.set_msa(capi.pack4_many(seqs), 700)
for dt in (1, 2, 3, 4, 5, 6):
    d.dist_matrix(capi.SRC_MSA, dt); M = d.matrix(); print("msa dist type", dt, "symmetric bit for bit:", np.array_equal(M, M.T))
reads = _util.synth_reads(rng, 2000, 1500, mean_bl=2e-3, lo=2e-4, hi=2e-2)
d.set_reads(reads); d.sketch(15, 1000, fetch=False); d.dist_matrix(capi.SRC_MASH); M = d.matrix(); print("mash symmetric:", np.array_equal(M, M.T))
d.close()
PY
for v in 1 0; do
  echo "DPR_NJP_PERMUTE=$v"
  DPR_NJP_PERMUTE=$v DPR_NJ_EPOCH_LOG=1 timeout -k 10 300 python profiles/nj_big.py 100000 10000 2 2>&1 | grep -v "graph capture" | grep "n=80000\|n=64000\|wall_s" | cut -c1-130
done
timeout -k 10 300 python profiles/nj_big.py 30000 10000 3 2>&1 | tail -2 | cut -c1-130
timeout -k 10 900 python -m pytest tests/test_gpu_nj.py -x -q -m gpu -k "not large_shape" 2>&1 | tail -2
