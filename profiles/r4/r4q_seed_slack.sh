# round 4, call q: what would a seed bound WITHOUT gathers list?  (fused kernel, bound = candidate's old q + slack)
O=gpurun_out/r4/q; mkdir -p $O
python3 - <<PY | tee $O/max_entry.txt
import sys, os, numpy as np, subprocess, tempfile
sys.path.insert(0, os.getcwd())
import dipper_amd
from dipper_amd import capi
from tests import _util
import shutil
tmp = tempfile.mkdtemp(prefix="q_", dir="/dev/shm")
inp = _util.gen_synth(tmp, "a", 30000, 10000, 1, 2e-5, 2e-6, 2e-4)
packed = np.asarray(inp["packed4"]); shutil.rmtree(tmp, ignore_errors=True)
d = dipper_amd.Dipper(0); d.set_msa(packed, 10000); d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
rows = np.stack([d.matrix_row(i) for i in range(0, 30000, 150)])
print("max entry over 200 sampled rows", rows.max(), "mean", rows.mean(), "=> slack 6*2*max/(n-3) at n=30000:", 12 * rows.max() / 29997)
d.close()
PY
for sl in "" 0 3e-8 1e-7 3e-7 1e-6 3e-6 1e-5; do
  if [ -z "$sl" ]; then v=""; else v="DPR_NJP_SEED_SLACK=$sl"; fi
  echo "== slack [$sl]"; env $v python3 profiles/nj_target.py --no-torch --reps 1 2>&1 | grep -o '"nj_ms": [0-9.]*\|"units_listed": [0-9]*\|"digest": "[0-9a-f]*"' | paste - - - | tee -a $O/seed_slack_30k.txt
done
