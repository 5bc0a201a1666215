import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, dipper_amd
from dipper_amd import capi
from tests import _util
rng = np.random.default_rng(3)
seqs = _util.synth_alignment(rng, 3000, 700, mean_bl=2e-3, lo=2e-4, hi=2e-2)
d = dipper_amd.Dipper(0); d.set_nj_mode(0); d.set_msa(capi.pack4_many(seqs), 700)
for dt in (1, 2, 3, 4, 5, 6):
    d.dist_matrix(capi.SRC_MSA, dt); M = d.matrix()
    bits = M.view(np.uint64)
    print("msa dist type", dt, "bits symmetric:", np.array_equal(bits, bits.T), "NaNs:", int(np.isnan(M).sum()), "equal_nan:", np.array_equal(M, M.T, equal_nan=True))
d.close()
