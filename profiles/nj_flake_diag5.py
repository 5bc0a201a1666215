#!/usr/bin/env python3
"""Diagnosis helper 5: same input, contexts created after different garbage was left in freed device memory:
row sums, then merge logs of one nj_run(400) call each."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # before the library: one HIP runtime per process
import dipper_amd
from dipper_amd import capi
from tests import _util

n, L = 30000, 1000
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-4, lo=2e-5, hi=2e-3)
packed = capi.pack4_many(seqs)
capi.set_nj_mode(mode)

def dirty(val):
    xs = [torch.full((2 * 1024 * 1024 * 1024 // 8,), val, dtype=torch.float64, device="cuda") for _ in range(12)]
    torch.cuda.synchronize()
    del xs
    torch.cuda.empty_cache()

res = []
for val in (1e-3, float("nan"), 0.0, 1e-3, 7.0, 0.0):
    dirty(val)
    d = dipper_amd.Dipper(0)
    d.set_msa(packed, L)
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    u = d.row_sums().view(np.uint64).copy()
    r = d.nj_run(max_iters=iters)
    u2 = d.row_sums().view(np.uint64).copy()
    d.close()
    res.append((val, u, r, u2))
for i in range(1, len(res)):
    ueq = np.array_equal(res[i][1], res[0][1])
    a, b = res[0][2], res[i][2]
    ne = np.nonzero((a["merge_x"] != b["merge_x"]) | (a["merge_y"] != b["merge_y"]) | (a["bl_x"] != b["bl_x"]) | (a["bl_y"] != b["bl_y"]))[0]
    print("dirt %r vs first: row sums equal %s, merges %s" % (res[i][0], ueq, "identical" if ne.size == 0 else "first diff at %d: (%d,%d) vs (%d,%d), bl %r vs %r" % (ne[0], a["merge_x"][ne[0]], a["merge_y"][ne[0]], b["merge_x"][ne[0]], b["merge_y"][ne[0]], a["bl_x"][ne[0]], b["bl_x"][ne[0]])))
capi.set_nj_mode(1)
