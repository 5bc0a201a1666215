#!/usr/bin/env python3
"""Diagnosis helper 6: distance matrix + row sums after NaN / 7.0 garbage vs after zeros (device synchronised)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # before the library: one HIP runtime per process
import dipper_amd
from dipper_amd import capi
from tests import _util

n, L = 30000, 1000
seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-4, lo=2e-5, hi=2e-3)
packed = capi.pack4_many(seqs)
capi.set_nj_mode(0)

def dirty(val):
    xs = [torch.full((2 * 1024 * 1024 * 1024 // 8,), val, dtype=torch.float64, device="cuda") for _ in range(12)]
    torch.cuda.synchronize()
    del xs
    torch.cuda.empty_cache()

def build(val):
    dirty(val)
    d = dipper_amd.Dipper(0)
    d.set_msa(packed, L)
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    torch.cuda.synchronize()
    return d

d0 = build(0.0)
u0 = d0.row_sums().view(np.uint64).copy()
rows0 = [d0.matrix_row(i).view(np.uint64).copy() for i in range(n)]
d0.close()
for val in (float("nan"), 7.0):
    d1 = build(val)
    u1 = d1.row_sums().view(np.uint64)
    ne = np.nonzero(u1 != u0)[0]
    print("dirt %r: row sums differ in %d entries" % (val, ne.size), ne[:8].tolist(), u1[ne[:3]].view(np.float64).tolist(), u0[ne[:3]].view(np.float64).tolist())
    bad = 0
    for i in range(n):
        r = d1.matrix_row(i).view(np.uint64)
        w = np.nonzero(r != rows0[i])[0]
        if w.size:
            bad += 1
            if bad <= 5:
                print("   row", i, "cols", w[:6].tolist(), r[w[:3]].view(np.float64).tolist(), rows0[i][w[:3]].view(np.float64).tolist())
    print("   matrix rows with differences:", bad)
    d1.close()
capi.set_nj_mode(1)
