#!/usr/bin/env python3
"""Diagnosis helper 3: full distance matrices of two builds (different garbage in freed device memory) compared."""
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

def dirty(val):
    xs = [torch.full((2 * 1024 * 1024 * 1024 // 8,), val, dtype=torch.float64, device="cuda") for _ in range(12)]
    torch.cuda.synchronize()
    del xs
    torch.cuda.empty_cache()

def build(val):
    dirty(val)
    capi.set_nj_mode(0)
    d = dipper_amd.Dipper(0)
    d.set_msa(packed, L)
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    return d

d0 = build(1e-3)
rows0 = [d0.matrix_row(i).view(np.uint64).copy() for i in range(n)]
d0.close()
d1 = build(0.0)
bad = []
for i in range(n):
    r = d1.matrix_row(i).view(np.uint64)
    ne = np.nonzero(r != rows0[i])[0]
    if ne.size:
        bad.append((i, ne.size, ne[:4].tolist(), rows0[i][ne[:2]].view(np.float64).tolist(), r[ne[:2]].view(np.float64).tolist()))
d1.close()
print("rows with differences:", len(bad))
for b in bad[:20]:
    print(b)
capi.set_nj_mode(1)
