#!/usr/bin/env python3
"""Diagnosis helper 2: is the distance matrix itself reproducible when device memory holds garbage?"""
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
rows = list(range(0, 64)) + list(range(14000, 14064)) + list(range(n - 64, n))

def dirty(val):
    xs = [torch.full((2 * 1024 * 1024 * 1024 // 8,), val, dtype=torch.float64, device="cuda") for _ in range(12)]   # 24 GB
    torch.cuda.synchronize()
    del xs
    torch.cuda.empty_cache()

def matrix(mode, val):
    dirty(val)
    capi.set_nj_mode(mode)
    d = dipper_amd.Dipper(0)
    try:
        d.set_msa(packed, L)
        d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
        M = np.stack([d.matrix_row(i) for i in rows])
        r = d.nj_run(max_iters=400)
        return M, r
    finally:
        d.close()

out = [matrix(m, v) for m, v in ((0, 1e-3), (0, float("nan")), (0, 0.0), (1, 1e-3), (1, float("nan")), (1, 0.0))]
for i in range(1, len(out)):
    dm = out[i][0] != out[0][0]
    both_nan = np.isnan(out[i][0]) & np.isnan(out[0][0])
    dm &= ~both_nan
    print(f"matrix run{i} vs run0: {int(dm.sum())} differing entries", (np.argwhere(dm)[:5].tolist() if dm.any() else ""))
    a, b = out[0][1], out[i][1]
    ne = np.nonzero((a["merge_x"] != b["merge_x"]) | (a["merge_y"] != b["merge_y"]) | (a["bl_x"] != b["bl_x"]))[0]
    print(f"   first 400 merges run{i} vs run0:", "identical" if ne.size == 0 else ("first diff at %d" % ne[0]))
capi.set_nj_mode(1)
