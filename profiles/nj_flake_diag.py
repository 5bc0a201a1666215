#!/usr/bin/env python3
"""Diagnosis helper: are the two NJ paths reproducible run to run inside one process whose device memory was
used (and freed) by other work before?  Prints the first iteration where two merge logs differ."""
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
# dirty the device memory: a placement run of 20 000 tips (frees everything afterwards)
d = dipper_amd.Dipper(0)
d.set_msa(packed[:20000], L)
d.place_run(capi.SRC_MSA, 20000, dist_type=2)
d.close()

x = torch.full((3 * 1024 * 1024 * 1024 // 8,), float("nan"), dtype=torch.float64, device="cuda")   # 3 GB of NaN
y = torch.full((6 * 1024 * 1024 * 1024 // 8,), 1e-3, dtype=torch.float64, device="cuda")
del x, y
torch.cuda.empty_cache()

def run(mode):
    capi.set_nj_mode(mode)
    d = dipper_amd.Dipper(0)
    try:
        d.set_msa(packed, L)
        d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
        return d.nj_run()
    finally:
        d.close()

def first_diff(a, b):
    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
        ne = np.nonzero(a[key] != b[key])[0]
        if ne.size:
            return key, int(ne[0]), int(ne.size)
    return None

res = [(m, run(m)) for m in (1, 0, 1, 0, 1, 0)]
for i in range(len(res)):
    for j in range(i + 1, len(res)):
        fd = first_diff(res[i][1], res[j][1])
        print(f"run{i}(mode {res[i][0]}) vs run{j}(mode {res[j][0]}):", "identical" if fd is None else fd)
capi.set_nj_mode(1)
