#!/usr/bin/env python3
"""Diagnosis helper 4: two streaming-mode contexts on the same input advanced in lockstep; where do they part?"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # before the library: one HIP runtime per process
import dipper_amd
from dipper_amd import capi
from tests import _util

n, L = 30000, 1000
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-4, lo=2e-5, hi=2e-3)
packed = capi.pack4_many(seqs)
capi.set_nj_mode(mode)
ds = []
for _ in range(2):
    d = dipper_amd.Dipper(0)
    d.set_msa(packed, L)
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    ds.append(d)
u = [d.row_sums() for d in ds]
print("initial row sums equal:", np.array_equal(u[0].view(np.uint64), u[1].view(np.uint64)))
step = 16
for it in range(0, 480, step):
    r = [d.nj_run(max_iters=step) for d in ds]
    same = all(np.array_equal(r[0][k], r[1][k]) for k in ("merge_x", "merge_y", "bl_x", "bl_y"))
    if not same:
        k = [int(np.nonzero((r[0][key] != r[1][key]))[0][0]) if np.any(r[0][key] != r[1][key]) else 10**9 for key in ("merge_x", "merge_y", "bl_x", "bl_y")]
        j = min(k)
        print("iterations %d..%d differ first at +%d:" % (it, it + step, j))
        for a in (0, 1):
            print("   ctx%d: x=%d y=%d blx=%r bly=%r" % (a, r[a]["merge_x"][j], r[a]["merge_y"][j], r[a]["bl_x"][j], r[a]["bl_y"][j]))
        for jj in range(max(0, j - 2), j):
            print("   before: x=%d y=%d" % (r[0]["merge_x"][jj], r[0]["merge_y"][jj]))
        break
else:
    print("480 iterations identical")
capi.set_nj_mode(1)
