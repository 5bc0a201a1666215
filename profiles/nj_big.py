#!/usr/bin/env python3
"""NJ hot path at a large size on one GPU: python profiles/nj_big.py [tips] [sites] [runs]
(same generator and branch-length scale as bench.py's sharded leg: mean 2e-5 substitutions per site at 10 000 sites)"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t0 = time.perf_counter()
k = 10000 / L
seqs = _util.synth_alignment(np.random.default_rng(8), n, L, mean_bl=2e-5 * k, lo=2e-6 * k, hi=2e-4 * k)
packed = capi.pack4_many(seqs)
del seqs
print(f"input {n} x {L} in {time.perf_counter()-t0:.1f}s", flush=True)
d = dipper_amd.Dipper(0)
d.set_msa(packed, L)
out = []
for r in range(runs):
    t0 = time.perf_counter()
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    res = d.nj_run()
    wall = time.perf_counter() - t0
    dist_ms, nj_ms = d.timing()
    sc, full = d.prune_stats()
    out.append(dict(wall_s=wall, dist_ms=dist_ms, nj_ms=nj_ms, units_scanned=sc, us_per_iteration=nj_ms * 1e3 / (n - 2)))
    print(json.dumps(out[-1]), flush=True)
