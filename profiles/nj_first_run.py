#!/usr/bin/env python3
"""first vs later NJ runs of one process (what a CLI run pays that bench.py's in-process steps do not)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd, bench
from dipper_amd import capi
n, L = 30000, 10000
seqs = bench.make_input(n, L, 1)
packed = capi.pack4_many(seqs)
d = dipper_amd.Dipper(0)
d.reserve_nj(n)
d.set_msa(packed, L)
for r in range(4):
    t0 = time.perf_counter()
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    t1 = time.perf_counter()
    d.nj_run()
    t2 = time.perf_counter()
    print("run %d: dist call %.1f ms, nj call %.1f ms; device dist %.1f nj %.1f ms" % (r, (t1 - t0) * 1e3, (t2 - t1) * 1e3, *d.timing()), flush=True)
