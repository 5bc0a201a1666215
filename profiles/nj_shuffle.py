#!/usr/bin/env python3
"""pruned NJ on the bench input in generator order and in a shuffled order (the CLI shuffles its input, src/tree_generation.cu:341-344)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd, bench
from dipper_amd import capi
n, L = 30000, 10000
seqs = bench.make_input(n, L, 1)
for name, order in (("generator order", np.arange(n)), ("shuffled", np.random.default_rng(5).permutation(n))):
    packed = capi.pack4_many([seqs[i] for i in order])
    d = dipper_amd.Dipper(0)
    d.set_msa(packed, L)
    for r in range(2):
        d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
        d.nj_run()
    sc, full = d.prune_stats()
    print("%s: dist %.1f ms, nj %.1f ms, units scanned %d" % (name, *d.timing(), sc), flush=True)
    d.close()
