"""Repeatability / prefix property of k-closest placement: 100 000, 20 000, 100 000, 20 000 unaligned tips through Mash; the traces of the
first 20 000 tips of all four runs must be equal (the decision for tip i depends on tips < i only).  Written to pin down a race between
the two wavefronts that split an edge (NOTES.md, round 5, item 8): it showed 2 - 39 differing tips between runs before the fix.
  python3 profiles/place_prefix_diag.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n, m, L = 100000, 20000, 3000
seqs = _util.synth_alignment(np.random.default_rng(2), n, L, mean_bl=1e-3, lo=1e-4, hi=1e-2)
d = dipper_amd.Dipper(0)
runs = []
for rep, cnt in enumerate((n, m, n, m)):
    d.set_reads(seqs[:cnt]); d.sketch(15, 1000, fetch=False)
    r = d.place_run(capi.SRC_MASH, cnt, k=15)
    runs.append(r["trace"][2:m].copy())
    print("run", rep, cnt, "done", flush=True)
d.close()
for a in range(4):
    for b in range(a + 1, 4):
        eq = np.array_equal(runs[a], runs[b])
        msg = ""
        if not eq:
            bad = np.flatnonzero((runs[a] != runs[b]).any(axis=1))
            msg = "first differing tip %d (of %d differing): %r vs %r" % (bad[0] + 2, len(bad), runs[a][bad[0]], runs[b][bad[0]])
        print(a, b, eq, msg, flush=True)
