"""Mash pair-distance kernel alone: sketches of n unaligned reads (seeded indels, tests/_util.synth_reads) and the
whole lower triangle through mash_dist_lookup_kernel (few launches: safe under rocprofv3 --pmc).
usage: python profiles/mash_pairs.py [n] [L] [mean_bl] [identical]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
mbl = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-5
ident = len(sys.argv) > 4 and sys.argv[4] == "identical"
rng = np.random.default_rng(1)
seqs = _util.synth_reads(rng, n, L, mean_bl=mbl, lo=mbl / 10, hi=mbl * 10)
if ident:
    seqs = [seqs[0]] * n
d = dipper_amd.Dipper(0)
d.set_nj_mode(0)
d.set_reads(seqs)
t0 = time.perf_counter()
d.sketch(15, 1000, fetch=False)
ts = time.perf_counter() - t0
for rep in range(2):
    t0 = time.perf_counter()
    d.dist_matrix(capi.SRC_MASH, 0, 15)
    dt = time.perf_counter() - t0
    print(f"n={n} L={L} bl={mbl} {'identical ' if ident else ''}: sketch {ts*1e3:.0f} ms; dist_matrix {dt*1e3:.1f} ms (device {d.timing()[0]:.1f} ms) -> {n*(n-1)/2/(d.timing()[0]*1e-3)/1e6:.1f} M pairs/s")
lens = np.array([len(s) for s in seqs])
print("read lengths: min %d median %d max %d" % (lens.min(), int(np.median(lens)), lens.max()))
d.close()
