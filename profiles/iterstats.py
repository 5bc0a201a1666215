"""Units listed per iteration over a pruned NJ run (DPR_NJ_ITERSTATS=1): python profiles/iterstats.py [tips] [sites]"""
import ctypes as C, os, sys
import numpy as np
os.environ["DPR_NJ_ITERSTATS"] = "1"
os.environ["DPR_NJ_EPOCH_LOG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-5 * 10000 / L, lo=2e-6 * 10000 / L, hi=2e-4 * 10000 / L)
d = dipper_amd.Dipper(0)
d.set_msa(capi.pack4_many(seqs), L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
r = d.nj_run()
print("nj ms", d.timing()[1], "units", d.prune_stats())
out = np.zeros(2 * (n - 2), dtype=np.uint64)
lib = capi.load_library()
lib.dpr_get_iterstats.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.dpr_get_iterstats(d.h, out.ctypes.data, n - 2) == 0
s = out.reshape(-1, 2).astype(np.int64)
edges = sorted(set([0, 10, 100, 1000] + list(range(0, n - 2, max(1, (n - 2) // 25))) + [n - 2]))
for a, b in zip(edges[:-1], edges[1:]):
    if b <= a:
        continue
    print("it %6d..%6d (n = %6d): units/iter mean %7.0f p50 %7.0f p90 %7.0f max %8d | most per block mean %.1f" % (
        a, b, n - a, s[a:b, 0].mean(), np.percentile(s[a:b, 0], 50), np.percentile(s[a:b, 0], 90), s[a:b, 0].max(), s[a:b, 1].mean()))
