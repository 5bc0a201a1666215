import ctypes as C, os, sys
import numpy as np
os.environ["DPR_NJ_ITERSTATS"] = "1"
sys.path.insert(0, ".")
import dipper_amd, bench
from dipper_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
seqs = bench.make_input(n, 10000, 1)
d = dipper_amd.Dipper(0)
d.set_msa(capi.pack4_many(seqs), 10000)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
r = d.nj_run()
print("nj ms", d.timing()[1])
out = np.zeros(2 * (n - 2), dtype=np.uint64)
lib = capi.load_library()
lib.dpr_get_iterstats.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.dpr_get_iterstats(d.h, out.ctypes.data, n - 2) == 0
s = out.reshape(-1, 2).astype(np.int64)
for a, b in [(0, 10), (10, 100), (100, 1000), (1000, 5000), (5000, 10000), (10000, 20000), (20000, 25000), (25000, 28000), (28000, n - 2)]:
    print(a, b, "units/iter mean %.0f p50 %.0f p90 %.0f | max per block mean %.1f p90 %.0f" % (
        s[a:b, 0].mean(), np.percentile(s[a:b, 0], 50), np.percentile(s[a:b, 0], 90), s[a:b, 1].mean(), np.percentile(s[a:b, 1], 90)))
