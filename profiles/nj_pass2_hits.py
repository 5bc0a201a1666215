import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tests") else ".")
os.environ["DPR_NJ_PHASES"] = "-2"
import dipper_amd
from dipper_amd import capi
from tests import _util
n, L = 30000, 10000
seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-5, lo=2e-6, hi=2e-4)
d = dipper_amd.Dipper(0)
d.set_msa(capi.pack4_many(seqs), L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
d.nj_run()
buf = np.zeros(4 * 2048 * 8, np.uint64)
L_ = capi.load_library()
L_.dpr_get_nj_phase_stamps.argtypes = [C.c_void_p]
assert L_.dpr_get_nj_phase_stamps(buf.ctypes.data) == 0
print("nj ms", d.timing()[1], "pass-2 row hits in the whole run:", int(buf[5]), "wave-unit entries:", int(buf[6]), "per iteration: %.1f hits, %.1f entries" % (buf[5] / (n - 2), buf[6] / (n - 2)))
mx = buf[32768:32768 + n - 2]
units, subs = (mx >> np.uint64(32)).astype(np.int64), (mx & np.uint64(0xffffffff)).astype(np.int64)
it = np.arange(n - 2)
print("largest number of units one test block lists, per iteration: mean %.2f  p50 %d  p90 %d  p99 %d  max %d" % (units.mean(), *np.percentile(units, [50, 90, 99]).astype(int), units.max()))
print("  ... sub-units of that block: mean %.2f  p50 %d  p90 %d  p99 %d  max %d" % (subs.mean(), *np.percentile(subs, [50, 90, 99]).astype(int), subs.max()))
print("sub-units listed per iteration (all blocks): %.1f" % (buf[7] / (n - 2)))
for a, b in ((0, 5000), (5000, 15000), (15000, 25000), (25000, n - 2)):
    print("  iterations %d-%d: mean max units/block %.2f" % (a, b, units[a:b].mean()))
