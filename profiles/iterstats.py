"""Units listed per iteration over a pruned NJ run (DPR_NJ_ITERSTATS=1): python profiles/iterstats.py [tips] [sites] [gen_synth options]"""
import ctypes as C, os, sys
import numpy as np
os.environ["DPR_NJ_ITERSTATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess, tempfile
import dipper_amd
from dipper_amd import capi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
extra = sys.argv[3:]          # passed to tools/bin/gen_synth (e.g. --model gtr+g+i --indel-gaps)
tmp = tempfile.mkdtemp(prefix="its_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "1", "--mean-bl", repr(2e-5 * 10000 / L),
                "--lo", repr(2e-6 * 10000 / L), "--hi", repr(2e-4 * 10000 / L), "--packed4", p4] + extra, check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(p4); os.rmdir(tmp)
d = dipper_amd.Dipper(0)
d.set_msa(packed, L)
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
