"""Phase stamps of ONE iteration of the pruned NJ loop (DPR_NJ_PHASES=<iteration>): for every block of the scan and
the post kernel, thread 0 stamps the 100 MHz wall clock at fixed points; prints, per kernel and role, when the blocks
reach each point relative to the first stamp of the kernel.  usage: python profiles/nj_phases.py [tips sites iteration]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, L, it = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (30000, 10000, 12000)
os.environ["DPR_NJ_PHASES"] = str(it)
os.environ["DPR_NJ_NOGRAPH"] = os.environ.get("DPR_NJ_NOGRAPH", "")
if not os.environ["DPR_NJ_NOGRAPH"]:
    del os.environ["DPR_NJ_NOGRAPH"]
import dipper_amd
from dipper_amd import capi
from tests import _util

seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-5, lo=2e-6, hi=2e-4)
d = dipper_amd.Dipper(0)
d.set_msa(capi.pack4_many(seqs), L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
d.nj_run()
print("nj_ms", d.timing()[1])
buf = np.zeros(4 * 2048 * 8, np.uint64)
L_ = capi.load_library()
L_.dpr_get_nj_phase_stamps.argtypes = [C.c_void_p]
assert L_.dpr_get_nj_phase_stamps(buf.ctypes.data) == 0
buf = buf[:2 * 2048 * 8].reshape(2, 2048, 8).astype(np.int64)
t_scan0 = buf[0][buf[0] > 0].min()
for k, name in ((0, "scan"), (1, "post")):
    b = buf[k]
    used = b[:, 0] > 0
    if not used.any():
        print(name, "no stamps")
        continue
    t0 = b[used][:, 0].min()
    print(f"== {name}: {used.sum()} blocks stamped; first stamp at {10*(t0 - t_scan0)} ns after the scan's first; all times in ns after this kernel's first stamp")
    for j in range(5 if k == 0 else 7):
        col = b[used][:, j]
        ok = col > 0
        if ok.any():
            v = 10 * (col[ok] - t0)
            print(f"   stamp {j}: blocks {ok.sum():5d}  min {v.min():7d}  median {int(np.median(v)):7d}  max {v.max():7d}")
    if k == 0:
        work = used & (b[:, 3] > 0) & (b[:, 1] > 0)
        dur = 10 * (b[work][:, 3] - b[work][:, 1])
        hits, entries = b[work][:, 5], b[work][:, 6]
        order = np.argsort(dur)
        p1 = 10 * (b[work][:, 2] - b[work][:, 1])
        p2 = 10 * (b[work][:, 7] - b[work][:, 2])
        tail = 10 * (b[work][:, 3] - b[work][:, 7])
        print("   unit/new-row blocks by (data arrived -> loop done): dur_ns [pass 1, pass 2 (wave 0), rest], pass-2 row hits, pass-2 entries (waves x units)")
        for o in list(order[:3]) + list(order[len(order)//2-2:len(order)//2+2]) + list(order[-8:]):
            print(f"      {dur[o]:6d} ns  [{p1[o]:6d} {p2[o]:6d} {tail[o]:6d}]  hits {hits[o]:4d}  entries {entries[o]:3d}")
    if k == 1:
        # roles: test blocks stamp 4 and 5, update blocks do not
        test = used & (b[:, 5] > 0)
        upd = used & (b[:, 5] == 0)
        for nm, m in (("test", test), ("update", upd)):
            if m.any():
                e = 10 * (b[m][:, 6] - t0)
                s0 = 10 * (b[m][:, 0] - t0)
                print(f"   {nm} blocks {m.sum()}: start median {int(np.median(s0))} max {s0.max()}; end median {int(np.median(e))} max {e.max()}")
                for j in range(1, 7):
                    ok = b[m][:, j] > 0
                    if ok.any():
                        dj = 10 * (b[m][ok][:, j] - b[m][ok][:, 0])
                        print(f"      since block start -> stamp {j}: median {int(np.median(dj))} max {dj.max()}")
d.close()
