"""pruned vs streaming NJ on one input: where do the merge logs differ?  python profiles/nj_diff_diag.py [tips sites mean_bl]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (30000, 1000)
mbl = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-4
import subprocess, tempfile
_tmp = tempfile.mkdtemp(prefix="njdiff_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
_p4 = os.path.join(_tmp, "a.p4")
subprocess.run([os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "1",
                "--mean-bl", repr(mbl), "--lo", repr(mbl / 10), "--hi", repr(mbl * 10), "--packed4", _p4], check=True)      # native generator (tools/gen_synth.cpp)
packed = np.fromfile(_p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(_p4); os.rmdir(_tmp)
res = {}
dirty = os.environ.get("DIAG_DIRTY")
for mode in (1, 0, 1):
    if dirty is not None:
        from tests.conftest import dirty_device_memory
        capi.load_library()
        dirty_device_memory(8 << 30, int(dirty) & 0xFF)
    d = dipper_amd.Dipper(0)
    d.set_nj_mode(mode)
    d.set_msa(packed, L)
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    r = d.nj_run()
    print("mode", "pruned" if mode else "streaming", "nj_ms", round(d.timing()[1], 1), flush=True)
    d.close()
    if mode in res:
        print("pruned run twice: identical", all(np.array_equal(res[mode][k], r[k]) for k in ("merge_x", "merge_y", "bl_x", "bl_y")))
    res[mode] = r
a, b = res[1], res[0]
for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
    bad = np.nonzero(a[key] != b[key])[0]
    print(key, "differences:", len(bad), "first", bad[:8])
    for i in bad[:5]:
        print("   it", i, "pruned", repr(a[key][i]), "stream", repr(b[key][i]), "x,y", a["merge_x"][i], a["merge_y"][i])
