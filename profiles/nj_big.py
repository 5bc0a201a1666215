"""NJ hot path at a large size on one GPU: python profiles/nj_big.py [tips] [sites] [runs]
(same generator and branch-length scale as bench.py's sharded leg: mean 2e-5 substitutions per site at 10 000 sites)"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t0 = time.perf_counter()
k = 10000 / L
import subprocess, tempfile
_tmp = tempfile.mkdtemp(prefix="njbig_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
_p4 = os.path.join(_tmp, "a.p4")
subprocess.run([os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "8",
                "--mean-bl", repr(2e-5 * k), "--lo", repr(2e-6 * k), "--hi", repr(2e-4 * k), "--packed4", _p4], check=True)
packed = np.fromfile(_p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(_p4); os.rmdir(_tmp)
print(f"input {n} x {L} in {time.perf_counter()-t0:.1f}s", flush=True)
d = dipper_amd.Dipper(0)
d.set_msa(packed, L)
out = []
for r in range(runs):
    t0 = time.perf_counter()
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    res = d.nj_run()
    wall = time.perf_counter() - t0
    dist_ms, nj_ms = d.timing()
    sc, full = d.prune_stats()
    import hashlib
    h = hashlib.sha256()
    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
        h.update(np.ascontiguousarray(res[key]).tobytes())
    out.append(dict(wall_s=wall, dist_ms=dist_ms, nj_ms=nj_ms, units_scanned=sc, us_per_iteration=nj_ms * 1e3 / (n - 2), digest=h.hexdigest()[:16]))
    print(json.dumps(out[-1]), flush=True)
