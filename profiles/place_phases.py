"""Phase clocks of place_update_kernel (DPR_PLACE_CLOCKS=1 overwrites the trace with wall_clock64 deltas, 10 ns units):
reduce = block partials -> winner, split = the edge split (wavefront 0's share), bfs = the closest-list update from round 2 on.
  python3 profiles/place_phases.py [tips 100000] [sites 3000]"""
import os, sys, shutil, tempfile
os.environ["DPR_PLACE_CLOCKS"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or ".")
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
tmp = tempfile.mkdtemp(prefix="plp_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
inp = _util.gen_synth(tmp, "a", n, L, 9, 2e-5, 2e-6, 2e-4)
data = np.asarray(inp["packed4"])
shutil.rmtree(tmp, ignore_errors=True)
d = dipper_amd.Dipper(0)
d.set_msa(data, L)
st = d.place_run(capi.SRC_MSA, n, dist_type=2)
t = st["trace"][100:]
print("tips %d: reduce %.2f us  split %.2f us  bfs %.2f us" % ((n,) + tuple(t.mean(axis=0) / 100.0)))
for k, name in enumerate(("reduce", "split", "bfs")):
    print(name, "quantiles 10/50/90/99 % (us):", np.quantile(t[:, k], [0.1, 0.5, 0.9, 0.99]) / 100.0)
for lo in range(0, n - 100, max(1, (n - 100) // 5)):
    seg = t[lo:lo + (n - 100) // 5]
    print("tips %6d..: reduce %.2f split %.2f bfs %.2f" % ((lo + 100,) + tuple(seg.mean(axis=0) / 100.0)))
