"""Closest-list walks of a k-closest placement from scratch (one-tip launch pairs below 150 000 tips):
    python3 profiles/place_walks_scratch.py [tips] [sites]"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
import tempfile, shutil
tmp = tempfile.mkdtemp(prefix="walk_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
inp = _util.gen_synth(tmp, "a", n, L, 1, 2e-4, 2e-5, 2e-3, shuffle=7)
data = np.asarray(inp["packed4"])
shutil.rmtree(tmp, ignore_errors=True)
d = dipper_amd.Dipper(0)
d.set_msa(data, L)
t0 = time.perf_counter()
res = d.place_run(capi.SRC_MSA, n, dist_type=2)
wall = time.perf_counter() - t0
w = d.place_walks_per_tip(n)
a = np.where(w < 0, -w - 1, w).astype(np.int64)
dist_ms, tree_ms = d.place_timing()
top = np.argsort(a)[-8:]
print(json.dumps({"tips": n, "sites": L, "seconds": wall, "distance_ms": dist_ms, "tree_ms": tree_ms, "stats": d.place_walks(),
                  "reached_quantiles_50_90_99_999_max": [int(x) for x in np.quantile(a, [0.5, 0.9, 0.99, 0.999, 1.0])],
                  "over_64": int((a > 64).sum()), "over_1000": int((a > 1000).sum()), "over_10000": int((a > 10000).sum()),
                  "largest_walks_(tip, slots)": [(int(t), int(a[t])) for t in top]}))
d.close()
