"""Closest-list walks of a placement / --add run (what the outliers of the update launches are made of):
    python3 profiles/place_walks.py [backbone tips] [queries] [sites]
--add of `queries` onto a divide-and-conquer backbone of `backbone tips` (aligned input), then the distribution of the slots
each query's walk reached, the update time they predict (64 queue entries per round trip) and the run's timing."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.setrecursionlimit(1000000)
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util, _orc
m = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
n = m + nq
import tempfile, shutil
tmp = tempfile.mkdtemp(prefix="walk_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
inp = _util.gen_synth(tmp, "a", n, L, 1, 1e-3, 1e-4, 1e-2, shuffle=7)
data = np.asarray(inp["packed4"])
shutil.rmtree(tmp, ignore_errors=True)
d = dipper_amd.Dipper(0)
d.set_msa(np.ascontiguousarray(data[:m]), L)
bb = d.dc_run(capi.SRC_MSA, m, max(m // 20, 3), dist_type=2)
names = [f"T{i}" for i in range(n)]
nwk = _util.newick_from_placement(names[:m], bb["head"], bb["e"], bb["nxt"], bb["len"], m)
st, leaf_names = _util.backbone_state(_orc.load(), nwk, n)
order = [int(x[1:]) for x in leaf_names] + list(range(m, n))
d.set_msa(np.ascontiguousarray(data[np.asarray(order)]), L)
t0 = time.perf_counter()
res = d.place_run(capi.SRC_MSA, n, first=m, dist_type=2, state={k: st[k] for k in ("head", "e", "nxt", "belong", "len")})
wall = time.perf_counter() - t0
w = d.place_walks_per_tip(n)[m:]
a = np.where(w < 0, -w - 1, w).astype(np.int64)
dist_ms, tree_ms = d.place_timing()
rounds = (a + 63) // 64
print(json.dumps({"backbone": m, "queries": nq, "sites": L, "add_s": wall, "distance_ms": dist_ms, "tree_ms": tree_ms, "stats": d.place_walks(),
                  "reached_quantiles_50_90_99_999_max": [int(x) for x in np.quantile(a, [0.5, 0.9, 0.99, 0.999, 1.0])],
                  "tips_with_walk_over_64": int((a > 64).sum()), "over_1000": int((a > 1000).sum()), "over_10000": int((a > 10000).sum()),
                  "sum_rounds_of_64": int(rounds.sum()), "sum_rounds_in_walks_over_64": int(rounds[a > 64].sum()),
                  "largest_walks": sorted(a.tolist())[-8:]}))
d.close()
