"""--add timing (BASELINE configs[4] shape, scaled): python profiles/add_bench.py [backbone tips] [queries] [sites] [kind m|r]
Backbone tree = divide-and-conquer tree of the first m tips (built here), imported like Tree::Tree +
initializeDeviceArrays, then the queries are placed with addQuery (dpr_place_run first = m)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.setrecursionlimit(1000000)
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util, _orc
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
kind = sys.argv[4] if len(sys.argv) > 4 else "m"
n = m + nq
import tempfile, shutil
tmp = tempfile.mkdtemp(prefix="addb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
inp = _util.gen_synth(tmp, "a", n, L, 1, 1e-3, 1e-4, 1e-2, reads=(kind == "r"), shuffle=7)      # native generator, seeded (tools/gen_synth.cpp)
data = inp["reads"] if kind == "r" else np.asarray(inp["packed4"])
shutil.rmtree(tmp, ignore_errors=True)
from profiles import _mgpu
d, rank, world, dist = _mgpu.open_dipper()      # multi-GPU: see profiles/_mgpu.py
def load(order):
    if kind == "r":
        d.set_reads_packed(*_util.reads_reorder(data, order)); d.sketch(15, 1000, fetch=False)
    else:
        d.set_msa(np.ascontiguousarray(data[order]), L)
src = capi.SRC_MASH if kind == "r" else capi.SRC_MSA
load(np.arange(m))
t0 = time.perf_counter()
bb = d.dc_run(src, m, max(m // 20, 3), dist_type=2, k=15)
t1 = time.perf_counter()
names = [f"T{i}" for i in range(n)]
nwk = _util.newick_from_placement(names[:m], bb["head"], bb["e"], bb["nxt"], bb["len"], m)
t2 = time.perf_counter()
st, leaf_names = _util.backbone_state(_orc.load(), nwk, n)      # Tree::Tree ids + adjacency (host, Python mirror)
order = [int(x[1:]) for x in leaf_names] + list(range(m, n))      # backbone tips in import order, then the queries
t3 = time.perf_counter()
load(np.asarray(order))
t4 = time.perf_counter()
res = d.place_run(src, n, first=m, dist_type=2, k=15, state={k: st[k] for k in ("head", "e", "nxt", "belong", "len")})
t5 = time.perf_counter()
if rank == 0:
  dist_ms, tree_ms = d.place_timing()
  batches, beside = d.place_policy()
  print(json.dumps(dict(n_gpus=world, kind=kind, backbone=m, queries=nq, sites=L, backbone_tree_s=t1 - t0, newick_host_s=t3 - t1,
                      upload_s=t4 - t3, add_s=t5 - t4, add_device_ms=d.timing()[1], queries_per_s=nq / (t5 - t4), distance_wait_ms=dist_ms, tree_part_ms=tree_ms,
                      batches=batches, batches_beside_tree_kernels=beside, distance_busy_ms=d.place_overlap()[1],
                      policy_env={k: v for k, v in os.environ.items() if k.startswith("DPR_PLACE")})))
  if os.environ.get("DPR_PLACE_CLOCKS"):       # phase clocks of the four-tip update launch (10 ns units): evaluation / rescans + winner / split + BFS
      tr = np.asarray(res["trace"])[m + 8:]
      print("multi-tip update phases, mean us per tip: evaluate %.2f  rescan+winner %.2f  split+bfs %.2f" % tuple(tr.mean(axis=0) / 100.0))
      for k, name in enumerate(("evaluate", "rescan+winner", "split+bfs")):
          print(name, "quantiles 10/50/90/99 % (us):", np.quantile(tr[:, k], [0.1, 0.5, 0.9, 0.99]) / 100.0)
_mgpu.finish(dist)
