"""k-closest placement from scratch (BASELINE configs[2] shape): python3 profiles/place_bench.py [tips] [sites] [kind m|r] [mean branch]
Reads (kind r: Mash sketches) or an alignment (kind m) from tools/bin/gen_synth (bench.py's protocol: mean branch 2e-5 at 10 000 sites,
scaled to the length); prints one JSON line: seconds, distance / tree part, batches and how many were produced beside the tree kernels."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
kind = sys.argv[3] if len(sys.argv) > 3 else "r"
mean = float(sys.argv[4]) if len(sys.argv) > 4 else 2e-5
tmp = tempfile.mkdtemp(prefix="plb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
inp = _util.gen_synth(tmp, "a", n, L, 9, mean, mean / 10, mean * 10, reads=(kind == "r"))
data = inp["reads"] if kind == "r" else np.asarray(inp["packed4"])
shutil.rmtree(tmp, ignore_errors=True)
d = dipper_amd.Dipper(0)
t0 = time.perf_counter()
if kind == "r":
    d.set_reads_packed(*data)
    d.sketch(15, 1000, fetch=False)
    t1 = time.perf_counter()
    st = d.place_run(capi.SRC_MASH, n, k=15)
else:
    d.set_msa(data, L)
    t1 = time.perf_counter()
    st = d.place_run(capi.SRC_MSA, n, dist_type=2)
t2 = time.perf_counter()
dist_ms, tree_ms = d.place_timing()
batches, beside = d.place_policy()
print(json.dumps({"kind": kind, "tips": n, "sites": L, "sketch_s": t1 - t0, "placement_s": t2 - t1, "device_ms": d.timing()[1], "distance_wait_ms": dist_ms, "tree_part_ms": tree_ms,
                  "batches": batches, "batches_beside_tree_kernels": beside, "distance_busy_ms": d.place_overlap()[1],
                  "policy_env": {k: v for k, v in os.environ.items() if k.startswith("DPR_PLACE")}}))
d.close()
