"""Divide-and-conquer mode timing (BASELINE configs[3]): python profiles/dc_bench.py [tips] [sites] [kind m|r] [backbone|0] [mean branch length]
Synthetic alignment as bench.py (Yule-Harding tree, JC69, no indels); backbone defaults to tips/20 as the CLI."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import bench, dipper_amd
from dipper_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
kind = sys.argv[3] if len(sys.argv) > 3 else "m"
B = int(sys.argv[4]) if len(sys.argv) > 4 and int(sys.argv[4]) > 0 else n // 20
mean_bl = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-3   # bench.py's 2e-5 gives near-clonal tips: clusters larger than the backbone, which the reference rejects
t0 = time.perf_counter()
from tests import _util
seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=mean_bl, lo=mean_bl / 10, hi=mean_bl * 10)
rng = np.random.default_rng(7)
perm = rng.permutation(n)          # the CLI shuffles the input order (src/tree_generation.cu:341-344)
seqs = [seqs[i] for i in perm]
print(f"input {n} x {L} generated in {time.perf_counter()-t0:.1f}s", flush=True)
from profiles import _mgpu
d, rank, world, dist = _mgpu.open_dipper()      # multi-GPU: see profiles/_mgpu.py
t0 = time.perf_counter()
if kind == "r":
    d.set_reads(seqs)
    t1 = time.perf_counter()
    d.sketch(15, 1000, fetch=False)
    t2 = time.perf_counter()
    st = d.dc_run(capi.SRC_MASH, n, B, k=15)
else:
    d.set_msa(capi.pack4_many(seqs), L)
    t1 = t2 = time.perf_counter()
    st = d.dc_run(capi.SRC_MSA, n, B, dist_type=2)
t3 = time.perf_counter()
cl = st["cluster_id"][B:]
sizes = np.bincount(cl)
sizes = sizes[sizes > 0]
out = dict(kind=kind, mean_bl=mean_bl, tips=n, sites=L, backbone=B, upload_s=t1 - t0, sketch_s=t2 - t1, dc_s=t3 - t2,
           tips_per_s=n / (t3 - t1), stats=st["stats"],
           cluster_size_quantiles={q: float(np.quantile(sizes, q)) for q in (0.5, 0.9, 0.99, 1.0)})
out["n_gpus"] = world
if rank == 0:
    print(json.dumps(out), flush=True)
_mgpu.finish(dist)
