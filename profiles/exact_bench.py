"""Exact placement mode timing: python profiles/exact_bench.py [tips] [sites] [mean branch length]"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
mean_bl = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=mean_bl, lo=mean_bl / 10, hi=mean_bl * 10)
perm = np.random.default_rng(7).permutation(n)
seqs = [seqs[i] for i in perm]
d = dipper_amd.Dipper(0)
d.set_msa(capi.pack4_many(seqs), L)
t0 = time.perf_counter()
st = d.place_exact_run(capi.SRC_MSA, n, dist_type=2)
t1 = time.perf_counter()
k = d.place_run(capi.SRC_MSA, n, dist_type=2)
t2 = time.perf_counter()
print(json.dumps(dict(tips=n, sites=L, exact_s=t1 - t0, exact_us_per_tip=(t1 - t0) / n * 1e6, max_depth=int(st["dep"][:2 * n - 1].max()),
                      kclosest_s=t2 - t1, kclosest_us_per_tip=(t2 - t1) / n * 1e6)))
