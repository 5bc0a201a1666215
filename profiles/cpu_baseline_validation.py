"""Validates the two extrapolations of bench.py's CPU baselines ONCE, outside the bench's time budget (round 3's verdict, weak #9):
  (a) the RapidNJ-style baseline (oracle/rapidnj_baseline.c) run IN FULL on the bench's 30 000-tip matrix, next to what
      bench.py's rule -- the largest block that fits 20 s, scaled by (N/m)^e with e measured between 3 000 and 6 000 tips -- predicts;
  (b) the oracle NJ (the reference's arithmetic, O(n^3)): a full 10 000-tip run next to the first-k-iterations sample scaled by the
      sum of n^2.
CPU baseline code only (the oracle library is the cpu_baseline checker); the GPU builds the distance matrix.
usage: python3 profiles/cpu_baseline_validation.py [tips 30000] [threads 16]"""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import dipper_amd
from dipper_amd import capi
from tests import _orc, _util
import shutil, tempfile

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(16, bench.host_cores())
L = 10000
tmp = tempfile.mkdtemp(prefix="cpub_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
inp = _util.gen_synth(tmp, "a", n, L, 1, 2e-5, 2e-6, 2e-4)
packed = np.asarray(inp["packed4"])
shutil.rmtree(tmp, ignore_errors=True)
d = dipper_amd.Dipper(0)
d.set_msa(packed, L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
orc = _orc.load()
out = {"tips": n, "threads": threads}

# (a) RapidNJ-style: bench.py's estimate, then the full run
est = bench.cpu_baseline_rapidnj(d, n, threads)
out["rapidnj_style_bench_estimate"] = est
D = bench.gpu_matrix_block(d, n)
t0 = time.perf_counter()
r = orc.rapidnj_run(D, threads=threads)
full = time.perf_counter() - t0
assert r["joins"] == n - 2
out["rapidnj_style_full_run"] = {"seconds": full, "tips_per_s": n / full, "estimate_over_measured": est["value"] / (n / full)}
print(json.dumps(out), flush=True)

# (b) oracle NJ: full run at 10 000 tips vs the bench's sampling rule applied to the same 10 000-tip block
m = min(n, 10000)
Dm = np.ascontiguousarray(D[:m, :m])
del D
t0 = time.perf_counter()
ref = orc.nj_run(np.tril(Dm, -1), threads=threads)
full_m = time.perf_counter() - t0
import ctypes as C
from tests._orc import _p, c_f64p, c_i32p
kmax = 512
mx = np.zeros(kmax, np.int32); my = np.zeros(kmax, np.int32); bx = np.zeros(kmax); by = np.zeros(kmax); last = C.c_double()
def run(k):
    Dc = np.tril(Dm, -1).copy()
    t = time.perf_counter()
    orc.lib.orc_nj_run(_p(Dc, c_f64p), m, m, threads, k, _p(mx, c_i32p), _p(my, c_i32p), _p(bx, c_f64p), _p(by, c_f64p), C.byref(last), None)
    return time.perf_counter() - t
t_init = run(0)
k = 256
tk = run(k)
per_it = (tk - t_init) / k
s_sample = sum(float(m - i) ** 2 for i in range(k))
s_full = sum(float(j) ** 2 for j in range(3, m + 1))
pred = t_init + per_it * k * s_full / s_sample
out["oracle_nj_10k"] = {"tips": m, "full_run_s": full_m, "sample": f"init + first {k} iterations ({tk:.2f} s) scaled by sum(n^2)", "predicted_s": pred,
                        "predicted_over_measured": pred / full_m}
print(json.dumps(out), flush=True)
d.close()
