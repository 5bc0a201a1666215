"""Kernel times of the one-exchange row-sharded NJ loop with 8 VIRTUAL ranks on one GPU (run under rocprofv3 --kernel-trace
--stats): every rank's SCAN streams 1/8 of the rows and its POST does the whole update, as on 8 GPUs -- except that the
pulls of rows x / y stay in local HBM and the kernels of the 8 ranks run one after the other on one stream.
  python profiles/njs_vworld_stats.py [tips] [sites] [iters] [world] [plan: 1 peer | 2 mailbox]"""
import os, sys, subprocess, tempfile, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 256
world = int(sys.argv[4]) if len(sys.argv) > 4 else 8
plan = int(sys.argv[5]) if len(sys.argv) > 5 else 2
tmp = tempfile.mkdtemp(prefix="vw_")
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "1", "--packed4", p4], check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(p4); os.rmdir(tmp)
d = dipper_amd.Dipper(0, virtual_world=world)
d.set_nj_exchange(plan)
d.set_msa(packed, L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
d.nj_run(max_iters=8)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
t0 = time.perf_counter()
res = d.nj_run(max_iters=iters)
wall = time.perf_counter() - t0
_, loop_ms = d.timing()
print(json.dumps({"tips": n, "virtual_ranks": world, "plan": d.nj_exchange_info()["plan"], "iterations": int(res["iters"]),
                  "loop_ms": loop_ms, "us_per_iteration_all_ranks_serialised": loop_ms * 1e3 / res["iters"]}), flush=True)
d.close()
