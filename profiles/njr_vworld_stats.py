"""Kernel times of the ROW-SHARDED PRUNED NJ loop (njr.hip) with VIRTUAL ranks on one GPU (run under rocprofv3 --kernel-trace
--stats through profiles/prof.sh): every rank scans the listed units of its own rows, extracts its slices of the winner's two
columns and runs the whole update + its own unit tests, as on G GPUs -- except that the exchange stays in local HBM and the
kernels of the ranks run one after the other on one stream.  Average duration per kernel = per-rank time of that launch.
  python3 profiles/njr_vworld_stats.py [tips] [sites] [world] [plan: 1 collective | 2 mailbox] [iters, -1 = all]"""
import os, sys, subprocess, tempfile, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
if "--no-torch" not in sys.argv:
    import torch  # noqa: F401  (rocprofv3 --kernel-trace and hipGraph: see DESIGN.md section 6; also what bench.py runs on)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
import dipper_amd
from dipper_amd import capi
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
n = int(args[0]) if len(args) > 0 else 30000
L = int(args[1]) if len(args) > 1 else 10000
world = int(args[2]) if len(args) > 2 else 8
plan = int(args[3]) if len(args) > 3 else 2
iters = int(args[4]) if len(args) > 4 else -1
tmp = tempfile.mkdtemp(prefix="vw_")
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "1", "--model", "gtr+g+i", "--indel-gaps",
                "--mean-bl", repr(2e-5 * 10000 / L), "--lo", repr(2e-6 * 10000 / L), "--hi", repr(2e-4 * 10000 / L), "--packed4", p4], check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(p4); os.rmdir(tmp)
d = dipper_amd.Dipper(0, virtual_world=world) if world > 1 else dipper_amd.Dipper(0)
d.set_nj_mode(1)
if world > 1:
    d.set_nj_multi_plan(3)
    d.set_nj_exchange(plan)
d.set_msa(packed, L)
t0 = time.perf_counter()
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
t1 = time.perf_counter()
res = d.nj_run(max_iters=iters)
wall = time.perf_counter() - t1
_, loop_ms = d.timing()
import hashlib
h = hashlib.sha256()
for k in ("merge_x", "merge_y", "bl_x", "bl_y"):
    h.update(np.ascontiguousarray(res[k]).tobytes())
print(json.dumps({"tips": n, "sites": L, "virtual_ranks": world, "plan": {1: "collective", 2: "mailbox"}.get(plan), "iterations": int(res["iters"]),
                  "build_s": t1 - t0, "loop_ms": loop_ms, "us_per_iteration_all_ranks_serialised": loop_ms * 1e3 / max(int(res["iters"]), 1),
                  "digest": h.hexdigest()[:16]}), flush=True)
d.close()
