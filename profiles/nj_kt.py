"""Per-kernel averages (HIP events around every stride-th iteration, eager) of the first `iters` iterations of the pruned NJ:
python profiles/nj_kt.py [tips] [sites] [iters] [stride]   (rocprofv3 is not needed; DPR_NJP_POST2=0/1 selects the post kernel)"""
import json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 10
k = 10000 / L
tmp = tempfile.mkdtemp(prefix="njkt_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "8",
                "--mean-bl", repr(2e-5 * k), "--lo", repr(2e-6 * k), "--hi", repr(2e-4 * k), "--packed4", p4], check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(p4); os.rmdir(tmp)
d = dipper_amd.Dipper(0)
d.set_msa(packed, L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
res = d.nj_run(max_iters=iters)                 # graph replay, untimed kernels: the loop time
_, nj_ms = d.timing()
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
d.set_nj_kernel_timing(stride)
res2 = d.nj_run(max_iters=iters)
kt = d.nj_kernel_timing()
print(json.dumps({"tips": n, "iters": iters, "post2": os.environ.get("DPR_NJP_POST2", "1"), "loop_us_per_iteration_graph": nj_ms * 1e3 / iters, **kt}))
d.close()
