"""NJ loop time per segment of a run (graph replay): python profiles/nj_segments.py [tips] [sites] [segment]"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
seg = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
k = 10000 / L
tmp = tempfile.mkdtemp(prefix="njseg_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "8",
                "--mean-bl", repr(2e-5 * k), "--lo", repr(2e-6 * k), "--hi", repr(2e-4 * k), "--packed4", p4], check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(p4); os.rmdir(tmp)
d = dipper_amd.Dipper(0)
d.set_msa(packed, L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
out, done = [], 0
while done < n - 2:
    res = d.nj_run(max_iters=seg)
    _, ms = d.timing()
    sc, _ = d.prune_stats()
    out.append((done, round(ms * 1e3 / max(res["iters"], 1), 2), int(sc)))
    done += res["iters"]
    if res["iters"] == 0:
        break
print(json.dumps({"post2": os.environ.get("DPR_NJP_POST2", "1"), "us_per_iteration_by_segment_start": out, "total_ms": sum(o[1] for o in out) * seg / 1e3}))
d.close()
