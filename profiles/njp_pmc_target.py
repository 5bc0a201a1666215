#!/usr/bin/env python3
"""Target of the rocprofv3 --pmc passes on the two kernels of the pruned NJ loop (profiles/pmc_njp.sh): 30 000 x 10 000,
the first `iters` iterations, launched eagerly (DPR_NJ_NOGRAPH=1) so that every dispatch is its own counter record.
(Counter collection serialises the dispatches: the whole run of 60 000 launches does not finish in minutes.)"""
import os, sys, subprocess, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n, L = (int(sys.argv[2]) if len(sys.argv) > 2 else 30000), 10000
tmp = tempfile.mkdtemp(prefix="pmc_")
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "1", "--packed4", p4], check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(p4); os.rmdir(tmp)
d = dipper_amd.Dipper(0)
d.set_msa(packed, L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
res = d.nj_run(max_iters=iters)
print("iterations", res["iters"], "units listed", d.prune_stats()[0], flush=True)
d.close()
