"""Divide-and-conquer at the authors' sequence length (configs[3] on one GPU): python3 profiles/protocol_dc.py [tips] [sites]
gen_synth input (gtr+g+i, inherited deletion gaps, mean branch 2e-4, shuffled), dpr_dc_run, one JSON line; the rocprofv3 target of
profiles/r6/kernel_stats_dc_1m_10000_sites.csv."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dipper_amd
from dipper_amd import capi
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
tmp = tempfile.mkdtemp(prefix="pdc_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
t0 = time.perf_counter()
inp = _util.gen_synth(tmp, "a", n, L, 10, 2e-4, 2e-5, 2e-3, shuffle=7, extra=("--model", "gtr+g+i", "--indel-gaps", "--threads", "16"))
tg = time.perf_counter() - t0
d = dipper_amd.Dipper(0)
d.set_msa(inp["packed4"], L)
shutil.rmtree(tmp, ignore_errors=True)
t0 = time.perf_counter()
st = d.dc_run(capi.SRC_MSA, n, n // 20, dist_type=capi.DIST_JC)
wall = time.perf_counter() - t0
print(json.dumps({"tips": n, "sites": L, "generated_s": tg, "seconds": wall, "tips_per_s": n / wall, "stats": {k: (float(v) if isinstance(v, float) else int(v)) for k, v in st["stats"].items()}}))
d.close()
