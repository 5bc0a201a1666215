"""Where the wall time of `dipper -i m ... -m 2 -d 2` goes at 30 000 tips: loader, HIP start-up floor, input, tree, exit.
usage (GPU box): python profiles/cli_breakdown.py [tips sites]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _util

EXE = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
n, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (30000, 10000)


def run(args, env=None, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run(args, capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, r)
    return best


with tempfile.TemporaryDirectory() as tmp:
    seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-5, lo=2e-6, hi=2e-4)
    fa = os.path.join(tmp, "in.fa")
    _util.write_fasta(fa, ["T%d" % (i + 1) for i in range(n)], seqs, width=0)
    small = os.path.join(tmp, "small.fa")
    _util.write_fasta(small, ["T%d" % (i + 1) for i in range(64)], seqs[:64], width=0)
    out = os.path.join(tmp, "o.nwk")
    dt, r = run([EXE, "-h"])
    print("dipper -h (loader only)           : %.0f ms" % (dt * 1e3))
    env = dict(os.environ, LD_DEBUG="statistics")
    _, r = run([EXE, "-h"], env=env, reps=1)
    for ln in r.stderr.splitlines():
        if "total startup time" in ln or "time needed for relocation" in ln or "time needed to load objects" in ln:
            print("   ", ln.strip())
    dt, r = run([EXE, "--dump-fasta", "-I", fa], reps=2)
    print("dipper --dump-fasta 300 MB (read only, no GPU): %.0f ms" % (dt * 1e3))
    dt, r = run([EXE, "-i", "m", "-I", small, "-O", out, "-m", "2", "-d", "2"])
    print("64 tips end to end (HIP start-up + exit floor): %.0f ms" % (dt * 1e3))
    print("    " + " | ".join(l for l in r.stderr.splitlines() if " in:" in l))
    os.environ["DPR_LOG"] = "cli"
    for extra in ([], ["--seed", "-1"]):
        dt, r = run([EXE, "-i", "m", "-I", fa, "-O", out, "-m", "2", "-d", "2"] + extra, reps=4)
        print("%d tips end to end %s: %.0f ms" % (n, extra, dt * 1e3))
        print("    " + " | ".join(l.strip() for l in r.stderr.splitlines() if " in:" in l or "ms" in l))
