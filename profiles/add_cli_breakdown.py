"""Where the wall time of the `dipper -a -t backbone.nwk` command goes (BASELINE configs[4] shape): the command's own stderr lines
(DPR_LOG=cli adds the host-side phases) for the aligned and the Mash input.
  python3 profiles/add_cli_breakdown.py [backbone 500000] [queries 50000] [kind m|r]"""
import os, subprocess, sys, tempfile, time, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import _util
m = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
kind = sys.argv[3] if len(sys.argv) > 3 else "m"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
tmp = tempfile.mkdtemp(prefix="addcli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    n, L = m + nq, (1000 if kind == "m" else 3000)
    inp = _util.gen_synth(tmp, "a", n, L, 10, 1e-3, 1e-4, 1e-2, fasta=True, reads=(kind == "r"), shuffle=7)
    buf = np.memmap(inp["fasta"], dtype=np.uint8, mode="r")
    cut = int(np.flatnonzero(buf == ord(">"))[m])
    bb = os.path.join(tmp, "bb.fa")
    open(bb, "wb").write(buf[:cut].tobytes())
    del buf
    fmt = ["-i", kind] + (["-d", "2"] if kind == "m" else [])
    env = dict(os.environ, DPR_LOG="cli", DPR_HOST_THREADS="16")
    r = subprocess.run([EXE] + fmt + ["-m", "3", "-I", bb, "-O", os.path.join(tmp, "bb.nwk")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-400:]
    for rep in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([EXE] + fmt + ["-a", "-t", os.path.join(tmp, "bb.nwk"), "-I", inp["fasta"], "-O", os.path.join(tmp, "out.nwk")], capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-400:]
        print("== run %d: wall %.3f s" % (rep, wall))
        print(r.stderr.strip())
finally:
    shutil.rmtree(tmp, ignore_errors=True)
