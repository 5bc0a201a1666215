"""Inputs of the --add profile at the authors' sequence length: python3 profiles/protocol_add_setup.py <dir> [backbone] [queries] [sites]
writes <dir>/all.fa (backbone + queries, shuffled) and <dir>/bb.nwk (the command's own divide-and-conquer tree of the first
`backbone` records); the profiled step is then `dipper -i m -d 2 -a -t <dir>/bb.nwk -I <dir>/all.fa -O <dir>/out.nwk`."""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests import _util
out = sys.argv[1]
m = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
os.makedirs(out, exist_ok=True)
sc = 1000.0 / L
inp = _util.gen_synth(out, "all", m + nq, L, 11, 1e-3 * sc, 1e-4 * sc, 1e-2 * sc, shuffle=7, fasta=True, extra=("--model", "gtr+g+i", "--indel-gaps", "--threads", "16"))
os.unlink(os.path.join(out, "all.p4"))
buf = np.memmap(inp["fasta"], dtype=np.uint8, mode="r")
starts = np.flatnonzero(buf == ord(">"))
with open(os.path.join(out, "bb.fa"), "wb") as f:
    f.write(buf[:int(starts[m])].tobytes())
del buf
exe = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
r = subprocess.run([exe, "-i", "m", "-d", "2", "-m", "3", "-I", os.path.join(out, "bb.fa"), "-O", os.path.join(out, "bb.nwk")], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-500:]
os.unlink(os.path.join(out, "bb.fa"))
print("ready:", inp["fasta"], os.path.getsize(inp["fasta"]), "bytes")
