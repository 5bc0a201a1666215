"""End-to-end `dipper` CLI timing (FASTA -> Newick) on the GPU box: python profiles/e2e_cli.py [tips] [sites]"""
import os, subprocess, sys, time
sys.path.insert(0, ".")
import bench
from tests import _util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
seqs = bench.make_input(n, L, 1)
fa = "/tmp/e2e_%d.fa" % n
_util.write_fasta(fa, ["T%d" % (i + 1) for i in range(n)], seqs, width=0)
print("fasta bytes", os.path.getsize(fa))
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run(["dipper_amd/bin/dipper", "-i", "m", "-I", fa, "-O", "/tmp/e2e.nwk", "-m", "2", "-d", "2"], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print("rep", rep, "rc", r.returncode, "wall %.3f s -> %.0f tips/s" % (dt, n / dt))
    print(r.stderr.strip().replace("\n", " | "))
print("newick bytes", os.path.getsize("/tmp/e2e.nwk"))
