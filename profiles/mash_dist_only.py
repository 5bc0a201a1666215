#!/usr/bin/env python3
"""Mash sketch + full distance matrix only (few kernel launches; safe under rocprofv3 --pmc)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench, dipper_amd
from dipper_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
capi.set_nj_mode(0)
seqs = bench.make_input(n, 10000, 1)
d = dipper_amd.Dipper(0)
d.set_reads(seqs)
d.sketch(15, 1000, fetch=False)
t0 = time.perf_counter()
d.dist_matrix(capi.SRC_MASH, 0, 15)
dt = time.perf_counter() - t0
print(f"n={n}: dist_matrix {dt*1e3:.1f} ms (device {d.timing()[0]:.1f} ms) -> {n*(n-1)/2/dt/1e6:.1f} M pairs/s")
