#!/usr/bin/env python3
"""Mash sketch + k-closest placement timing: python profiles/place_bench.py [tips] [sites] [kind m|r]"""
import sys, time
sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import bench, dipper_amd
from dipper_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
kind = sys.argv[3] if len(sys.argv) > 3 else "r"
seqs = bench.make_input(n, L, 1)
from profiles import _mgpu
d, rank, world, dist = _mgpu.open_dipper()      # multi-GPU: see profiles/_mgpu.py
t0 = time.perf_counter()
if kind == "r":
    d.set_reads(seqs)
    t1 = time.perf_counter()
    d.sketch(15, 1000, fetch=False)
    t2 = time.perf_counter()
    st = d.place_run(capi.SRC_MASH, n, k=15)
else:
    d.set_msa(capi.pack4_many(seqs), L)
    t1 = t2 = time.perf_counter()
    st = d.place_run(capi.SRC_MSA, n, dist_type=2)
t3 = time.perf_counter()
if rank == 0:
    print("distance / tree part of the run: %.0f / %.0f ms" % d.place_timing())
    print(f"{world} GPU(s) {kind} n={n} L={L}: upload {t1-t0:.2f}s sketch {t2-t1:.3f}s placement {t3-t2:.2f}s ({d.timing()[1]:.0f} ms on device) -> {n/(t3-t1):.0f} tips/s")
_mgpu.finish(dist)
