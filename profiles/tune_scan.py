"""Sweep the tuning knobs of the Q-argmin scan at n = N (run on the GPU box)."""
import itertools
import sys

import numpy as np

sys.path.insert(0, ".")
import dipper_amd
from dipper_amd import capi
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = 2000
seqs = bench.make_input(n, L, 1)
d = dipper_amd.Dipper(0)
d.set_msa(capi.pack4_many(seqs), L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
lib = capi.load_library()
alg = 4.0 * n * n + 4.0 * n
import ctypes as C
for nt, grid in itertools.product((0, 1), (512, 1024, 2048, 4096, 8192)):
    ms = C.c_float()
    nbytes = 4 * n * n
    assert lib.dpr_bw_probe(d.h, nbytes, nt, grid, 10, C.byref(ms)) == 0
    print(f"plain read nt={nt} grid={grid:5d} {ms.value*1e3:7.1f} us  {nbytes/ms.value/1e6:7.1f} GB/s", flush=True)
ref = None
for rg, nt, grid in itertools.product((16, 64), (1,), (2048, 3072, 4096, 6144, 8192)):
    lib.dpr_scan_tune(rg, nt, grid)
    i, j, q, ms = d.argmin_once(reps=10)
    if ref is None:
        ref = (i, j, q)
    assert (i, j, q) == ref
    print(f"rg={rg&127:2d} filt={rg>>7} nt={nt} grid={grid:4d}  {ms*1e3:7.1f} us  {alg/ms/1e6:7.1f} GB/s", flush=True)
