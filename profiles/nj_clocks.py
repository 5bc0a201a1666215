#!/usr/bin/env python3
"""Phase clocks of the three pruned-NJ kernels (wall_clock64, 10 ns units; thread 0 of block 0, prep also the middle block), averaged over a run."""
import ctypes as C, os, sys
import numpy as np
os.environ["DPR_NJ_ITERSTATS"] = "1"
sys.path.insert(0, ".")
import dipper_amd, bench
from dipper_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
seqs = bench.make_input(n, 10000, 1)
d = dipper_amd.Dipper(0)
d.set_msa(capi.pack4_many(seqs), 10000)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
r = d.nj_run()
print("nj ms", d.timing()[1])
out = np.zeros(2 * (n + 33), dtype=np.uint64)
lib = capi.load_library()
lib.dpr_get_iterstats.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.dpr_get_iterstats(d.h, out.ctypes.data, n + 33) == 0
clk = out[2 * n + 2:].astype(np.float64)
names = ["hop 1 (state, tables, records)", "hop 2 (unit data, gathers)", "tree + division", "reductions + tests", "list append (atomic)", "list store"]
for blk, o in (("block 0", 0), ("middle block", 8)):
    cnt = clk[o]
    if cnt:
        print(blk, "launches %d:" % cnt, "; ".join("%s %.2f us" % (nm, clk[o + 1 + k] / cnt / 100.0) for k, nm in enumerate(names)))
for name, o, names in (("scan block 0", 16, ["hop 1 (state, list)", "hop 2 (unit data)", "passes + reductions", "block winner + record"]),
                       ("post block 0", 32, ["hop 1 (state, slots, records)", "record reduction", "hop 2 (rows px, py)", "update + stores", "tree + chunk sum"])):
    cnt = clk[o]
    if cnt:
        print(name, "launches %d:" % cnt, "; ".join("%s %.2f us" % (nm, clk[o + 1 + k] / cnt / 100.0) for k, nm in enumerate(names)))
