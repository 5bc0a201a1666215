"""Phase clocks of place_update_kernel (DPR_PLACE_CLOCKS=1 overwrites the trace with wall_clock64 deltas, 10 ns units)."""
import os, sys
os.environ["DPR_PLACE_CLOCKS"] = "1"
sys.path.insert(0, ".")
import numpy as np
import bench, dipper_amd
from dipper_amd import capi
n, L = 20000, 2000
seqs = bench.make_input(n, L, 1)
d = dipper_amd.Dipper(0)
d.set_msa(capi.pack4_many(seqs), L)
st = d.place_run(capi.SRC_MSA, n, dist_type=2)
t = st["trace"][100:]
print("reduce %.2f us  split %.2f us  bfs %.2f us" % tuple(t.mean(axis=0) / 100.0))
print("bfs quantiles (us):", np.quantile(t[:, 2], [0.1, 0.5, 0.9, 0.99]) / 100.0)
