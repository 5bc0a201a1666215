"""Gaps between consecutive tree kernels of a placement run, with and without a distance kernel running beside them.
  rocprofv3 --kernel-trace --output-format csv -d <dir> -o p -- python3 profiles/place_bench.py 100000 3000 r ; python3 profiles/place_gaps.py <dir>
Prints, per class (alone / beside a mash_dist_index_kernel): kernels, mean duration per kernel name, mean gap to the next tree kernel."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
name = lambda r: r["Kernel_Name"]
tree = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r)[:34]) for r in rows if "place_tip" in name(r) or "place_update" in name(r)]
mash = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "mash_dist_index" in name(r))
tree.sort()
ms = np.array([m[0] for m in mash]); me = np.array([m[1] for m in mash])
def beside(t):
    k = np.searchsorted(ms, t, side="right") - 1
    return k >= 0 and t < me[k]
st = np.array([t[0] for t in tree]); en = np.array([t[1] for t in tree])
gap = st[1:] - en[:-1]
b = np.array([beside(t) for t in en[:-1]])
for cls, sel in (("alone", ~b), ("beside", b)):
    if sel.sum() == 0: continue
    print(f"{cls}: {sel.sum()} tree kernels, mean gap to the next {gap[sel].mean()/1e3:.2f} us (median {np.median(gap[sel])/1e3:.2f}, p90 {np.quantile(gap[sel],0.9)/1e3:.2f}); total gap {gap[sel].sum()/1e6:.0f} ms")
    for nm in sorted(set(t[2] for t in tree)):
        s2 = sel & np.array([t[2] == nm for t in tree[:-1]])
        if s2.sum(): print(f"   {nm}: {s2.sum()} launches, mean duration {(en[:-1][s2]-st[:-1][s2]).mean()/1e3:.2f} us, gap after it {gap[s2].mean()/1e3:.2f} us")
