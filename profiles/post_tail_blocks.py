"""Which blocks end a launch of the fused post kernel?  DPR_NJ_PHASES stamps of one iteration (30 000 tips), per block:
the 20 blocks that end last, with their role, strip, first row group and every stamp.  python3 profiles/post_tail_blocks.py [iteration]"""
import os, sys, subprocess, tempfile, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
it = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
os.environ["DPR_NJ_PHASES"] = str(it)
import numpy as np
import dipper_amd
from dipper_amd import capi
n, L = 30000, 10000
tmp = tempfile.mkdtemp(prefix="ptb_", dir="/dev/shm")
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "1", "--packed4", p4], check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16); os.unlink(p4); os.rmdir(tmp)
d = dipper_amd.Dipper(0); d.set_msa(packed, L); d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
res = d.nj_run(max_iters=it + 40)
lib = capi.load_library(); lib.dpr_get_nj_phase_stamps.argtypes = [C.c_void_p]
buf = np.zeros(4 * 2048 * 8, np.uint64); assert lib.dpr_get_nj_phase_stamps(buf.ctypes.data) == 0
b = buf[2048 * 8:2 * 2048 * 8].reshape(2048, 8).astype(np.int64)
used = b[:, 0] > 0; t0 = b[used][:, 0].min()
P = 30000 if it < 6000 else None
G16 = (30000 + 15) // 16
blocks = []                       # test blocks in launch order: (strip, first group), 64 groups each (prep_blocks)
c = 0
while 32 * c < G16 and c * 512 < 30000 - 1:
    g0 = 32 * c
    while g0 < G16:
        blocks.append((c, g0)); g0 += 64
    c += 1
x, y = int(res["merge_x"][it]), int(res["merge_y"][it])
print("iteration", it, "blocks stamped", int(used.sum()), "test blocks", len(blocks), "merged slots", x, y)
end = np.where(used, b[:, 6], 0)
order = np.argsort(-end)[:20]
for bx in order:
    role = "test" if bx < len(blocks) else "update"
    info = blocks[bx] if bx < len(blocks) else ("-", "-")
    st = [(int(10 * (v - t0)) if v > 1000000 else None) for v in b[bx][:7]]
    print(f"block {bx:5d} {role:6s} strip {info[0]!s:>3} first group {info[1]!s:>5}  stamps(ns) {st}")
med = np.median(end[used & (np.arange(2048) < len(blocks))] - t0) * 10
print("median end of test blocks", med, "ns; update blocks", np.median(end[used & (np.arange(2048) >= len(blocks))] - t0) * 10)
d.close()
