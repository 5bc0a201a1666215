"""Phase stamps of ONE launch of njp_post2_kernel (large shape of the pruned NJ): python profiles/nj_phases2.py [tips sites iteration]
stamps (100 MHz wall clock, thread 0 of every block): 0 start, 1 first loads arrived, 2 winner known, 3 winner-dependent loads
arrived, 4 coarse test passed (T), 5 unit bounds arrived (T), 6 end; word 7: role 1 U, 2 M, 3 T skipped by its coarse bound, 4 T full"""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
n, L, it = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100000, 10000, 5000)
os.environ["DPR_NJ_PHASES"] = str(it)
import dipper_amd
from dipper_amd import capi
k = 10000 / L
tmp = tempfile.mkdtemp(prefix="njph_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
p4 = os.path.join(tmp, "a.p4")
subprocess.run([os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools", "bin", "gen_synth"), "--tips", str(n), "--sites", str(L), "--seed", "8",
                "--mean-bl", repr(2e-5 * k), "--lo", repr(2e-6 * k), "--hi", repr(2e-4 * k), "--packed4", p4], check=True)
packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
os.unlink(p4); os.rmdir(tmp)
d = dipper_amd.Dipper(0)
d.set_msa(packed, L)
d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
d.nj_run(max_iters=it + 64)
buf = np.zeros(4 * 2048 * 8, np.uint64)
L_ = capi.load_library()
L_.dpr_get_nj_phase_stamps.argtypes = [C.c_void_p]
assert L_.dpr_get_nj_phase_stamps(buf.ctypes.data) == 0
b = buf[:2 * 2048 * 8].reshape(2, 2048, 8).astype(np.int64)[1]
used = b[:, 0] > 0
t0 = b[used][:, 0].min()
print(f"post2 launch of iteration {it}: {used.sum()} blocks stamped; ns since the first block's start")
for code, nm in ((1, "U"), (2, "M"), (3, "T skipped"), (4, "T full"), (0, "(returned early)")):
    m = used & (b[:, 7] == code)
    if not m.any():
        continue
    row = f"  {nm:16s} {m.sum():5d} blocks:"
    for j in range(7):
        ok = b[m][:, j] > 0
        if ok.any():
            v = 10 * (b[m][ok][:, j] - t0)
            row += f"  s{j} med {int(np.median(v)):6d} max {v.max():6d} |"
    print(row)
d.close()
