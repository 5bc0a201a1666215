"""Worst-case table of the two single-GPU NJ algorithms (verdict r2 item 3): NJ time, units scanned and device memory at
30 000 tips for alignments of growing divergence and for a uniform-random matrix, exact pruned scan vs full streaming scan.
  python profiles/nj_worstcase.py [tips] [sites] > gpurun_out/r3/nj_worstcase.jsonl
Inputs from tools/bin/gen_synth (seeded); one JSON line per (input, algorithm)."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import dipper_amd
from dipper_amd import capi

ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
GEN = os.path.join(ROOT, "tools", "bin", "gen_synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
BUDGET = float(os.environ.get("NJWC_BUDGET_S", "25"))
hip = C.CDLL("libamdhip64.so")
hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]


def used_gb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return (t.value - f.value) / 1e9


def run(tag, setup, dist_type, modes=(1, 0, -1)):
    for mode in modes:
        d = dipper_amd.Dipper(0)
        rec = {"input": tag, "tips": n, "algorithm": {1: "pruned only", 0: "stream only", -1: "default (adaptive)"}[mode]}
        try:
            d.set_nj_mode(0 if mode == 0 else 1)
            if mode == 1:
                d.set_nj_adaptive(0)          # pruned scans only
            setup(d)
            base = used_gb()
            t0 = time.perf_counter()
            d.dist_matrix(*dist_type)
            rec["device_gb_after_build"] = used_gb()
            # in chunks of 2000 iterations (dpr_nj_run resumes), at most BUDGET seconds of NJ per case: a case that does not
            # finish reports the iterations it reached (the early iterations are the expensive ones)
            import hashlib
            h = hashlib.sha256()
            done, nj_ms_tot, dist_ms = 0, 0.0, None
            peak = used_gb()
            try:
                while done < n - 2 and nj_ms_tot < BUDGET * 1e3:
                    res = d.nj_run(max_iters=2000)
                    k = int(res["iters"])
                    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
                        h.update(np.ascontiguousarray(res[key][:k]).tobytes())
                    dm, nm = d.timing()
                    dist_ms = dm if dist_ms is None else dist_ms
                    nj_ms_tot += nm
                    done += k
                    peak = max(peak, used_gb())
                    if k == 0:
                        break
                rec["digest"] = h.hexdigest()[:16]
            except capi.DipperError as e:
                rec["error"] = str(e)[:160]
            rec["iters"] = done
            rec["finished"] = done == n - 2
            rec["wall_s"] = time.perf_counter() - t0
            rec.update(dist_ms=dist_ms, nj_ms=nj_ms_tot, device_gb_peak_seen=max(base, peak))
            if mode != 0:
                try:
                    sc, full = d.prune_stats()
                    rec.update(units_listed=int(sc), units_per_full_scan=int(full), listed_fraction_of_full_scans=sc / (full * (n - 2.0)))
                    si, se = d.nj_adaptive_stats()
                    rec.update(streamed_iterations=si, epochs_switched_to_streaming=se)
                except Exception:
                    pass
        finally:
            d.close()
        print(json.dumps(rec), flush=True)


tmp = tempfile.mkdtemp(prefix="njwc_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    for mean in (2e-5, 1e-4, 1e-3, 1e-2, 1e-1):
        tag = "alignment mean branch %g" % mean
        if only and ("%g" % mean) not in only:
            continue
        p4 = os.path.join(tmp, "a.p4")
        subprocess.run([GEN, "--tips", str(n), "--sites", str(L), "--seed", "1", "--mean-bl", str(mean), "--lo", str(mean / 10),
                        "--hi", str(mean * 10), "--packed4", p4], check=True)
        packed = np.fromfile(p4, dtype=np.uint64).reshape(n, (L + 15) // 16)
        os.unlink(p4)
        # JC69 (-d 2) as BASELINE's configs; where JC saturates (p >= 0.75 -> inf / NaN) also the uncorrected distance (-d 1)
        run(tag + ", JC69", lambda d: d.set_msa(packed, L), (capi.SRC_MSA, capi.DIST_JC))
        if mean >= 1e-2:
            run(tag + ", p-distance", lambda d: d.set_msa(packed, L), (capi.SRC_MSA, 1))
        del packed
    if not only or "const" in only:
        m = n * (n - 1) // 2
        run("constant matrix (every distance 1: one global tie)", lambda d: d.set_matrix_lower(np.ones(m), n), (capi.SRC_MATRIX,))
    if not only or "ints" in only:
        m = n * (n - 1) // 2
        ints = np.random.default_rng(4).integers(1, 4, size=m).astype(np.float64)
        run("small integers 1..3 (ties everywhere)", lambda d: d.set_matrix_lower(ints, n), (capi.SRC_MATRIX,))
    if not only or "random" in only:
        rng = np.random.default_rng(3)
        m = n * (n - 1) // 2
        low = rng.random(m)          # packed strict lower triangle, uniform [0, 1): what `-i d` may feed (src/matrix_reader.cu:23-45)
        run("uniform random matrix", lambda d: d.set_matrix_lower(low, n), (capi.SRC_MATRIX,))
finally:
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
