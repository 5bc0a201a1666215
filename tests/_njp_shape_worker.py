"""Worker of tests/test_gpu_nj.py::test_large_shape_post_kernels: the pruned NJ with the LARGE launch shape of the post kernel
forced at small sizes (DPR_NJ_BIG_P is read once per process, hence a process of its own), checked against the oracle.
    python -m tests._njp_shape_worker  -> prints one JSON line"""
import json
import sys

import numpy as np


def main():
    import os
    import dipper_amd
    from dipper_amd import capi
    from tests import _orc, _util
    orc = _orc.load()
    poison = os.environ.get("DPR_SHAPE_POISON")
    if poison:
        # every context of this process is created on memory that held this byte pattern (0xFF: NaN as fp64): what a kernel reads
        # without having written it -- the arrays njp_post2_kernel hands from launch to launch, the coarse bounds -- shows
        from tests.conftest import dirty_device_memory
        capi.load_library()
        real = dipper_amd.Dipper

        def dirty_dipper(*a, **k):
            dirty_device_memory(3 << 30, int(poison))
            return real(*a, **k)
        dipper_amd.Dipper = dirty_dipper
    out = []
    # ("ties" and "random" create negative distances -- (1 + 1 - 3) / 2 -- i.e. the slack branch of njp_post2_kernel's bounds)
    cases = [("additive", 700, 3), ("additive", 2100, 4), ("ties", 900, 5), ("additive", 4500, 6), ("msa", 5000, 7), ("random", 1800, 8)]
    for kind, n, seed in cases:
        rng = np.random.default_rng(seed)
        d = dipper_amd.Dipper(0)
        try:
            d.set_nj_mode(1)
            d.set_nj_adaptive(0)                 # the pruned plan from the first to the last iteration
            if kind == "msa":
                seqs = _util.synth_alignment(rng, n, 600, mean_bl=2e-3, lo=2e-4, hi=2e-2)
                d.set_msa(capi.pack4_many(seqs), 600)
                d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
                D = d.matrix()
            else:
                if kind == "ties":
                    D = rng.integers(1, 4, size=(n, n)).astype(np.float64)
                    D = np.tril(D, -1) + np.tril(D, -1).T
                elif kind == "random":
                    D = rng.random((n, n)) * 3.0 + 0.01
                    D = np.tril(D, -1) + np.tril(D, -1).T
                else:
                    D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
                d.set_matrix_full(D)
                d.dist_matrix(capi.SRC_MATRIX)
            res = d.nj_run()
        finally:
            d.close()
        ref = orc.nj_run(np.tril(D, -1))
        ok = res["iters"] == n - 2 and all(np.array_equal(res[k], ref[k]) for k in ("merge_x", "merge_y", "bl_x", "bl_y")) and res["last_d"] == ref["last_d"]
        out.append({"kind": kind, "n": n, "ok": bool(ok)})
    # resumed runs (dpr_nj_run with max_iters: the maxima njp_post2_kernel hands from launch to launch live across the calls) and
    # the adaptive plan (small-integer distances: the run is handed to the streaming loop and back; every hand-back builds an epoch)
    for kind, n, seed in (("resumed", 3000, 11), ("adaptive", 2600, 12)):
        rng = np.random.default_rng(seed)
        if kind == "adaptive":
            D = rng.integers(1, 4, size=(n, n)).astype(np.float64)
            D = np.tril(D, -1) + np.tril(D, -1).T
        else:
            D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
        ref = orc.nj_run(np.tril(D, -1))
        d = dipper_amd.Dipper(0)
        try:
            d.set_nj_mode(1)
            d.set_nj_adaptive(1 if kind == "adaptive" else 0)
            d.set_matrix_full(D)
            d.dist_matrix(capi.SRC_MATRIX)
            got = {k: [] for k in ("merge_x", "merge_y", "bl_x", "bl_y")}
            done = 0
            for chunk in ((1, 7, 100, 33, 1000, 10 ** 6) if kind == "resumed" else (10 ** 6,)):
                res = d.nj_run(max_iters=chunk)
                for key in got:
                    got[key].append(res[key][:res["iters"]])
                done += res["iters"]
            stats = d.nj_adaptive_stats() if kind == "adaptive" else None
        finally:
            d.close()
        ok = done == n - 2 and all(np.array_equal(np.concatenate(got[k]), ref[k]) for k in got) and res["last_d"] == ref["last_d"]
        out.append({"kind": kind, "n": n, "ok": bool(ok), "stats": str(stats)})
    # the unit-sharded multi-GPU plan emulated on one GPU (virtual ranks: every rank tests and scans only its own units, the update
    # runs once with the first rank's launch, the other ranks' launches are tests only)
    for world, kind, n, seed in ((3, "additive", 2500, 21), (8, "ties", 1300, 22), (4, "msa", 3000, 23)):
        rng = np.random.default_rng(seed)
        capi.set_nj_virtual_shards(world)
        try:
            d = dipper_amd.Dipper(0)
            try:
                d.set_nj_mode(1)
                d.set_nj_adaptive(0)
                if kind == "msa":
                    seqs = _util.synth_alignment(rng, n, 500, mean_bl=2e-3, lo=2e-4, hi=2e-2)
                    d.set_msa(capi.pack4_many(seqs), 500)
                    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
                    D = d.matrix()
                else:
                    if kind == "ties":
                        D = rng.integers(1, 4, size=(n, n)).astype(np.float64)
                        D = np.tril(D, -1) + np.tril(D, -1).T
                    else:
                        D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
                    d.set_matrix_full(D)
                    d.dist_matrix(capi.SRC_MATRIX)
                parts = [d.nj_run(max_iters=k) for k in (n // 3, 7, -1)]
            finally:
                d.close()
        finally:
            capi.set_nj_virtual_shards(1)
        ref = orc.nj_run(np.tril(D, -1))
        ok = all(np.array_equal(np.concatenate([q[key][:q["iters"]] for q in parts]), ref[key]) for key in ("merge_x", "merge_y", "bl_x", "bl_y")) \
            and parts[-1]["last_d"] == ref["last_d"]
        out.append({"kind": "unit-sharded x%d %s" % (world, kind), "n": n, "ok": bool(ok)})
    # NaN / +inf distances that leave the run going (tests/_nonfinite.py) on the large shape: njp_post2_kernel takes its maxima
    # from the previous launch and tracks the range of the entries (eabs = +inf here); epochs rebuilt along the way
    from tests import _nonfinite
    os.environ["DPR_NJ_EPOCH_MIN"] = "256"      # (read at every dpr_nj_run)
    for name, D in _nonfinite.matrices(1500, 41):
        d = dipper_amd.Dipper(0)
        try:
            d.set_nj_mode(1)
            d.set_nj_adaptive(0)
            try:
                done, code = _nonfinite.check(d, orc, D, chunks=(400, 3, 10 ** 9), threads=8)
                ok = done >= 1490
            except AssertionError as e:
                ok = False
                sys.stderr.write("nonfinite %s: %s\n" % (name, e))
        finally:
            d.close()
        out.append({"kind": "nonfinite " + name, "n": 1500, "ok": bool(ok)})
    del os.environ["DPR_NJ_EPOCH_MIN"]
    # no Q candidate below 10000 from the first iteration on (tests/test_gpu_nj.py::test_nj_no_candidate): error -4, no hang
    for n in (8, 1500):
        D = np.full((n, n), -1.0e5)
        np.fill_diagonal(D, 0.0)
        d = dipper_amd.Dipper(0)
        try:
            d.set_nj_mode(1)
            d.set_nj_adaptive(0)
            d.set_matrix_full(D)
            d.dist_matrix(capi.SRC_MATRIX)
            try:
                d.nj_run()
                code = 0
            except dipper_amd.DipperError as e:
                code = e.code
        finally:
            d.close()
        out.append({"kind": "no candidate", "n": n, "ok": code == -4})
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
