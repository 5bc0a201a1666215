"""Worker of tests/test_gpu_multiproc.py: ONE rank of a row-sharded NJ run whose ranks are separate PROCESSES on one GPU
(dpr_comm_init_local: no RCCL -- it refuses two ranks on one device; mailbox plan of njs.hip).  The parent carries the
192-byte peer descriptions between the workers through their pipes.
    python -m tests._njs_worker <rank> <world> <n> <seed> <out.npz> [source]"""
import os
import sys

import numpy as np


def main():
    rank, world, n, seed = (int(v) for v in sys.argv[1:5])
    out = sys.argv[5]
    source = sys.argv[6] if len(sys.argv) > 6 else "matrix"
    import dipper_amd
    from dipper_amd import capi
    from tests import _util
    rng = np.random.default_rng(seed)
    d = dipper_amd.Dipper(0)
    d.comm_init_local(rank, world)
    if os.environ.get("DPR_TEST_NJ_ROWS_PRUNED") == "1":     # the row-sharded exact pruned NJ (njr.hip) instead of the streaming loop
        d.set_nj_mode(1)
        d.set_nj_multi_plan(3)
    else:
        d.set_nj_mode(0)
    blob = d.peer_export(n)
    sys.stdout.write(blob.hex() + "\n")
    sys.stdout.flush()
    blobs = [bytes.fromhex(sys.stdin.readline().strip()) for _ in range(world)]
    d.peer_attach(blobs)
    if source == "msa":
        L = 800
        seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
        d.set_msa(capi.pack4_many(seqs), L)
        d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    else:
        D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
    info0 = d.nj_exchange_info()
    if os.environ.get("DPR_TEST_NJS_FAULT"):      # "iteration,rank": read HERE (the test's worker), handed to the context's debug setter
        fi, fr = os.environ["DPR_TEST_NJS_FAULT"].split(",")
        d.set_debug_fault(int(fi), int(fr))
        # tests of the cross-check: the run must END with DPR_ERR_COMM on every rank (the message names the row sums)
        try:
            d.nj_run()
            verdict = "no error"
        except capi.DipperError as e:
            verdict = "code %d: %s" % (e.code, e)
        np.savez(out, verdict=verdict, plan=info0["plan"])
        d.close()
        sys.stdout.write("done\n")
        sys.stdout.flush()
        return
    first = d.nj_run(max_iters=3)          # a resumed run: barrier + flush between the calls
    k = first["iters"]
    res = d.nj_run()
    info = d.nj_exchange_info()
    merged = {key: np.concatenate([first[key][:k], res[key][:res["iters"]]]) for key in ("merge_x", "merge_y", "bl_x", "bl_y")}
    np.savez(out, iters=k + res["iters"], last_d=res["last_d"], plan=info0["plan"], launches=info["launches"],
             collectives=info["collectives"], **merged)
    d.close()
    sys.stdout.write("done\n")
    sys.stdout.flush()


if __name__ == "__main__":
    main()
