"""GPU tests of the multi-GPU (row-sharded) NJ path on ONE GPU: virtual ranks run the same sharded
kernels and buffers, with the two per-iteration all-gathers done as device copies; plus an RCCL
1-rank round trip for the transport plumbing."""
import numpy as np
import pytest

from tests import _util

pytestmark = pytest.mark.gpu


PLANS = {"legacy": 0, "peer": 1, "mailbox": 2}


@pytest.mark.parametrize("plan", ["legacy", "peer", "mailbox"])
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("n", [5, 64, 130, 700])
def test_virtual_ranks_match_oracle(orc, world, n, plan):
    """every exchange plan of the row-sharded loop (njs.hip: one exchange + two launches per iteration with the rows of
    the merged pair pulled through row views; nj.hip: round 2's two-exchange loop) gives the oracle's merge log"""
    import dipper_amd
    from dipper_amd import capi
    rng = np.random.default_rng(n * 10 + world)
    D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
    d = dipper_amd.Dipper(0, virtual_world=world)
    try:
        d.set_nj_exchange(PLANS[plan])
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        assert d.nj_exchange_info()["plan"] == plan
        Dsym = np.tril(D, -1) + np.tril(D, -1).T
        assert np.array_equal(d.matrix(), Dsym)
        U = orc.row_sums(np.ascontiguousarray(Dsym))
        assert np.array_equal(d.row_sums(), U)
        rc, i, j, q = orc.nj_argmin(np.ascontiguousarray(Dsym), n, U)
        gi, gj, gq, _ = d.argmin_once()
        assert (gi, gj, gq) == (i, j, q)
        ref = orc.nj_run(np.tril(D, -1))
        res = d.nj_run()
        assert res["iters"] == n - 2
        for k in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(res[k], ref[k]), k
        assert res["last_d"] == ref["last_d"]
        info = d.nj_exchange_info()
        per_it = 4 if plan == "legacy" else 2
        assert info["launches"] == per_it * (n - 2)
        # the matrix the hooks see after the run is consistent again (row buffers flushed): D[1][0] is the last distance
        if n > 2:
            assert d.matrix_row(1)[0] == ref["last_d"]
    finally:
        d.close()


@pytest.mark.parametrize("plan", ["peer", "mailbox"])
def test_virtual_ranks_interrupted_runs_and_ties(orc, plan):
    """resumed runs (dpr_nj_run with max_iters: the row buffers are flushed at the end of every call and the next call
    starts without pending rows) and tie-heavy input, where the new node is merged again at once -- the case in which a
    pulled row comes from a row buffer instead of the matrix"""
    import dipper_amd
    from dipper_amd import capi
    rng = np.random.default_rng(5)
    n = 400
    D = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    D = np.tril(D, -1) + np.tril(D, -1).T
    ref = orc.nj_run(np.tril(D, -1))
    for world in (2, 5):
        d = dipper_amd.Dipper(0, virtual_world=world)
        try:
            d.set_nj_exchange(PLANS[plan])
            d.set_matrix_full(D)
            d.dist_matrix(capi.SRC_MATRIX)
            got = {k: [] for k in ("merge_x", "merge_y", "bl_x", "bl_y")}
            done = 0
            for chunk in (1, 2, 7, 1, 100, 3, 10 ** 6):
                res = d.nj_run(max_iters=chunk)
                k = res["iters"]
                for key in got:
                    got[key].append(res[key][:k])
                done += k
                if done < n - 2:      # between the calls the matrix in memory is the oracle's
                    cur = orc.nj_run(np.tril(D, -1), max_iters=done)
                    assert np.array_equal(d.row_sums()[:n - done], cur["U"][:n - done])
            assert done == n - 2
            for key in got:
                assert np.array_equal(np.concatenate(got[key]), ref[key]), key
            assert res["last_d"] == ref["last_d"]
        finally:
            d.close()


@pytest.mark.parametrize("plan", ["peer", "mailbox"])
def test_virtual_ranks_detect_a_corrupted_pull(monkeypatch, plan):
    """the cross-check of the one-exchange plans (NjsRec::ux, njs.hip): dpr_ctx_set_debug_fault(25, 2) makes virtual rank 2 of 4 use a
    wrong value for one element of a pulled row at iteration 25; the ranks' replicated row sums differ from then on and the
    run ends with DPR_ERR_COMM at iteration 26 -- while the same run without the fault completes."""
    import dipper_amd
    from dipper_amd import capi
    D = _util.random_additive_matrix(np.random.default_rng(3), 300, zero_frac=0.2)
    for fault in (None, (25, 2)):
        d = dipper_amd.Dipper(0, virtual_world=4)
        try:
            d.set_nj_exchange(PLANS[plan])
            if fault:
                d.set_debug_fault(*fault)
            d.set_matrix_full(D)
            d.dist_matrix(capi.SRC_MATRIX)
            if fault is None:
                assert d.nj_run()["iters"] == 298
            else:
                with pytest.raises(capi.DipperError) as ei:
                    d.nj_run()
                assert ei.value.code == -5 and "row sums differ after 26 iterations" in str(ei.value)
        finally:
            d.close()


def test_virtual_ranks_msa(orc):
    import dipper_amd
    from dipper_amd import capi
    rng = np.random.default_rng(77)
    n, L = 333, 1500
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    packed = capi.pack4_many(seqs)
    one = dipper_amd.Dipper(0)
    one.set_msa(packed, L)
    one.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    M1 = one.matrix()
    r1 = one.nj_run()
    one.close()
    d = dipper_amd.Dipper(0, virtual_world=4)
    d.set_msa(packed, L)
    d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    assert np.array_equal(d.matrix(), M1)
    r4 = d.nj_run()
    d.close()
    for k in ("merge_x", "merge_y", "bl_x", "bl_y"):
        assert np.array_equal(r4[k], r1[k]), k
    assert r4["last_d"] == r1["last_d"]


def test_rccl_single_rank_roundtrip():
    import dipper_amd
    d = dipper_amd.Dipper(0)
    try:
        d.comm_selftest()
        uid = d.comm_unique_id()
        assert len(uid) == 128
    finally:
        d.close()
