"""GPU parity of the ROW-SHARDED exact pruned NJ (dipper_amd/csrc/njr.hip; north_star's row-block split of the N x N matrix,
src/neighborJoining.cu:117-148,197-249, under the pruned algorithm) with VIRTUAL ranks: all ranks of the plan live in one
context on one GPU, each with its own epoch buffers (only its own chunks of rows), vectors, lists and window -- the code a rank
runs is the code a process rank runs; only the transport differs (device copies stand in for the all-gathers of the collective
plan; the mailbox plan stores into the other ranks' windows as it would across devices).  Merge logs bit for bit against the
oracle / the single-GPU pruned run.  Process ranks on one GPU: tests/test_gpu_multiproc.py."""
import numpy as np
import pytest

from tests import _util

pytestmark = pytest.mark.gpu

PLANS = {"collective": 1, "mailbox": 2}


def _ctx(world, plan):
    import dipper_amd
    d = dipper_amd.Dipper(0, virtual_world=world)
    d.set_nj_mode(1)
    d.set_nj_multi_plan(3)              # rows sharded, pruned
    d.set_nj_exchange(PLANS[plan])
    return d


def _same(res, ref, what=""):
    assert res["iters"] == ref["iters"], what
    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
        if not np.array_equal(res[key], ref[key][:len(res[key])]):
            bad = int(np.flatnonzero(res[key] != ref[key][:len(res[key])])[0])
            raise AssertionError(f"{what}: {key} differs first at iteration {bad}: {res[key][bad]} vs {ref[key][bad]}")
    assert res["last_d"] == ref["last_d"], what


@pytest.mark.parametrize("plan", ["collective", "mailbox"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_virtual_ranks_equal_oracle(orc, monkeypatch, world, plan):
    """small matrices (one to three ownership chunks of 1 024 positions: some ranks own nothing), many epochs, resumed runs"""
    from dipper_amd import capi
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "200")
    for n, kind in ((2500, "additive"), (1300, "ties"), (40, "additive"), (3, "additive")):
        rng = np.random.default_rng(n + world)
        D = _util.random_additive_matrix(rng, n, zero_frac=0.4 if kind == "ties" else 0.0)
        if kind == "ties":
            D = np.round(D, 1)
        ref = orc.nj_run(np.tril(D, -1), threads=8)
        d = _ctx(world, plan)
        try:
            d.set_matrix_full(D)
            d.dist_matrix(capi.SRC_MATRIX)
            parts = [d.nj_run(max_iters=k) for k in (n // 3, 7, -1)]
            for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
                assert np.array_equal(np.concatenate([p[key] for p in parts]), ref[key]), (key, n, kind)
            assert parts[-1]["last_d"] == ref["last_d"]
            info = d.nj_exchange_info()
            assert info["launches"] >= 0
        finally:
            d.close()


@pytest.mark.parametrize("world,plan", [(4, "collective"), (8, "mailbox"), (3, "mailbox")])
def test_virtual_ranks_msa_8k_equal_single_gpu_and_oracle_prefix(orc, world, plan):
    """8 000 tips (8 chunks; natural epochs 8 000 -> 6 400 -> ... -> 2 097): the distances come from the sharded distance kernels
    (tip-order rows block-cyclic by 64), the epoch builds are sharded permutes; whole log == single-GPU pruned run == oracle."""
    import dipper_amd
    from dipper_amd import capi
    n, L = 8000, 500
    seqs = _util.synth_alignment(np.random.default_rng(7), n, L, mean_bl=1e-3, lo=1e-4, hi=1e-2)
    packed = capi.pack4_many(seqs)
    one = dipper_amd.Dipper(0)
    try:
        one.set_nj_mode(1)
        one.set_msa(packed, L)
        one.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
        M = one.matrix()
        ref1 = one.nj_run()
    finally:
        one.close()
    d = _ctx(world, plan)
    try:
        d.set_msa(packed, L)
        d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
        assert np.array_equal(d.matrix(), M)              # (through the position-space rows of their owners)
        res = d.nj_run()
        info = d.nj_exchange_info()
    finally:
        d.close()
    _same(res, ref1, f"{world} virtual ranks, {plan}")
    assert res["iters"] == n - 2
    assert info["collectives"] == (2 * (n - 2) if plan == "collective" else 0), info
    ref = orc.nj_run(np.tril(M, -1), threads=8, max_iters=300)
    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
        assert np.array_equal(res[key][:300], ref[key][:300]), key


@pytest.mark.parametrize("world,n,grid", [(5, 3000, None), (6, 9300, None), (7, 1024, None), (8, 4096, "4"), (3, 5121, "7")])
def test_virtual_ranks_odd_worlds_chunk_edges_and_tiny_scan_grids(orc, monkeypatch, world, n, grid):
    """rank counts that are no powers of two (the kernels divide chunk indices by the rank count through a 16-bit reciprocal),
    position counts on and next to chunk boundaries, ranks without rows, and unit-scan grids smaller than the rank count (every
    rank gets one scan block that walks all its listed units): the oracle's log, mailbox plan, many epochs"""
    from dipper_amd import capi
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "500")
    if grid:
        monkeypatch.setenv("DPR_NJP_GRID", grid)
    rng = np.random.default_rng(n * 31 + world)
    D = np.round(rng.random((n, n)) * 0.9 + 0.1, 3)
    D = np.tril(D, -1) + np.tril(D, -1).T
    ref = orc.nj_run(np.tril(D, -1), threads=16)
    d = _ctx(world, "mailbox")
    try:
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        assert "row-sharded pruned" in d.nj_exchange_info()["note"]
        res = d.nj_run()
    finally:
        d.close()
    _same(res, ref, f"{world} ranks, {n} tips")


@pytest.mark.parametrize("case", [0, 1, 2, 3], ids=["nan_pair", "inf_few", "nan_and_inf", "inf_row"])
def test_virtual_ranks_nonfinite(orc, monkeypatch, case):
    from tests import _nonfinite
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "300")
    n = 2200
    name, D = _nonfinite.matrices(n, 61)[case]
    d = _ctx(4, "mailbox")
    try:
        _nonfinite.check(d, orc, D, chunks=(n // 2, 3, 10 ** 9), threads=8)
    finally:
        d.close()


def test_virtual_ranks_matrix_state_after_partial_run(orc, monkeypatch):
    """after k iterations the sharded position-space rows and the replicated row sums equal the oracle's active matrix"""
    from dipper_amd import capi
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "256")
    n = 1500
    D = _util.random_additive_matrix(np.random.default_rng(15), n)
    d = _ctx(3, "collective")
    try:
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        k = n // 2 + 30
        d.nj_run(max_iters=k)
        na = n - k
        part = orc.nj_run(np.tril(D, -1), max_iters=k, threads=8)
        assert np.array_equal(d.row_sums()[:na], part["U"][:na])
        assert np.array_equal(d.matrix()[:na, :na], part["D"][:na, :na])
    finally:
        d.close()


def test_replicated_state_check_catches_a_diverging_rank(monkeypatch):
    """dpr_ctx_set_debug_fault(iteration, rank): that rank's header record of that iteration carries a wrong row-sum word -- every
    rank's extract kernel compares the words and the run ends with DPR_ERR_COMM instead of going on with ranks that disagree."""
    import dipper_amd
    from dipper_amd import capi
    D = _util.random_additive_matrix(np.random.default_rng(3), 1200, zero_frac=0.2)
    for plan in ("collective", "mailbox"):
        d = _ctx(4, plan)
        try:
            d.set_debug_fault(25, 2)
            d.set_matrix_full(D)
            d.dist_matrix(capi.SRC_MATRIX)
            with pytest.raises(capi.DipperError) as ei:
                d.nj_run()
            assert ei.value.code == -5 and "row sums differ after 25 iterations" in str(ei.value), str(ei.value)
        finally:
            d.close()
