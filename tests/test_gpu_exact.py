"""GPU parity of the exact placement mode (src/placement.cu) through the C ABI:
tree state, reverse slots, node depths and per-tip (eid, frac, add) against the oracle's literal
restatement (bit-exact)."""
import numpy as np
import pytest

from tests import _util
from tests.test_gpu_mash_place import _reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import dipper_amd
    d = dipper_amd.Dipper(0)
    yield d
    d.close()


def _same_exact_state(got, ref, n):
    live = 4 * n - 4
    assert ref["next_slot"] == live
    assert np.array_equal(got["trace"][2:], ref["trace"][2:])
    for key in ("head", "e", "nxt", "belong", "len", "rev"):
        m = 2 * n - 1 if key == "head" else live      # initialize covers 2N-1 nodes (src/placement.cu:521-527)
        assert np.array_equal(got[key][:m], ref[key][:m]), key
    assert np.array_equal(got["dep"][:2 * n - 1], ref["dep"][:2 * n - 1])


@pytest.mark.parametrize("n", [3, 4, 9, 64, 300, 1100, 2600])
def test_exact_matrix_source(gpu, orc, n):
    from dipper_amd import capi
    rng = np.random.default_rng(n)
    D = _util.random_additive_matrix(rng, n, zero_frac=0.3 if n > 9 else 0.0)
    D *= 0.9 / D.max()
    gpu.set_matrix_full(D)
    got = gpu.place_exact_run(capi.SRC_MATRIX, n)
    ref = orc.place_exact_run(D)
    _same_exact_state(got, ref, n)
    names = [f"T{i}" for i in range(n)]
    nw = _util.newick_from_placement(names, got["head"], got["e"], got["nxt"], got["len"], n, fmt=repr)
    assert np.abs(_util.patristic(nw, names) - D).max() < 1e-12 * n


def test_exact_noisy_and_caterpillar(gpu, orc):
    """non-additive input (clamps, ties, the (0,0,2) tuples) and a ladder-shaped metric (depth ~ n:
    one tree level per workgroup barrier, level lists longer than the workgroup)"""
    from dipper_amd import capi
    rng = np.random.default_rng(99)
    n = 500
    D = np.round(rng.random((n, n)) * 0.5, 2)
    D = np.tril(D, -1) + np.tril(D, -1).T
    gpu.set_matrix_full(D)
    _same_exact_state(gpu.place_exact_run(capi.SRC_MATRIX, n), orc.place_exact_run(D), n)
    n = 700
    pos = np.cumsum(rng.uniform(0.0005, 0.001, size=n))        # caterpillar: tips hang off a path
    pend = rng.uniform(0.0005, 0.001, size=n)
    D = np.abs(pos[:, None] - pos[None, :]) + pend[:, None] + pend[None, :]
    np.fill_diagonal(D, 0.0)
    order = np.argsort(pos)                                     # insertion along the path -> deep tree
    D = D[order][:, order]
    gpu.set_matrix_full(D)
    ref = orc.place_exact_run(D)
    assert ref["dep"][:2 * n - 1].max() > 300
    _same_exact_state(gpu.place_exact_run(capi.SRC_MATRIX, n), ref, n)


def test_exact_msa_and_mash_sources(gpu, orc):
    from dipper_amd import capi
    rng = np.random.default_rng(123)
    n, L = 700, 2000         # more than two row batches of 256
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    gpu.set_msa(capi.pack4_many(seqs), L)
    gpu.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    M = gpu.matrix()
    got = gpu.place_exact_run(capi.SRC_MSA, n, dist_type=capi.DIST_JC)
    _same_exact_state(got, orc.place_exact_run(M), n)

    reads = _reads(rng, 300, 3000, 6000)
    gpu.set_reads(reads)
    gpu.sketch(k=15, S=1000, fetch=False)
    gpu.dist_matrix(capi.SRC_MASH, 0, 15)
    M = gpu.matrix()
    got = gpu.place_exact_run(capi.SRC_MASH, len(reads), k=15)
    _same_exact_state(got, orc.place_exact_run(M), len(reads))


def test_exact_default_tuple_wins_falls_back_to_literal_schedule(gpu, orc):
    """Large distances: the pendant length of every candidate exceeds 2, so the default tuple (slot 0, 2.0) wins the argmin,
    the reference's swap in updateTreeStructure (src/placement.cu:236-239) is taken and its depths stop being tree depths.
    Only the level-by-depth schedule reproduces what the reference computes from there on: the fast schedule flags the
    case and the run is repeated literally -- same state as the oracle."""
    from dipper_amd import capi
    rng = np.random.default_rng(4)
    n = 120
    D = _util.random_additive_matrix(rng, n) * 6.0
    gpu.set_matrix_full(D)
    got = gpu.place_exact_run(capi.SRC_MATRIX, n)
    ref = orc.place_exact_run(D)
    assert np.any(ref["trace"][2:, 2] == 2.0)          # the default tuple did win
    _same_exact_state(got, ref, n)


@pytest.mark.parametrize("n", [9, 300, 1100])
def test_exact_literal_schedule_forced(monkeypatch, orc, n):
    """the one-workgroup literal schedule (rounds 1-2) stays under test: DPR_EXACT_LITERAL=1"""
    import dipper_amd
    from dipper_amd import capi
    monkeypatch.setenv("DPR_EXACT_LITERAL", "1")
    rng = np.random.default_rng(n)
    D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
    D *= 0.9 / D.max()
    d = dipper_amd.Dipper(0)
    try:
        d.set_matrix_full(D)
        _same_exact_state(d.place_exact_run(capi.SRC_MATRIX, n), orc.place_exact_run(D), n)
    finally:
        d.close()


@pytest.mark.parametrize("top_mem", [False, "poll", True, "levels"], ids=["top_climbing_lds", "top_polling_lds", "top_in_memory", "top_levels_lds"])
def test_exact_larger_tree_both_top_variants(monkeypatch, orc, top_mem):
    """4 000 tips: hundreds of top nodes above the one-wavefront subtrees; the top-tree pass (round 6) by climbs that meet at the
    parents' LDS words, (forced: what top trees of 2 048 - 4 096 nodes use) with every node polling the LDS words of the values it
    waits for, (forced: what trees beyond ~100 000 tips use) level by level with its values in memory, and (forced: rounds 3-5)
    level by level with a workgroup barrier per level and the values in LDS"""
    import dipper_amd
    from dipper_amd import capi
    if top_mem is True:
        monkeypatch.setenv("DPR_EXACT_TOP_MEM", "1")
    elif top_mem == "levels":
        monkeypatch.setenv("DPR_EXACT_TOP_LEVELS", "1")
    elif top_mem == "poll":
        monkeypatch.setenv("DPR_EXACT_TOP_POLL", "1")
    rng = np.random.default_rng(77)
    n = 4000
    D = _util.random_additive_matrix(rng, n, zero_frac=0.2)
    D *= 0.9 / D.max()
    ref = orc.place_exact_run(D)
    d = dipper_amd.Dipper(0)
    try:
        d.set_matrix_full(D)
        _same_exact_state(d.place_exact_run(capi.SRC_MATRIX, n), ref, n)
    finally:
        d.close()


@pytest.mark.parametrize("n", [300, 4000])
@pytest.mark.parametrize("sm", [256, 512, 1024])
def test_exact_small_subtrees_of_one_workgroup(monkeypatch, orc, sm, n):
    """DPR_EXACT_SM: small subtrees of up to 256 / 512 / 1 024 nodes, one WORKGROUP each with a barrier per level and 1 / 2 / 4 nodes per thread (what a run switches to
    when its top tree approaches the 2 047 nodes of the climbing schedule; forced here from the first tip on)"""
    import dipper_amd
    from dipper_amd import capi
    monkeypatch.setenv("DPR_EXACT_SM", str(sm))
    rng = np.random.default_rng(sm + n)
    D = _util.random_additive_matrix(rng, n, zero_frac=0.2)
    D *= 0.9 / D.max()
    ref = orc.place_exact_run(D)
    d = dipper_amd.Dipper(0)
    try:
        d.set_matrix_full(D)
        _same_exact_state(d.place_exact_run(capi.SRC_MATRIX, n), ref, n)
    finally:
        d.close()
