"""GPU parity of the divide-and-conquer mode through the C ABI: backbone tree,
cluster assignment and the concurrent cluster trees against the oracle's sequential restatement of
src/divide_and_conquer/placement_close_k.cu on the GPU's own distance matrix (bit-exact state)."""
import os

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


def _same_dc_state(got, ref, n, B):
    live = 4 * n - 4
    assert ref["next_slot"] == live
    assert np.array_equal(got["cluster_id"], ref["cluster_id"])
    for key in ("head", "e", "nxt", "belong", "len"):
        m = 2 * n if key == "head" else live
        assert np.array_equal(got[key][:m], ref[key][:m]), key
    assert np.array_equal(got["cid"][:5 * live], ref["cid"][:5 * live])
    assert np.array_equal(got["cdis"][:5 * live], ref["cdis"][:5 * live])
    assert np.array_equal(got["trace"][2:B], ref["trace"][2:B])              # backbone: (eid, frac, add)
    assert np.array_equal(got["trace"][B:, 1:], ref["trace"][B:, 1:])        # members: (frac, add)


def _clone_heavy_alignment(rng, n, L, clones, of=3):
    """alignment whose last `clones` tips are light mutations of tip `of`: one big cluster"""
    seqs = _util.synth_alignment(rng, n - clones, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    base = np.frombuffer(seqs[of], dtype=np.uint8)
    for _ in range(clones):
        s = base.copy()
        pos = rng.integers(0, L, size=int(rng.integers(1, 12)))
        s[pos] = _util.BASES[rng.integers(0, 4, size=len(pos))]
        seqs.append(s.tobytes())
    return seqs


@pytest.mark.parametrize("n,B,flags", [(600, 120, 0), (600, 120, 1), (60, 25, 0), (1500, 300, 0)])
def test_dc_msa(gpu, orc, n, B, flags):
    from dipper_amd import capi
    rng = np.random.default_rng(n + flags)
    L = 1500
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    gpu.set_msa(capi.pack4_many(seqs), L)
    gpu.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    M = gpu.matrix()
    got = gpu.dc_run(capi.SRC_MSA, n, B, dist_type=capi.DIST_JC, flags=flags)
    ref = orc.dc_run(M, B, skip_last_backbone=0 if flags else 1)
    _same_dc_state(got, ref, n, B)
    assert got["stats"]["clusters"] == len(set(ref["cluster_id"][B:]))
    names = [f"T{i}" for i in range(n)]
    nw = _util.newick_from_placement(names, got["head"], got["e"], got["nxt"], got["len"], n, fmt=repr)
    assert sorted(_util.parse_newick(nw)[2].values()) == sorted(names)   # every tip exactly once


@pytest.mark.parametrize("dist_type", [1, 4, 5])
def test_dc_msa_big_cluster_and_other_models(gpu, orc, dist_type):
    """one cluster of ~150 members: several 64x64 (32x32) pair tiles per cluster, long BFS frontiers"""
    from dipper_amd import capi
    rng = np.random.default_rng(77 + dist_type)
    n, B, L = 520, 200, 1200
    seqs = _clone_heavy_alignment(rng, n, L, clones=150)
    gpu.set_msa(capi.pack4_many(seqs), L)
    gpu.dist_matrix(capi.SRC_MSA, dist_type)
    M = gpu.matrix()
    got = gpu.dc_run(capi.SRC_MSA, n, B, dist_type=dist_type)
    ref = orc.dc_run(M, B, skip_last_backbone=1)
    assert np.bincount(ref["cluster_id"][B:]).max() >= 100
    _same_dc_state(got, ref, n, B)


def test_dc_memory_groups(gpu, orc, monkeypatch):
    """budget 0: every cluster is its own memory group (the grouping loop of dc_cluster_phase)"""
    from dipper_amd import capi
    rng = np.random.default_rng(5)
    n, B, L = 300, 60, 800
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    gpu.set_msa(capi.pack4_many(seqs), L)
    gpu.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    M = gpu.matrix()
    monkeypatch.setenv("DPR_DC_BUDGET_MB", "0")
    got = gpu.dc_run(capi.SRC_MSA, n, B, dist_type=capi.DIST_JC)
    assert got["stats"]["groups"] == got["stats"]["clusters"] > 1
    _same_dc_state(got, orc.dc_run(M, B, skip_last_backbone=1), n, B)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dc_sharded_virtual_ranks(gpu, orc, world):
    """multi-GPU path of dpr_dc_run on one GPU: query shares per rank, clusters dealt to the ranks,
    states merged as old + sum(new - old) -- bit-identical to the single-rank run and to the oracle"""
    from dipper_amd import capi
    rng = np.random.default_rng(31)
    n, B, L = 900, 150, 1000
    seqs = _clone_heavy_alignment(rng, n, L, clones=60)
    gpu.set_msa(capi.pack4_many(seqs), L)
    gpu.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    M = gpu.matrix()
    ref = orc.dc_run(M, B, skip_last_backbone=1)
    got = gpu.dc_run(capi.SRC_MSA, n, B, dist_type=capi.DIST_JC, flags=capi.dc_virtual_ranks(world))
    _same_dc_state(got, ref, n, B)
    if world == 3:
        reads = _reads(rng, 300, 3000, 5000)
        gpu.set_reads(reads)
        gpu.sketch(k=15, S=1000, fetch=False)
        gpu.dist_matrix(capi.SRC_MASH, 0, 15)
        M = gpu.matrix()
        got = gpu.dc_run(capi.SRC_MASH, len(reads), 80, k=15, flags=capi.dc_virtual_ranks(world))
        _same_dc_state(got, orc.dc_run(M, 80, skip_last_backbone=0), len(reads), 80)


def test_dc_mash(gpu, orc):
    from dipper_amd import capi
    rng = np.random.default_rng(321)
    reads = _reads(rng, 420, 3000, 6000)
    n, B = len(reads), 90
    gpu.set_reads(reads)
    gpu.sketch(k=15, S=1000, fetch=False)
    gpu.dist_matrix(capi.SRC_MASH, 0, 15)
    M = gpu.matrix()
    got = gpu.dc_run(capi.SRC_MASH, n, B, k=15)
    ref = orc.dc_run(M, B, skip_last_backbone=0)
    _same_dc_state(got, ref, n, B)
    assert np.bincount(ref["cluster_id"][B:]).max() > 12       # more than one 12-row group in a cluster


def test_dc_mash_large_sketch_without_index_is_rejected(monkeypatch):
    """Sketch size above the lookup tables (1024) with the inverted index switched off and divergent reads: none of
    the three kernels that can write the query-minor distance block of the assignment phase is available.  The call
    must fail loudly (it used to fall through to the literal row kernel, which ignores the layout flag, and assign
    clusters from a mis-laid block without any error)."""
    import dipper_amd
    from dipper_amd import capi
    monkeypatch.setenv("DPR_MASH_KERNEL", "noindex")
    rng = np.random.default_rng(77)
    reads = [rng.choice(_util.BASES, size=2500).tobytes() for _ in range(300)]      # unrelated reads: ~S tokens per sketch
    d = dipper_amd.Dipper(0)
    try:
        d.set_reads(reads)
        d.sketch(k=15, S=2000, fetch=False)
        with pytest.raises(capi.DipperError) as ei:
            d.dc_run(capi.SRC_MASH, len(reads), 60, k=15)
        assert "transposed" in str(ei.value) or "sketch size" in str(ei.value)
    finally:
        d.close()


def test_dc_additive_recovery_at_size(gpu):
    """size-independent property: with the distance to the last backbone tip computed
    (DPR_DC_EXACT_LAST) a tree metric is recovered exactly; here through sequences, so check that
    every tip is placed once and the backbone + cluster bookkeeping adds up at a larger size"""
    from dipper_amd import capi
    rng = np.random.default_rng(9)
    n, B, L = 6000, 300, 1000
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    gpu.set_msa(capi.pack4_many(seqs), L)
    got = gpu.dc_run(capi.SRC_MSA, n, B, dist_type=capi.DIST_JC, flags=capi.DC_EXACT_LAST)
    cl = got["cluster_id"]
    assert np.all(cl[:B] == -1) and np.all(cl[B:] >= 0) and np.all(cl[B:] < 4 * B - 4)
    live = 4 * n - 4
    assert np.all(got["e"][:live] >= 0) and np.all(got["belong"][:live] >= 0)
    deg = np.bincount(got["belong"][:live], minlength=2 * n)
    assert np.all(deg[:n] == 1) and deg[n] == 2 and np.all(deg[n + 1:2 * n - 1] == 3)


def test_dc_rejects_matrix_source_and_bad_backbone(gpu):
    from dipper_amd import capi
    D = np.zeros((8, 8))
    gpu.set_matrix_full(D)
    with pytest.raises(capi.DipperError):
        gpu.dc_run(capi.SRC_MATRIX, 8, 4)
    seqs = _util.synth_alignment(np.random.default_rng(1), 8, 64)
    gpu.set_msa(capi.pack4_many(seqs), 64)
    with pytest.raises(capi.DipperError):
        gpu.dc_run(capi.SRC_MSA, 8, 2)
    with pytest.raises(capi.DipperError):
        gpu.dc_run(capi.SRC_MSA, 8, 8)
