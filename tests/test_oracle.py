"""CPU tests: the oracle against the reference-derived known answers, public
vectors, hand-computed cases and size-independent properties."""
import json
import math
import os

import numpy as np
import pytest

from tests import _util

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KA = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))


def test_murmur_known_answers(orc):
    for v in KA["murmur3_x64_128"]:
        h1, h2 = orc.murmur(v["data"].encode(), v["seed"])
        assert h1 == int(v["h1"], 16)
        if "h2" in v:
            assert h2 == int(v["h2"], 16)


def test_mash_dist_known_answers(orc):
    i = np.arange(1000, dtype=np.uint64)
    cases = [
        (2 * i + 10, 2 * i + 10, 0.0),
        (2 * i + 10, np.where(i < 500, 2 * i + 10, 2 * i + 11), 0.027031007207210963),
        (7 + i, np.full(1000, 7, dtype=np.uint64), 0.41437383991701832),
        (np.full(1000, 7, dtype=np.uint64), 7 + i, 0.0),
    ]
    for row, col, d in cases:
        # A = column (lower index, outer list), B = row (inner list)
        assert orc.mash_dist(col.astype(np.uint64), row.astype(np.uint64), 15) == pytest.approx(d, rel=0, abs=1e-17)


def test_pack_encoders(orc):
    s = b"ACGTUNacgt-XACGTACGTACGTAC"
    w4 = orc.pack4(s)
    codes = [(int(w4[i // 16]) >> (4 * (i % 16))) & 15 for i in range(len(s))]
    assert codes == [0, 1, 2, 3, 3, 4, 4, 4, 4, 4, 4, 4] + [0, 1, 2, 3] * 3 + [0, 1]
    assert int(w4[1]) >> (4 * (len(s) - 16)) == 0          # tail of the last word is zero
    w2 = orc.pack2(s)
    codes2 = [(int(w2[i // 32]) >> (2 * (i % 32))) & 3 for i in range(len(s))]
    assert codes2 == [0, 1, 2, 3, 3, 0, 0, 0, 0, 0, 0, 0] + [0, 1, 2, 3] * 3 + [0, 1]
    assert orc.pack4(b"").size == 0 and orc.pack2(b"").size == 0


def test_kmer_hash_canonical(orc):
    # canonical = lexicographically smaller of forward / reverse complement, forward on ties
    for kmer in (b"ACGTACGTACGTACG", b"TTTTTTTTTTTTTTT", b"GATTACAGATTACAG", b"AC", b"GT"):
        k = len(kmer)
        comp = bytes({65: 84, 67: 71, 71: 67, 84: 65}[c] for c in reversed(kmer))
        canon = kmer if kmer <= comp else comp
        assert orc.lib.orc_kmer_hash(orc.pack2(kmer).ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_uint64)), 0, k) == orc.murmur(canon, 42)[0]


def test_sketch_keeps_duplicates_and_pads(orc):
    seq = b"ACGT" * 10  # 40 bases, k=15 -> 26 k-mers with only 4 distinct canonical windows
    sk = orc.sketch(orc.pack2(seq), len(seq), k=15, S=1000)
    assert np.all(sk[:26] != np.uint64(2**64 - 1)) and np.all(sk[26:] == np.uint64(2**64 - 1))
    assert np.all(np.diff(sk[:26].astype(np.float64)) >= 0)
    assert len(set(sk[:26].tolist())) < 26          # duplicates kept
    assert np.all(orc.sketch(orc.pack2(b"ACG"), 3) == np.uint64(2**64 - 1))  # len < k


def test_jc_hand_computed(orc):
    a = b"ACGTACGTACGTACGTACGT"          # 20 sites
    b = b"ACGTACGTACGTACGTTTTT"          # 3 mismatches (positions 16,17,18: A->T,C->T,G->T)
    c = b"ACGTACGT--NNACGTACGT"          # 4 invalid in c
    P = orc.pack4_many([a, b, c])
    u, m = orc.msa_counts(P, 20)
    assert (u[1, 0], m[1, 0]) == (20, 17)
    assert (u[2, 0], m[2, 0]) == (20, 16)     # useful counts sites where EITHER is valid
    D1 = orc.msa_dist_lower(P, 20, 1)
    D2 = orc.msa_dist_lower(P, 20, 2)
    assert D1[1, 0] == 1 - 17 / 20
    assert D2[1, 0] == pytest.approx(-0.75 * math.log(1 - (3 / 20) / 0.75), rel=1e-15)
    assert D1[2, 0] == 1 - 16 / 20
    # all-invalid pair: 0/0 -> NaN
    Q = orc.pack4_many([b"NNNN", b"----"])
    assert np.isnan(orc.msa_dist_lower(Q, 4, 1)[1, 0])


@pytest.mark.parametrize("n", [3, 5, 12, 40, 130, 300])
def test_nj_recovers_additive_tree(orc, n):
    rng = np.random.default_rng(n)
    D = _util.random_additive_matrix(rng, n)
    r = orc.nj_run(np.tril(D, -1))
    assert r["iters"] == n - 2
    names = [f"T{i}" for i in range(n)]
    nw = _util.newick_from_merges(names, r["merge_x"], r["merge_y"], r["bl_x"], r["bl_y"], r["last_d"], fmt=repr)
    assert np.abs(_util.patristic(nw, names) - D).max() < 1e-12 * n


@pytest.mark.parametrize("n", [5, 40, 255, 256, 257, 300])
def test_argmin_key_matches_literal_kernel_emulation(orc, n):
    """The (band(i), j mod 256, j, i) key is OUR derivation of the reference's tie-breaking; pin it
    against a literal emulation of findMinDist<<<256,256>>> + first-occurrence min_element on
    tie-heavy matrices."""
    rng = np.random.default_rng(1000 + n)
    for trial in range(3):
        D = rng.integers(1, 3, size=(n, n)).astype(np.float64)
        D = np.tril(D, -1) + np.tril(D, -1).T
        U = orc.row_sums(np.ascontiguousarray(D))
        x, y, q = _util.ref_findmin_emulation(D, U, n)
        rc, i, j, qq = orc.nj_argmin(np.ascontiguousarray(D), n, U)
        assert rc == 0 and (i, j, qq) == (x, y, q)


def test_q_exactly_10000_is_no_candidate(orc):
    """The reference keeps a candidate only if `temp < minD` with minD initialised to 10000 (src/neighborJoining.cu:134-141):
    when the smallest Q of the matrix is EXACTLY 10000.0 no thread records it, every tuple stays (0, 0, 10000) and
    min_element returns the first of them -- the undefined (0, 0) merge.  n = 3 with d01 = -5000, d02 = d12 = -2500:
    q = -(d01 + d02 + d12) = 10000 for every pair, all arithmetic exact.  The oracle must report "no candidate" like the
    literal emulation (round 5's verdict: it accepted q == 10000.0); at 9999.5 it must find the pair."""
    D = np.array([[0.0, -5000.0, -2500.0], [-5000.0, 0.0, -2500.0], [-2500.0, -2500.0, 0.0]])
    U = orc.row_sums(np.ascontiguousarray(D))
    assert _util.ref_findmin_emulation(D, U, 3) == (0, 0, 10000.0)
    rc, _, _, _ = orc.nj_argmin(np.ascontiguousarray(D), 3, U)
    assert rc != 0
    res = orc.nj_run(np.tril(D, -1))
    assert res["iters"] == -1 and res["done"] == 0
    D2 = D.copy()
    D2[0, 1] = D2[1, 0] = -4999.5      # every q = 9999.5, exactly: candidates again
    U2 = orc.row_sums(np.ascontiguousarray(D2))
    x, y, q = _util.ref_findmin_emulation(D2, U2, 3)
    rc, i, j, qq = orc.nj_argmin(np.ascontiguousarray(D2), 3, U2)
    assert q < 10000.0 and rc == 0 and (i, j, qq) == (x, y, q)


def test_nj_threads_do_not_change_result(orc):
    rng = np.random.default_rng(3)
    D = _util.random_additive_matrix(rng, 200, zero_frac=0.4)
    a = orc.nj_run(np.tril(D, -1), threads=1)
    b = orc.nj_run(np.tril(D, -1), threads=4)
    for k in ("merge_x", "merge_y", "bl_x", "bl_y"):
        assert np.array_equal(a[k], b[k])


def test_row_sums_invariant_after_updates(orc):
    """U stays the active row sums after every update (last row)."""
    rng = np.random.default_rng(11)
    n = 60
    D = _util.random_additive_matrix(rng, n)
    for k in (1, 7, 30):
        r = orc.nj_run(np.tril(D, -1), max_iters=k)
        na = n - k
        M = r["D"][:na, :na]
        assert np.allclose(M.sum(axis=1), r["U"][:na], rtol=1e-9)
        assert np.array_equal(M, M.T)


@pytest.mark.parametrize("n", [8, 60, 200])
def test_placement_recovers_additive_tree(orc, n):
    """the reference's CPU k-closest placement recovers additive inputs to
    1e-15; same property for the restatement, 4N-4 directed edges, node N keeps degree 2."""
    rng = np.random.default_rng(50 + n)
    D = _util.random_additive_matrix(rng, n)
    D *= 0.9 / D.max()      # the algorithm's sentinels (list init 2, ineligible add 2) assume d <= 1
    st = orc.place_run(D)
    assert st["next_slot"] == 4 * n - 4
    deg = 0
    i = st["head"][n]
    while i != -1:
        deg += 1
        i = st["nxt"][i]
    assert deg == 2
    names = [f"T{i}" for i in range(n)]
    nw = _util.newick_from_placement(names, st["head"], st["e"], st["nxt"], st["len"], n, fmt=repr)
    assert np.abs(_util.patristic(nw, names) - D).max() < 1e-12 * n


@pytest.mark.parametrize("n,B", [(60, 25), (400, 100), (1000, 150)])
def test_dc_recovers_additive_tree(orc, n, B):
    """divide-and-conquer restatement (orc_dc_run): with every query-to-backbone distance computed
    (the Mash twin's behaviour) a tree metric is recovered exactly; every tip is placed once; clusters
    are eligible backbone slots."""
    rng = np.random.default_rng(n)
    D = _util.random_additive_matrix(rng, n)
    D *= 0.9 / D.max()
    st = orc.dc_run(D, B, skip_last_backbone=0)
    assert st["next_slot"] == 4 * n - 4
    cl = st["cluster_id"]
    assert np.all(cl[:B] == -1) and np.all(cl[B:] >= 0) and np.all(cl[B:] < 4 * B - 4)
    names = [f"T{i}" for i in range(n)]
    nw = _util.newick_from_placement(names, st["head"], st["e"], st["nxt"], st["len"], n, fmt=repr)
    assert np.abs(_util.patristic(nw, names) - D).max() < 1e-12 * n
    # the reference's aligned-input defect (distance to the last backbone tip never written) only moves
    # queries whose true edge is next to that tip: same tip set, valid tree
    st2 = orc.dc_run(D, B, skip_last_backbone=1)
    assert st2["next_slot"] == 4 * n - 4
    nw2 = _util.newick_from_placement(names, st2["head"], st2["e"], st2["nxt"], st2["len"], n, fmt=repr)
    assert sorted(_util.parse_newick(nw2)[2].values()) == sorted(names)


def test_dc_rejects_cluster_as_large_as_backbone(orc):
    """src/divide_and_conquer/placement_close_k.cu:1339-1346: exit above B, endless loop at B"""
    rng = np.random.default_rng(60)
    D = _util.random_additive_matrix(rng, 60)
    D *= 0.9 / D.max()
    assert orc.dc_run(D, 10)["next_slot"] == -2


@pytest.mark.parametrize("n", [8, 60, 300])
def test_exact_placement_recovers_additive_tree_and_depths(orc, n):
    """exact placement restatement (orc_place_exact_run): additive input recovered; the incrementally
    patched depths (DFS-rank scheme of src/placement.cu:366-416) equal the true depths below node N."""
    rng = np.random.default_rng(70 + n)
    D = _util.random_additive_matrix(rng, n, zero_frac=0.2)
    D *= 0.9 / D.max()
    st = orc.place_exact_run(D)
    assert st["next_slot"] == 4 * n - 4
    names = [f"T{i}" for i in range(n)]
    nw = _util.newick_from_placement(names, st["head"], st["e"], st["nxt"], st["len"], n, fmt=repr)
    assert np.abs(_util.patristic(nw, names) - D).max() < 1e-12 * n
    dep = {n: 0}
    todo = [n]
    while todo:
        v = todo.pop()
        i = st["head"][v]
        while i != -1:
            w = int(st["e"][i])
            if w not in dep:
                dep[w] = dep[v] + 1
                todo.append(w)
            i = st["nxt"][i]
    assert len(dep) == 2 * n - 1
    assert all(st["dep"][v] == d for v, d in dep.items())
    live = 4 * n - 4
    assert np.array_equal(st["rev"][st["rev"][:live]], np.arange(live))


def test_phylip_value_is_float_rounded(orc):
    assert orc.phylip_value("0.1") == float(np.float32(0.1))
    assert orc.phylip_value("1e-3") == float(np.float32(1e-3))


@pytest.mark.parametrize("n", [8, 200, 1500])
def test_rapidnj_style_baseline_is_exact_nj(orc, n):
    """oracle/rapidnj_baseline.c is only the CPU baseline bench.py times, but it must be a correct NJ: on an
    additive matrix it recovers the tree and produces the splits of the oracle's NJ."""
    rng = np.random.default_rng(n)
    D = _util.random_additive_matrix(rng, n)
    r = orc.rapidnj_run(D, threads=4)
    assert r["joins"] == n - 2
    names = [f"T{i}" for i in range(n)]
    sub = {i: names[i] for i in range(n)}
    for t in range(r["joins"]):
        a, b = int(r["child_a"][t]), int(r["child_b"][t])
        sub[n + t] = "(%s:%r,%s:%r)" % (sub.pop(a), float(r["bl_a"][t]), sub.pop(b), float(r["bl_b"][t]))
    p0, p1 = int(r["last_pair"][0]), int(r["last_pair"][1])
    nw = "(%s:%r,%s:%r);" % (sub[p0], r["last_d"] / 2, sub[p1], r["last_d"] / 2)
    assert np.abs(_util.patristic(nw, names) - D).max() < 1e-9
    ref = orc.nj_run(np.tril(D, -1))
    nw2 = _util.newick_from_merges(names, ref["merge_x"], ref["merge_y"], ref["bl_x"], ref["bl_y"], ref["last_d"], fmt=repr)
    assert _util.splits(nw, names) == _util.splits(nw2, names)
