"""GPU tests at BASELINE.json's full sizes through size-independent properties (the oracle finishes only
small cases): configs[1] 30 000 aligned tips NJ, configs[2] 100 000 unaligned tips Mash + placement,
configs[3] 1 000 000 tips divide-and-conquer.  Set DPR_SKIP_FULLSIZE=1 to skip (about 2 minutes)."""
import os

import numpy as np
import pytest

from tests import _util
from tests.conftest import dirty_device_memory as _dirty_device_memory

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("DPR_SKIP_FULLSIZE") == "1", reason="DPR_SKIP_FULLSIZE=1")]


def _tree_degrees_ok(st, n):
    live = 4 * n - 4
    assert np.all(st["e"][:live] >= 0) and np.all(st["belong"][:live] >= 0)
    deg = np.bincount(st["belong"][:live], minlength=2 * n)
    assert np.all(deg[:n] == 1) and deg[n] == 2 and np.all(deg[n + 1:2 * n - 1] == 3)
    assert np.all(st["len"][:live] >= 0)
    # slot i and its reverse carry the same length and swapped end points
    src, dst = st["belong"][:live], st["e"][:live]
    key = src.astype(np.int64) * (2 * n) + dst
    rkey = dst.astype(np.int64) * (2 * n) + src
    order, rorder = np.argsort(key), np.argsort(rkey)
    assert np.array_equal(key[order], rkey[rorder])
    assert np.array_equal(st["len"][:live][order], st["len"][:live][rorder])


@pytest.mark.parametrize("fill", [0xFF, 0x40], ids=["memory_0xFF", "memory_0x40"])
def test_config1_nj_30k_two_algorithms_agree(fill):
    """30 000 tips x 1 000 sites, JC69: the exact pruned scan and the full streaming scan (the reference's
    algorithm) are independent implementations and must produce the same merge log bit for bit; the log
    is a valid NJ history (x < y < active size, finite branch lengths).
    Runs on POISONED device memory in the default suite (24 GB filled with 0xFF = NaN patterns, resp. 0x40, and freed
    right before the contexts allocate): round 2's two-kernel NJ had a cross-thread race that showed as one NaN branch
    length in 30 000 iterations and only under 0xFF."""
    import dipper_amd
    from dipper_amd import capi
    n, L = 30000, 1000
    seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-4, lo=2e-5, hi=2e-3)
    packed = capi.pack4_many(seqs)
    del seqs
    res = {}
    for mode in (1, 0):
        capi.set_nj_mode(mode)
        d = dipper_amd.Dipper(0)
        try:
            _dirty_device_memory(24 << 30, fill)
            d.set_msa(packed, L)
            d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
            res[mode] = d.nj_run()
        finally:
            d.close()
    capi.set_nj_mode(1)
    a, b = res[1], res[0]
    assert a["iters"] == b["iters"] == n - 2
    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
        assert np.array_equal(a[key], b[key]), key
    assert a["last_d"] == b["last_d"]
    active = n - np.arange(n - 2)
    assert np.all(a["merge_x"] < a["merge_y"]) and np.all(a["merge_y"] < active) and np.all(a["merge_x"] >= 0)
    assert np.all(np.isfinite(a["bl_x"])) and np.all(np.isfinite(a["bl_y"]))
    # src/neighborJoining.cu:227-231: after the two clamps at most one branch of a pair is negative
    assert not np.any((a["bl_x"] < 0) & (a["bl_y"] < 0))


def test_config2_mash_placement_100k_prefix_property():
    """100 000 unaligned tips, Mash sketches + k-closest placement: the decision for tip i depends only on
    tips < i, so the per-tip (edge, split position, pendant length) of the first 20 000 tips equals a
    separate 20 000-tip run bit for bit; the final structure is a binary tree over all tips."""
    import dipper_amd
    from dipper_amd import capi
    n, m, L = 100000, 20000, 3000
    seqs = _util.synth_alignment(np.random.default_rng(2), n, L, mean_bl=1e-3, lo=1e-4, hi=1e-2)
    d = dipper_amd.Dipper(0)
    try:
        d.set_reads(seqs)
        d.sketch(15, 1000, fetch=False)
        full = d.place_run(capi.SRC_MASH, n, k=15)
        d.set_reads(seqs[:m])
        d.sketch(15, 1000, fetch=False)
        part = d.place_run(capi.SRC_MASH, m, k=15)
    finally:
        d.close()
    _tree_degrees_ok(full, n)
    _tree_degrees_ok(part, m)
    # slot ids and lengths are the same objects in both runs; only node ids are offset by the tip count
    assert np.array_equal(full["trace"][2:m], part["trace"][2:m])


def test_config3_divide_and_conquer_1m():
    """1 000 000 aligned tips x 400 sites, backbone 50 000: binary tree over all tips; every query sits in
    an eligible backbone slot; cluster ids and the backbone trace do not depend on the later tips
    (same backbone, first 150 000 tips only)."""
    import dipper_amd
    from dipper_amd import capi
    n, B, L, m = 1000000, 50000, 400, 150000
    seqs = _util.synth_alignment(np.random.default_rng(3), n, L, mean_bl=2e-3, lo=2e-4, hi=2e-2)
    seqs = [seqs[i] for i in np.random.default_rng(4).permutation(n)]
    packed = capi.pack4_many(seqs)
    del seqs
    d = dipper_amd.Dipper(0)
    try:
        d.set_msa(packed, L)
        full = d.dc_run(capi.SRC_MSA, n, B, dist_type=capi.DIST_JC)
        d.set_msa(packed[:m], L)
        part = d.dc_run(capi.SRC_MSA, m, B, dist_type=capi.DIST_JC)
    finally:
        d.close()
    _tree_degrees_ok(full, n)
    cl = full["cluster_id"]
    assert np.all(cl[:B] == -1) and np.all(cl[B:] >= 0) and np.all(cl[B:] < 4 * B - 4)
    assert np.array_equal(cl[:m], part["cluster_id"])
    assert np.array_equal(full["trace"][2:B], part["trace"][2:B])
    sizes = np.bincount(cl[B:])
    assert sizes.sum() == n - B and sizes.max() < B
    assert full["stats"]["clusters"] == int((sizes > 0).sum())


def _renumber_nodes(st, n_from, n_to):
    """Adjacency of an imported backbone built for n_from tips -> the same tree for n_to tips: internal node ids start
    at the tip count (src/tree.cpp:216-361 with another totalLeaves); slots, lengths and leaf ids are unchanged."""
    out = {}
    for key in ("e", "belong"):
        a = st[key].copy()
        a[a >= n_from] += n_to - n_from
        out[key] = np.full(8 * n_to, -1, np.int32)
        k = min(len(a), 8 * n_to)
        out[key][:k] = a[:k]
    for key, fill, dt in (("nxt", -1, np.int32), ("len", 2.0, np.float64)):
        out[key] = np.full(8 * n_to, fill, dt)
        k = min(len(st[key]), 8 * n_to)
        out[key][:k] = st[key][:k]
    head = np.full(2 * n_to, -1, np.int32)
    m_leaves = min(n_from, n_to)
    head[:m_leaves] = st["head"][:m_leaves]
    internal = st["head"][n_from:2 * n_from]
    k = min(len(internal), n_to)
    head[n_to:n_to + k] = internal[:k]
    out["head"] = head
    return out


def test_config4_add_50k_onto_500k(orc):
    """BASELINE configs[4] at its workload on one GPU (aligned input, 300 sites): --add of 50 000 queries onto a
    500 000-tip backbone (initializeDeviceArrays + addQuery, src/placement_close_k.cu:126-264,858-990).
    Size-independent properties: (a) the closest lists the import builds (parallel relaxation rounds) equal the oracle's
    leaf-by-leaf order on ALL 2 000 000 slots; (b) the result is a binary tree over all 550 000 tips; (c) the trace of
    the first 10 000 queries (edge, split position, pendant length) does not depend on the 40 000 later ones."""
    import sys
    import dipper_amd
    from dipper_amd import capi
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 100000))
    m, nq, q1, L = 500000, 50000, 10000, 300
    n = m + nq
    seqs = _util.synth_alignment(np.random.default_rng(5), n, L, mean_bl=2e-3, lo=2e-4, hi=2e-2)
    seqs = [seqs[i] for i in np.random.default_rng(6).permutation(n)]
    d = dipper_amd.Dipper(0)
    try:
        # backbone tree of the first m tips (divide-and-conquer run), written and re-imported like the CLI's -t file
        d.set_msa(capi.pack4_many(seqs[:m]), L)
        bb = d.dc_run(capi.SRC_MSA, m, m // 20, dist_type=capi.DIST_JC)
        names = ["T%d" % i for i in range(n)]
        nwk = _util.newick_from_placement(names[:m], bb["head"], bb["e"], bb["nxt"], bb["len"], m)
        del bb
        st, leaf_names = _util.backbone_state(orc, nwk, n)          # Tree::Tree ids + adjacency
        order = [int(x[1:]) for x in leaf_names]                     # backbone tips in import order, then the queries
        packed = capi.pack4_many([seqs[i] for i in order] + seqs[m:])
        del seqs
        adj = ("head", "e", "nxt", "belong", "len")
        # (a) the import alone (no query): closest lists vs the oracle's serial order
        st_m = _renumber_nodes(st, n, m)
        d.set_msa(packed[:m], L)
        got = d.place_run(capi.SRC_MSA, m, first=m, dist_type=capi.DIST_JC, state={k: st_m[k].copy() for k in adj})
        ref = {k: st_m[k].copy() for k in adj}
        ref["cid"] = np.full(40 * m, -1, np.int32)
        ref["cdis"] = np.full(40 * m, 2.0, np.float64)
        orc.place_init_lists(m, m, ref)
        live = 4 * m - 4
        assert np.array_equal(got["cid"][:5 * live], ref["cid"][:5 * live])
        assert np.array_equal(got["cdis"][:5 * live], ref["cdis"][:5 * live])
        for key in adj:
            assert np.array_equal(got[key][:(2 * m if key == "head" else live)], st_m[key][:(2 * m if key == "head" else live)]), key
        del got, ref
        # (b) all queries
        d.set_msa(packed, L)
        full = d.place_run(capi.SRC_MSA, n, first=m, dist_type=capi.DIST_JC, state={k: st[k].copy() for k in adj})
        _tree_degrees_ok(full, n)
        tr = full["trace"][m:]
        assert np.all(tr[:, 0] >= 0) and np.all(tr[:, 0] < 4 * n - 4) and np.all(tr[:, 1] >= 0) and np.all(tr[:, 2] >= 0)
        # (c) prefix property
        n2 = m + q1
        st2 = _renumber_nodes(st, n, n2)
        d.set_msa(packed[:n2], L)
        part = d.place_run(capi.SRC_MSA, n2, first=m, dist_type=capi.DIST_JC, state={k: st2[k].copy() for k in adj})
        assert np.array_equal(full["trace"][m:n2], part["trace"][m:n2])
    finally:
        d.close()


def test_config1_matrix_build_is_stream_ordered():
    """Regression: the 7 GB zero fill of a fresh 30 000-tip matrix ran on the null stream, which the context's
    non-blocking stream does not wait for, and wiped distance tiles that had already been written (zero blocks
    in rows 2 000-2 700, hence different trees from run to run).  Three builds -- on memory that held zeros, NaN
    patterns and 0x40 patterns before -- must give the same row sums bit for bit, and no row sum may be NaN."""
    import dipper_amd
    from dipper_amd import capi
    n, L = 30000, 1000
    seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-4, lo=2e-5, hi=2e-3)
    packed = capi.pack4_many(seqs)
    del seqs
    capi.set_nj_mode(0)
    try:
        sums = []
        for fill in (0x00, 0xFF, 0x40):
            d = dipper_amd.Dipper(0)          # (the context first: it loads the HIP runtime)
            try:
                _dirty_device_memory(24 << 30, fill)
                d.set_msa(packed, L)
                d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
                sums.append(d.row_sums().view(np.uint64).copy())
            finally:
                d.close()
    finally:
        capi.set_nj_mode(1)
    assert not np.any(np.isnan(sums[0].view(np.float64)))
    assert np.array_equal(sums[0], sums[1]) and np.array_equal(sums[0], sums[2])


def test_config1_pruned_nj_repeats_with_varied_launch_shapes(monkeypatch):
    """Ten pruned NJ runs of the 30 000-tip input with the scan grid, the graph length and the memory poison varied: every
    merge log digest must equal the first one.  The design depends on intra-launch ordering between update blocks and test
    blocks of the post kernel (njp.hip); a race there shows as a run-to-run difference under some launch shape."""
    import hashlib
    import dipper_amd
    from dipper_amd import capi
    n, L = 30000, 1000
    seqs = _util.synth_alignment(np.random.default_rng(1), n, L, mean_bl=2e-4, lo=2e-5, hi=2e-3)
    packed = capi.pack4_many(seqs)
    del seqs
    shapes = [(256, 32, None), (64, 8, 0xFF), (1024, 32, 0x40), (128, 128, 0xFF), (512, 16, None), (256, 32, 0xFF),
              (1024, 8, 0xFF), (64, 64, 0x40), (256, 1, 0xFF), (37, 32, 0xFF)]
    digests = []
    for grid, giters, fill in shapes:
        monkeypatch.setenv("DPR_NJP_GRID", str(grid))
        monkeypatch.setenv("DPR_NJ_GRAPH_ITERS", str(giters))
        d = dipper_amd.Dipper(0)
        try:
            if fill is not None:
                _dirty_device_memory(20 << 30, fill)
            d.set_nj_mode(1)
            d.set_msa(packed, L)
            d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
            res = d.nj_run()
        finally:
            d.close()
        assert res["iters"] == n - 2
        assert np.all(np.isfinite(res["bl_x"])) and np.all(np.isfinite(res["bl_y"])), (grid, giters, fill)
        h = hashlib.sha256()
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            h.update(np.ascontiguousarray(res[key]).tobytes())
        h.update(np.float64(res["last_d"]).tobytes())
        digests.append(h.hexdigest())
    assert len(set(digests)) == 1, digests


@pytest.mark.parametrize("source", ["msa", "mash"])
def test_placement_on_poisoned_memory(source):
    """k-closest placement (20 000 tips; the multi-tip path forced for the second half through DPR_PLACE_MULTI_MIN) on clean,
    0xFF- and 0x40-poisoned device memory: adjacency, lengths, closest lists and trace must be identical bit for bit."""
    import dipper_amd
    from dipper_amd import capi
    n = 20000
    rng = np.random.default_rng(21)
    if source == "msa":
        L = 600
        seqs = _util.synth_alignment(rng, n, L, mean_bl=2e-3, lo=2e-4, hi=2e-2)
        packed = capi.pack4_many(seqs)
    else:
        seqs = _util.synth_reads(rng, n, 1500, mean_bl=2e-3, lo=2e-4, hi=2e-2)
    os.environ["DPR_PLACE_MULTI_MIN"] = "10000"
    try:
        outs = []
        for fill in (None, 0xFF, 0x40):
            d = dipper_amd.Dipper(0)
            try:
                if fill is not None:
                    _dirty_device_memory(16 << 30, fill)
                if source == "msa":
                    d.set_msa(packed, L)
                    st = d.place_run(capi.SRC_MSA, n, dist_type=capi.DIST_JC)
                else:
                    d.set_reads(seqs)
                    d.sketch(15, 1000, fetch=False)
                    st = d.place_run(capi.SRC_MASH, n, k=15)
            finally:
                d.close()
            outs.append(st)
        live = 4 * n - 4
        for other in outs[1:]:
            for key, m in (("head", 2 * n), ("e", live), ("nxt", live), ("belong", live), ("len", live), ("cid", 5 * live), ("cdis", 5 * live)):
                assert np.array_equal(outs[0][key][:m].view(np.uint8), other[key][:m].view(np.uint8)), key
            assert np.array_equal(outs[0]["trace"].view(np.uint64), other["trace"].view(np.uint64))
        _tree_degrees_ok(outs[0], n)
    finally:
        os.environ.pop("DPR_PLACE_MULTI_MIN", None)


def test_config4_add_50k_onto_500k_mash_source(tmp_path, orc):
    """BASELINE configs[4] through its OWN distance path (unaligned reads, Mash sketches): --add of 50 000 queries onto a
    500 000-tip backbone.  Size-independent properties: the result is a binary tree over all 550 000 tips; every query's
    (edge, split position, pendant length) is valid; the trace of the first 10 000 queries does not depend on the 40 000
    later ones."""
    import sys
    import dipper_amd
    from dipper_amd import capi
    if not os.path.exists(_util.GEN_SYNTH):
        pytest.skip("tools not built")
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 100000))
    m, nq, q1 = 500000, 50000, 10000
    n = m + nq
    inp = _util.gen_synth(tmp_path, "r", n, 3000, 9, 1e-3, 1e-4, 1e-2, reads=True, shuffle=6)
    reads = inp["reads"]
    d = dipper_amd.Dipper(0)
    try:
        d.set_reads_packed(*_util.reads_prefix(reads, m))
        d.sketch(15, 1000, fetch=False)
        bb = d.dc_run(capi.SRC_MASH, m, m // 20, k=15)
        names = ["T%d" % i for i in range(n)]
        nwk = _util.newick_from_placement(names[:m], bb["head"], bb["e"], bb["nxt"], bb["len"], m)
        del bb
        st, leaf_names = _util.backbone_state(orc, nwk, n)
        order = [int(x[1:]) for x in leaf_names] + list(range(m, n))
        reads2 = _util.reads_reorder(reads, order)
        adj = ("head", "e", "nxt", "belong", "len")
        d.set_reads_packed(*reads2)
        d.sketch(15, 1000, fetch=False)
        full = d.place_run(capi.SRC_MASH, n, first=m, k=15, state={k: st[k].copy() for k in adj})
        _tree_degrees_ok(full, n)
        tr = full["trace"][m:]
        assert np.all(tr[:, 0] >= 0) and np.all(tr[:, 0] < 4 * n - 4) and np.all(tr[:, 1] >= 0) and np.all(tr[:, 2] >= 0)
        n2 = m + q1
        st2 = _renumber_nodes(st, n, n2)
        d.set_reads_packed(*_util.reads_prefix(reads2, n2))
        d.sketch(15, 1000, fetch=False)
        part = d.place_run(capi.SRC_MASH, n2, first=m, k=15, state={k: st2[k].copy() for k in adj})
        assert np.array_equal(full["trace"][m:n2], part["trace"][m:n2])
    finally:
        d.close()


@pytest.mark.timeout(900)
def test_exact_placement_run_that_crosses_the_top_tree_schedules(monkeypatch):
    """36 000 tips in random order: the top tree (nodes above the one-wavefront subtrees) grows past the 2 048 nodes the climbing
    schedule of px_top_kernel keeps in LDS, so ONE run uses the climbing schedule first and the polling one after (exact.hip:
    px_top_climb / px_top_poll; the structural records stop being packed at the same point).  The oracle needs hours at this
    size; the bar is the level-by-level schedule of rounds 3-5 (DPR_EXACT_TOP_LEVELS=1), which the smaller cases pin to the
    oracle: every array of the result bit for bit."""
    import dipper_amd
    from dipper_amd import capi
    n, L = 36000, 400
    seqs = _util.synth_alignment(np.random.default_rng(36), n, L, mean_bl=2e-3, lo=2e-4, hi=2e-2)
    perm = np.random.default_rng(7).permutation(n)
    packed = capi.pack4_many([seqs[i] for i in perm])
    del seqs
    res = {}
    for tag in ("default", "levels"):
        if tag == "levels":
            monkeypatch.setenv("DPR_EXACT_TOP_LEVELS", "1")
        d = dipper_amd.Dipper(0)
        try:
            d.set_msa(packed, L)
            res[tag] = d.place_exact_run(capi.SRC_MSA, n, dist_type=capi.DIST_JC)
        finally:
            d.close()
    a, b = res["default"], res["levels"]
    nodes, live = 2 * n - 1, 4 * n - 4
    assert np.array_equal(a["trace"][2:], b["trace"][2:])
    for key in ("head", "e", "nxt", "belong", "len", "rev", "dep"):
        m = nodes if key in ("head", "dep") else live
        assert np.array_equal(a[key][:m], b[key][:m]), key
    # the run did cross: nodes whose subtree has more than 64 nodes, counted on the final tree (parent = the neighbour one level up)
    dep = a["dep"][:nodes].astype(np.int64)
    src, dst = a["belong"][:live].astype(np.int64), a["e"][:live].astype(np.int64)
    up = dep[dst] < dep[src]
    parent = np.full(nodes, -1, dtype=np.int64)
    parent[src[up]] = dst[up]
    size = np.ones(nodes, dtype=np.int64)
    for v in np.argsort(-dep, kind="stable"):
        if parent[v] >= 0:
            size[parent[v]] += size[v]
    assert size.max() == nodes and int((size > 64).sum()) > 2048


@pytest.mark.timeout(900)
def test_exact_placement_on_poisoned_memory():
    """exact placement (12 000 tips, the small subtrees switch from one wavefront to one workgroup on the way) on clean, 0xFF- and
    0x40-poisoned device memory: records nobody wrote (the spare LDS record's global twin, pad fields, partials of workgroups
    without a subtree) must not reach a result -- adjacency, lengths, depths and trace identical bit for bit."""
    import dipper_amd
    from dipper_amd import capi
    n, L = 12000, 500
    seqs = _util.synth_alignment(np.random.default_rng(12), n, L, mean_bl=2e-3, lo=2e-4, hi=2e-2)
    perm = np.random.default_rng(5).permutation(n)
    packed = capi.pack4_many([seqs[i] for i in perm])
    outs = []
    for fill in (None, 0xFF, 0x40):
        d = dipper_amd.Dipper(0)
        try:
            if fill is not None:
                _dirty_device_memory(8 << 30, fill)
            d.set_msa(packed, L)
            outs.append(d.place_exact_run(capi.SRC_MSA, n, dist_type=capi.DIST_JC))
        finally:
            d.close()
    nodes, live = 2 * n - 1, 4 * n - 4
    for other in outs[1:]:
        assert np.array_equal(outs[0]["trace"][2:].view(np.uint64), other["trace"][2:].view(np.uint64))
        for key in ("head", "e", "nxt", "belong", "len", "rev", "dep"):
            m = nodes if key in ("head", "dep") else live
            assert np.array_equal(outs[0][key][:m].view(np.uint8), other[key][:m].view(np.uint8)), key
