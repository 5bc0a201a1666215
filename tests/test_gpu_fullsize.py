"""GPU tests at BASELINE.json's full sizes through size-independent properties (the oracle finishes only
small cases): configs[1] 30 000 aligned tips NJ, configs[2] 100 000 unaligned tips Mash + placement,
configs[3] 1 000 000 tips divide-and-conquer.  Set DPR_SKIP_FULLSIZE=1 to skip (about 2 minutes)."""
import os

import numpy as np
import pytest

from tests import _util

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


def test_config1_nj_30k_two_algorithms_agree():
    """30 000 tips x 1 000 sites, JC69: the exact pruned scan and the full streaming scan (the reference's
    algorithm) are independent implementations and must produce the same merge log bit for bit; the log
    is a valid NJ history (x < y < active size, finite branch lengths)."""
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


from tests.conftest import dirty_device_memory as _dirty_device_memory


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
