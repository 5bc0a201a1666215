"""GPU parity of the Mash path (a-8, a-9) and k-closest placement (a-10) through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from tests import _util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import dipper_amd
    d = dipper_amd.Dipper(0)
    yield d
    d.close()


def _reads(rng, n, lo, hi, related=True):
    base = rng.integers(0, 4, size=hi, dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = base[:L].copy() if related else rng.integers(0, 4, size=L, dtype=np.uint8)
        k = rng.poisson(L * 0.03)
        if k:
            pos = rng.integers(0, L, size=k)
            s[pos] = rng.integers(0, 4, size=k, dtype=np.uint8)
        b = _util.BASES[s].copy()
        if i % 7 == 3 and L > 10:
            b[rng.integers(0, L, size=3)] = ord("N")       # 2-bit packing maps these to 'A'
        out.append(b.tobytes())
    return out


@pytest.mark.parametrize("k", [15, 8, 2])
def test_kmer_hashes_and_sketches(gpu, orc, k):
    rng = np.random.default_rng(k)
    seqs = _reads(rng, 12, 200, 3000)
    seqs += [b"ACG", b"ACGTACGTACGTACG", b"ACGT" * 10, b"A" * 100, _reads(rng, 1, 20000, 20000)[0]]  # short, == k, duplicates, >1 chunk
    gpu.set_reads(seqs)
    for q, s in enumerate(seqs):
        h = gpu.kmer_hashes(q, k)
        p2 = orc.pack2(s)
        ref = np.array([orc.lib.orc_kmer_hash(p2.ctypes.data_as(C.POINTER(C.c_uint64)), p, k) for p in range(max(len(s) - k + 1, 0))], dtype=np.uint64)
        assert np.array_equal(h, ref), (q, k)
    sk = gpu.sketch(k=k, S=1000)
    for q, s in enumerate(seqs):
        assert np.array_equal(sk[q], orc.sketch(orc.pack2(s), len(s), k=k, S=1000)), (q, k)


def test_mash_dist_matrix_and_nj(gpu, orc):
    from dipper_amd import capi
    rng = np.random.default_rng(5)
    seqs = _reads(rng, 70, 1500, 4000) + [b"ACGT" * 300, b"ACGT" * 280 + b"TTGACC" * 20, b"AC", b"G" * 14]
    n = len(seqs)
    gpu.set_reads(seqs)
    sk = gpu.sketch(k=15, S=1000)
    gpu.dist_matrix(capi.SRC_MASH, 0, 15)
    M = gpu.matrix()
    assert np.array_equal(M, M.T) and np.all(np.diag(M) == 0)
    for i in range(n):
        ref = orc.mash_dist_row(sk, 15, i, i)
        assert np.allclose(M[i, :i], ref, rtol=1e-12, atol=0), i
    ref = orc.nj_run(np.tril(M, -1))
    res = gpu.nj_run()
    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
        assert np.array_equal(res[key], ref[key]), key


@pytest.mark.parametrize("kernel", ["tokens", "index", "table"])
@pytest.mark.parametrize("S,k", [(1000, 15), (64, 15), (257, 8), (2000, 15), (5, 4)])
def test_mash_dist_token_kernel_cases(orc, monkeypatch, S, k, kernel):
    """The three pair kernels -- run-encoded tokens (every sketch as runs of a reference list + literals), inverted
    index (one wavefront per row x 512 columns, posting lists in value order), bucket tables -- against the oracle's literal
    loop on the cases their rules have to get right: near-identical reads (long runs, a handful of literals), exact copies,
    reads with repeats (duplicate values: extra copies behind a run, in the reference itself, as literals), short reads
    (padding values = duplicates of the largest value), unrelated reads (all literals) and a reference (sketch 0) that is
    itself an outlier.  Integer intersection counts are exact, so the distances agree to the last bits of log()."""
    import dipper_amd
    from dipper_amd import capi
    # (the library reads these at every sketch / distance call)
    monkeypatch.setenv("DPR_MASH_KERNEL", kernel)      # index | tokens (whatever the token counts are) | table
    rng = np.random.default_rng(S * 31 + k)
    L = 2500
    clonal = _util.synth_reads(rng, 60, L, mean_bl=3e-4, lo=3e-5, hi=3e-3)
    rep = (b"ACGTTGCA" * 40 + clonal[3][:1200]) * 2                      # every k-mer twice -> duplicate sketch values
    base = clonal[5]
    seqs = []
    for first in (clonal[0], rep, _reads(rng, 1, L, L, related=False)[0]):     # three different references
        seqs = [first] + clonal[1:] + [base, base, rep, rep[: len(rep) // 2], b"ACGT" * 30, b"A" * 40, b"AC", base[:300]]
        seqs += _reads(rng, 6, 300, 3000, related=False)
        n = len(seqs)
        d = dipper_amd.Dipper(0)
        try:
            d.set_nj_mode(0)
            d.set_reads(seqs)
            sk = d.sketch(k=k, S=S)
            for q in (0, 1, n // 2, n - 1):
                assert np.array_equal(sk[q], orc.sketch(orc.pack2(seqs[q]), len(seqs[q]), k=k, S=S))
            d.dist_matrix(capi.SRC_MASH, 0, k)
            M = d.matrix()
            assert np.array_equal(M, M.T) and np.all(np.diag(M) == 0)
            for i in range(1, n):
                ref = orc.mash_dist_row(sk, k, i, i)
                assert np.allclose(M[i, :i], ref, rtol=1e-12, atol=0), (i, np.nonzero(~np.isclose(M[i, :i], ref, rtol=1e-12, atol=0))[0][:5])
            # placement batches and the transposed (divide-and-conquer assignment) output use the same kernel
            st = d.place_run(capi.SRC_MASH, n, k=k)
            Dm = np.tril(M, -1) + np.tril(M, -1).T
            ref_st = orc.place_run(Dm)
            assert np.array_equal(st["trace"][2:], ref_st["trace"][2:])
        finally:
            d.close()


def _same_state(a, b, n):
    live = 4 * n - 4
    for key in ("head", "e", "nxt", "belong", "len"):
        x, y = a[key], b[key]
        m = len(x) if key == "head" else live
        assert np.array_equal(x[:m], y[:m]), key
    assert np.array_equal(a["cid"][:5 * live], b["cid"][:5 * live])
    assert np.array_equal(a["cdis"][:5 * live], b["cdis"][:5 * live])
    assert np.array_equal(a["trace"][2:], b["trace"][2:])


@pytest.mark.parametrize("n", [3, 4, 9, 64, 300, 1100])
def test_placement_matrix_source(gpu, orc, n):
    from dipper_amd import capi
    rng = np.random.default_rng(n)
    D = _util.random_additive_matrix(rng, n, zero_frac=0.3 if n > 9 else 0.0)
    D *= 0.9 / D.max()
    gpu.set_matrix_full(D)
    got = gpu.place_run(capi.SRC_MATRIX, n)
    ref = orc.place_run(D)
    _same_state(got, ref, n)
    names = [f"T{i}" for i in range(n)]
    nw = _util.newick_from_placement(names, got["head"], got["e"], got["nxt"], got["len"], n, fmt=repr)
    assert np.abs(_util.patristic(nw, names) - D).max() < 1e-12 * n


def test_placement_noisy_matrix(gpu, orc):
    """non-additive input: clamps and ties of calculateBranchLength all get exercised"""
    from dipper_amd import capi
    rng = np.random.default_rng(99)
    n = 500
    D = np.round(rng.random((n, n)) * 0.5, 2)
    D = np.tril(D, -1) + np.tril(D, -1).T
    gpu.set_matrix_full(D)
    got = gpu.place_run(capi.SRC_MATRIX, n)
    _same_state(got, orc.place_run(D), n)


@pytest.mark.parametrize("multi", ["default", "multi", "multi-1024"])
def test_placement_msa_and_mash_sources(gpu, orc, monkeypatch, multi):
    """default: one scan + one update launch per tip at this size; multi: four tips per pair of launches from the third tip on
    (dirty-slot bookkeeping, block re-scans and -- on trees this small -- the overflow path that re-scans everything), with the
    256-thread and the 1024-thread update workgroup: same adjacency, lists and trace as the oracle's tip-by-tip order."""
    from dipper_amd import capi
    if multi != "default":
        monkeypatch.setenv("DPR_PLACE_MULTI_MIN", "3")
    if multi == "multi-1024":
        monkeypatch.setenv("DPR_PLACE_MULTI_BIG", "1")
    rng = np.random.default_rng(123)
    n, L = 400, 2000
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    gpu.set_msa(capi.pack4_many(seqs), L)
    gpu.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
    M = gpu.matrix()
    got = gpu.place_run(capi.SRC_MSA, n, dist_type=capi.DIST_JC)
    _same_state(got, orc.place_run(M), n)

    reads = _reads(rng, 300, 3000, 6000)
    gpu.set_reads(reads)
    gpu.sketch(k=15, S=1000, fetch=False)
    gpu.dist_matrix(capi.SRC_MASH, 0, 15)
    M = gpu.matrix()
    got = gpu.place_run(capi.SRC_MASH, len(reads), k=15)
    _same_state(got, orc.place_run(M), len(reads))


def test_placement_repeatable_in_both_launch_shapes(monkeypatch):
    """the edge split is shared by two wavefronts of the update launch (place_split_wave): one must not overwrite what the other still
    reads.  30 000 tips three times per launch shape (one tip per launch pair; four per pair from tip 64 on): every trace bit-equal
    -- a race there showed up as a handful of differing tips between two runs of the same input."""
    import dipper_amd
    from dipper_amd import capi
    n, L = 30000, 600
    seqs = _util.synth_alignment(np.random.default_rng(77), n, L, mean_bl=2e-3, lo=1e-4, hi=2e-2)
    packed = capi.pack4_many(seqs)
    traces = []
    for shape in ("single", "multi"):
        if shape == "multi":
            monkeypatch.setenv("DPR_PLACE_MULTI_MIN", "64")
        for rep in range(3):
            d = dipper_amd.Dipper(0)
            try:
                d.set_msa(packed, L)
                traces.append(d.place_run(capi.SRC_MSA, n, dist_type=capi.DIST_JC)["trace"].copy())
            finally:
                d.close()
    for k, t in enumerate(traces[1:], 1):
        if not np.array_equal(t, traces[0]):
            bad = np.flatnonzero((t != traces[0]).any(axis=1))
            raise AssertionError(f"run {k} differs from run 0 at {len(bad)} tips, first tip {bad[0]}: {t[bad[0]]} vs {traces[0][bad[0]]}")


@pytest.mark.parametrize("serial", [False, True], ids=["rounds", "serial"])
def test_backbone_import_lists(gpu, orc, monkeypatch, serial):
    """initializeDeviceArrays (src/placement_close_k.cu:126-264): closest lists of an imported backbone.
    The parallel relaxation rounds and the reference's leaf-by-leaf order (DPR_IMPORT_SERIAL) must both give
    the oracle's lists bit for bit -- on the reference's own t2 backbone (1000 tips, many zero lengths =
    ties) and on a caterpillar (diameter ~ m: many rounds)."""
    import os
    from dipper_amd import capi
    if serial:
        monkeypatch.setenv("DPR_IMPORT_SERIAL", "1")
    gold = os.path.join(os.path.dirname(__file__), "golden")
    t2 = open(os.path.join(gold, "t2.backbone.nwk")).readline()
    m = 300
    cat = "(" * (m - 1) + "L0:0.01"
    for i in range(1, m):
        cat += f",L{i}:{0.01 + 0.001 * (i % 7)}):{0.002 * (i % 3)}"
    cat = cat[:cat.rfind(":")] + ";"
    for nwk in (t2, cat):
        kids = _util.parse_newick(nwk)[0]
        mm = sum(1 for v in kids if not kids[v])
        n = mm + 1
        st, _ = _util.backbone_state(orc, nwk, n)
        ref = {k: v.copy() for k, v in st.items()}
        orc.place_init_lists(n, mm, ref)
        # one query (tip mm) is placed after the import: its row of a random matrix
        D = np.round(np.random.default_rng(mm).random((n, n)) * 0.3, 3)
        D = np.tril(D, -1) + np.tril(D, -1).T
        ref = orc.place_run(D, first=mm, state=ref)
        gpu.set_matrix_full(D)
        got = gpu.place_run(capi.SRC_MATRIX, n, first=mm, state={k: st[k].copy() for k in ("head", "e", "nxt", "belong", "len")})
        live = 4 * n - 4
        for key in ("head", "e", "nxt", "belong", "len"):
            assert np.array_equal(got[key][:(2 * n if key == "head" else live)], ref[key][:(2 * n if key == "head" else live)]), key
        assert np.array_equal(got["cid"][:5 * live], ref["cid"][:5 * live])
        assert np.array_equal(got["cdis"][:5 * live], ref["cdis"][:5 * live])
        assert np.array_equal(got["trace"][mm:], ref["trace"][mm:])


@pytest.mark.parametrize("S", [1, 5, 64, 65, 500, 1024, 2000])
def test_mash_other_sketch_sizes(gpu, orc, S):
    """-s other than 1000 (the reference corrupts its transpose there; intended behaviour built):
    the table kernel handles every S <= 1024 (partial last step, S not a multiple of 64), larger sketches go
    through the literal merge kernel; both against the oracle's merge."""
    from dipper_amd import capi
    rng = np.random.default_rng(S)
    seqs = _reads(rng, 40, 1500, 4000) + [b"ACGT" * 50, b"AC", b"ACGTTGCA" * 400]
    n = len(seqs)
    gpu.set_reads(seqs)
    sk = gpu.sketch(k=12, S=S)
    for q, s in enumerate(seqs):
        assert np.array_equal(sk[q], orc.sketch(orc.pack2(s), len(s), k=12, S=S)), q
    gpu.dist_matrix(capi.SRC_MASH, 0, 12)
    M = gpu.matrix()
    assert np.array_equal(M, M.T)
    for i in range(n):
        assert np.allclose(M[i, :i], orc.mash_dist_row(sk, 12, i, i), rtol=1e-12, atol=0), i
    # placement rows (lower-triangle batches, no mirror) give the same distances
    got = gpu.place_run(capi.SRC_MASH, n, k=12)
    _same_state(got, orc.place_run(M), n)


@pytest.mark.parametrize("S,k", [(300, 15), (1000, 11)])
def test_mash_index_kernel_several_chunks(orc, monkeypatch, S, k):
    """The inverted-index kernel over several 512-tip chunks (the last one partial): a clonal clade, a divergent clade,
    unrelated reads, short reads (padding values) and repeats, in shuffled order; the whole matrix against the oracle's
    literal loop on a sample of rows that covers every chunk, plus the transposed block of the divide-and-conquer
    assignment through dc_run's own test elsewhere."""
    import dipper_amd
    from dipper_amd import capi
    monkeypatch.setenv("DPR_MASH_KERNEL", "index")
    rng = np.random.default_rng(S + k)
    L = 1500
    seqs = _util.synth_reads(rng, 700, L, mean_bl=5e-5, lo=5e-6, hi=5e-4)
    seqs += _util.synth_reads(rng, 500, L, mean_bl=2e-3, lo=2e-4, hi=2e-2)
    seqs += _reads(rng, 60, 200, 2500, related=False)
    seqs += [seqs[3][:120], b"ACGT" * 12, b"A" * 30, (b"ACGTTGCA" * 30 + seqs[5][:600]) * 2, seqs[10], seqs[10]]
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    n = len(seqs)
    assert n > 2 * 512 and n % 512 != 0
    d = dipper_amd.Dipper(0)
    try:
        d.set_nj_mode(0)
        d.set_reads(seqs)
        sk = d.sketch(k=k, S=S)
        d.dist_matrix(capi.SRC_MASH, 0, k)
        M = d.matrix()
        assert np.array_equal(M, M.T) and np.all(np.diag(M) == 0)
        rows = sorted(set([1, 2, 63, 64, 511, 512, 513, 1023, 1024, 1025, n - 2, n - 1] + list(rng.integers(1, n, size=30))))
        for i in rows:
            ref = orc.mash_dist_row(sk, k, i, i)
            assert np.allclose(M[i, :i], ref, rtol=1e-12, atol=0), (i, np.nonzero(~np.isclose(M[i, :i], ref, rtol=1e-12, atol=0))[0][:5])
        # placement batches (rows i0 .. against columns below them) use the same kernel
        st = d.place_run(capi.SRC_MASH, n, k=k)
        Dm = np.tril(M, -1) + np.tril(M, -1).T
        assert np.array_equal(st["trace"][2:], orc.place_run(Dm)["trace"][2:])
    finally:
        d.close()


@pytest.mark.parametrize("n,S,k", [(2, 1, 2), (3, 2, 3), (511, 7, 6), (512, 64, 4), (513, 5, 2), (1025, 33, 5)])
def test_mash_index_kernel_edges(orc, monkeypatch, n, S, k):
    """Chunk boundaries (one tip short of a chunk, exactly one chunk, one tip over, two chunks + 1), tiny sketches and small
    k (few distinct hash values: almost every sketch value is a duplicate or a padding value): inverted-index kernel
    against the oracle's literal loop."""
    import dipper_amd
    from dipper_amd import capi
    monkeypatch.setenv("DPR_MASH_KERNEL", "index")
    rng = np.random.default_rng(n * 131 + S * 7 + k)
    seqs = _reads(rng, n, 12, 220, related=(n % 2 == 1))
    seqs[0] = b"AC"                                   # shorter than k for k > 2: an all-padding sketch
    d = dipper_amd.Dipper(0)
    try:
        d.set_nj_mode(0)
        d.set_reads(seqs)
        sk = d.sketch(k=k, S=S)
        d.dist_matrix(capi.SRC_MASH, 0, k)
        M = d.matrix()
        assert np.array_equal(M, M.T) and np.all(np.diag(M) == 0)
        rows = range(1, n) if n <= 513 else sorted(set([1, 511, 512, 513, 1023, 1024] + list(rng.integers(1, n, size=40))))
        for i in rows:
            ref = orc.mash_dist_row(sk, k, i, i)
            assert np.allclose(M[i, :i], ref, rtol=1e-12, atol=0), (i, np.nonzero(~np.isclose(M[i, :i], ref, rtol=1e-12, atol=0))[0][:5])
    finally:
        d.close()


@pytest.mark.timeout(900)
def test_mash_matrix_above_32768_tips_on_the_index_kernel(tmp_path, orc):
    """`-i r -m 2` above 32 768 tips builds the matrix in row blocks of 32 768 (ctx_nj.hip); until round 6 only the first block took
    the inverted-index kernel (its mirror write assumed the block starts at row 0) and the rest the literal pair kernel, ~30 x slower.
    36 000 tips: sampled rows of both blocks against the oracle's merge (orc_mash_dist_row = mashDistConstruction,
    src/mash.cu:426-455), symmetry across the block boundary, zero diagonal -- and the build takes index-kernel time."""
    import time
    import dipper_amd
    from dipper_amd import capi
    n = 36000
    inp = _util.gen_synth(tmp_path, "r36k", n, 1200, 3, 2e-3, 2e-4, 2e-2, reads=True)
    d = dipper_amd.Dipper(0)
    try:
        d.set_nj_mode(0)                   # plain slot-space matrix
        d.set_reads_packed(*inp["reads"])
        sk = d.sketch(15, 1000)
        t0 = time.perf_counter()
        d.dist_matrix(capi.SRC_MASH, 0, 15)
        dt = time.perf_counter() - t0
        rows = [1, 5, 700, 32767, 32768, 32769, 33000, 35999]
        got = {i: d.matrix_row(i) for i in rows}
        for i in rows:
            ref = orc.mash_dist_row(sk, 15, i, i)
            assert np.allclose(got[i][:i], ref, rtol=1e-12, atol=0), i
            assert got[i][i] == 0.0
        # the mirror of the second block's rows lives in the first block's rows and vice versa
        r700, r33000 = got[700], got[33000]
        assert r700[33000] == r33000[700] and r700[35999] == got[35999][700] and got[32767][32769] == got[32769][32767]
        assert dt < 2.0, dt                # 6.5e8 pairs: ~0.1 s on the index kernel, > 1.5 s with the literal kernel on the second block
    finally:
        d.close()


def test_mash_matrix_row_sharded_equals_single_rank():
    """The Mash matrix of ranks that share it by row blocks (streaming NJ over several GPUs): every rank walks all rows on the
    inverted index and keeps the pairs of the rows it owns, direct and mirrored (mash_dist_matrix_sharded).  Three virtual ranks:
    every row == the single-rank matrix, and the sharded NJ run == the single-rank merge log."""
    import dipper_amd
    from dipper_amd import capi
    rng = np.random.default_rng(77)
    seqs = _reads(rng, 500, 1200, 2500)
    n = len(seqs)
    one = dipper_amd.Dipper(0)
    try:
        one.set_nj_mode(0)
        one.set_reads(seqs)
        one.sketch(k=15, S=1000, fetch=False)
        one.dist_matrix(capi.SRC_MASH, 0, 15)
        M1 = one.matrix()
        ref = one.nj_run()
    finally:
        one.close()
    sh = dipper_amd.Dipper(0, virtual_world=3)
    try:
        sh.set_nj_mode(0)
        sh.set_reads(seqs)
        sh.sketch(k=15, S=1000, fetch=False)
        sh.dist_matrix(capi.SRC_MASH, 0, 15)
        M3 = sh.matrix()
        assert np.array_equal(M3, M1)
        res = sh.nj_run()
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(res[key], ref[key]), key
    finally:
        sh.close()
