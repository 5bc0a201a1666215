"""GPU parity: NJ hot path through the C ABI vs the CPU oracle (bit-exact merge log)."""
import numpy as np
import pytest

from tests import _util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[1, 0], ids=["pruned", "stream"])
def gpu(request):
    """Every test runs on both single-GPU NJ algorithms: exact pruned scan and full streaming scan."""
    import dipper_amd
    from dipper_amd import capi
    capi.set_nj_mode(request.param)
    d = dipper_amd.Dipper(0)
    yield d
    d.close()
    capi.set_nj_mode(1)


def _check_nj(gpu, orc, D):
    from dipper_amd import capi
    n = D.shape[0]
    gpu.set_matrix_full(D)
    gpu.dist_matrix(capi.SRC_MATRIX)
    # a-3: mirrored matrix and canonical row sums
    M = gpu.matrix()
    Dsym = np.tril(D, -1) + np.tril(D, -1).T
    assert np.array_equal(M, Dsym)
    U_ref = orc.row_sums(np.ascontiguousarray(Dsym))
    assert np.array_equal(gpu.row_sums(), U_ref)
    # a-4: one argmin
    if n > 2:
        rc, i, j, q = orc.nj_argmin(np.ascontiguousarray(Dsym), n, U_ref)
        gi, gj, gq, _ = gpu.argmin_once()
        assert rc == 0 and (gi, gj, gq) == (i, j, q)
    # a-5/a-6: the whole loop
    ref = orc.nj_run(np.tril(D, -1))
    res = gpu.nj_run()
    assert res["iters"] == ref["iters"] == max(n - 2, 0)
    assert np.array_equal(res["merge_x"], ref["merge_x"])
    assert np.array_equal(res["merge_y"], ref["merge_y"])
    assert np.array_equal(res["bl_x"], ref["bl_x"])
    assert np.array_equal(res["bl_y"], ref["bl_y"])
    assert res["last_d"] == ref["last_d"]
    return res


@pytest.mark.parametrize("n", [3, 4, 17, 64, 65, 200, 513, 700, 1025, 1500])
def test_nj_additive(gpu, orc, n):
    rng = np.random.default_rng(100 + n)
    D = _util.random_additive_matrix(rng, n)
    res = _check_nj(gpu, orc, D)
    names = [f"T{i}" for i in range(n)]
    nw = _util.newick_from_merges(names, res["merge_x"], res["merge_y"], res["bl_x"], res["bl_y"],
                                  res["last_d"], fmt=repr)
    assert np.abs(_util.patristic(nw, names) - D).max() < 1e-9


@pytest.mark.parametrize("n", [50, 300, 777])
def test_nj_ties(gpu, orc, n):
    """Exact Q ties everywhere: small-integer distances and tree metrics with zero branches."""
    rng = np.random.default_rng(n)
    D = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    D = np.tril(D, -1) + np.tril(D, -1).T
    _check_nj(gpu, orc, D)
    D2 = _util.random_additive_matrix(rng, n, zero_frac=0.5)
    _check_nj(gpu, orc, D2)


def test_nj_random_nonadditive(gpu, orc):
    rng = np.random.default_rng(5)
    n = 900
    D = rng.random((n, n))
    D = np.tril(D, -1) + np.tril(D, -1).T
    _check_nj(gpu, orc, D)


def test_nj_all_zero_matrix(gpu, orc):
    """All distances 0: every Q is 0, i.e. one global tie that the key (band, j mod 256, j, i) alone resolves, for all
    n - 2 iterations -- same merge log as the oracle."""
    n = 40
    _check_nj(gpu, orc, np.zeros((n, n)))


def test_nj_no_candidate(gpu, orc):
    """All Q >= 10000 from the first iteration on: with every distance -c the criterion is q = c n / (n - 2) > 0, so
    c = 1e5 leaves the reference's init tuple (0, 0, 10000) as the winner (it then merges slot 0 with itself:
    undefined, src/neighborJoining.cu:134-141,214).  The library returns DPR_ERR_NOCAND; the oracle agrees that the
    first scan finds nothing."""
    from dipper_amd import capi, DipperError
    n = 8
    D = np.full((n, n), -1.0e5)
    np.fill_diagonal(D, 0.0)
    rc, _, _, _ = orc.nj_argmin(np.ascontiguousarray(D), n, orc.row_sums(np.ascontiguousarray(D)))
    assert rc != 0
    gpu.set_matrix_full(D)
    gpu.dist_matrix(capi.SRC_MATRIX)
    with pytest.raises(DipperError) as ei:
        gpu.nj_run()
    assert ei.value.code == -4


def test_nj_q_exactly_10000_is_no_candidate(gpu, orc):
    """The smallest Q is exactly the reference's init value 10000.0: its strict `temp<minD` (src/neighborJoining.cu:134-141)
    records nothing, so this is the no-candidate case on both plans, as in the oracle (tests/test_oracle.py has the literal
    emulation); at q = 9999.5 it is an ordinary run."""
    from dipper_amd import capi, DipperError
    D = np.array([[0.0, -5000.0, -2500.0], [-5000.0, 0.0, -2500.0], [-2500.0, -2500.0, 0.0]])
    gpu.set_matrix_full(D)
    gpu.dist_matrix(capi.SRC_MATRIX)
    with pytest.raises(DipperError) as ei:
        gpu.argmin_once()
    assert ei.value.code == -4
    with pytest.raises(DipperError) as ei:
        gpu.nj_run()
    assert ei.value.code == -4
    D2 = D.copy()
    D2[0, 1] = D2[1, 0] = -4999.5
    _check_nj(gpu, orc, D2)


@pytest.mark.parametrize("n,L,inv", [(40, 100, 0.0), (130, 1000, 0.05), (300, 2500, 0.0), (257, 33, 0.2)])
def test_msa_dist_and_nj(gpu, orc, n, L, inv):
    from dipper_amd import capi
    rng = np.random.default_rng(n * 7 + L)
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2, invalid_frac=inv)
    packed = capi.pack4_many(seqs)
    assert np.array_equal(packed, orc.pack4_many(seqs))
    gpu.set_msa(packed, L)
    # integer counts: bit-exact
    u_ref, m_ref = orc.msa_counts(packed, L)
    for row in (1, n // 2, n - 1):
        u, m = gpu.msa_counts(row)
        assert np.array_equal(u, u_ref[row, :row]) and np.array_equal(m, m_ref[row, :row])
    for dt in (1, 2, 3, 4, 5, 6):   # uncorrected, JC69, Tajima-Nei, K2P, Tamura, Jin-Nei
        gpu.dist_matrix(capi.SRC_MSA, dt)
        M = gpu.matrix()
        assert np.array_equal(M, M.T, equal_nan=True) and np.all(np.diag(M) == 0)
        D_ref = orc.msa_dist_lower(packed, L, dt)
        lo = np.tril_indices(n, -1)
        a, b = M[lo], D_ref[lo]
        if dt == capi.DIST_UNCORRECTED:
            assert np.array_equal(a, b, equal_nan=True)          # pure integer/fp64 division
        else:
            ok = np.isfinite(b)
            assert np.array_equal(np.isnan(a), np.isnan(b))
            assert np.array_equal(np.isinf(a), np.isinf(b))
            # tolerance of north_star: 1e-6 relative (libm log/sqrt differ in the last bits)
            assert np.allclose(a[ok], b[ok], rtol=1e-11, atol=1e-300)
        if np.all(np.isfinite(M)):
            ref = orc.nj_run(np.tril(M, -1))
            res = gpu.nj_run()
            assert np.array_equal(res["merge_x"], ref["merge_x"]) and np.array_equal(res["merge_y"], ref["merge_y"])
            assert np.array_equal(res["bl_x"], ref["bl_x"]) and np.array_equal(res["bl_y"], ref["bl_y"])


def test_msa_fast_stages_band_table_and_block_hook_equal_the_plain_kernel(gpu, orc):
    """Round 6: per 16-word stage of a tile the pair kernel skips the not-a-base bookkeeping when no sequence of the tile has such a
    position there, inside the other stages it takes the seven-operation body only for the words in which BOTH a row of the wavefront
    and a column of the tile have one (words with such positions on one side only: a four-operation body with that side's X word),
    and distances of pairs with useful >= L - 15 come from a band table (msa.hip).  An alignment of 2 300 sites in
    which gaps sit in SOME stages of SOME sequences (runs inside one stage, a run across a stage boundary, a tip that is all
    gaps, 40 tips with an unknown base each in different stages, everything else clean): every type-1 / type-2 distance must equal
    the kernel with both switched off (DPR_MSA_NO_FAST / DPR_MSA_NO_BAND) bit for bit, the oracle's at rtol 1e-11, and the block
    hook (the launcher of placement batches, --add and the divide-and-conquer assignment) must return the same numbers in both
    orientations."""
    import os
    from dipper_amd import capi
    n, L = 400, 2300
    rng = np.random.default_rng(66)
    seqs = [bytearray(q) for q in _util.synth_alignment(rng, n, L, mean_bl=4e-3, lo=1e-4, hi=4e-2)]
    seqs[3][100:140] = b"-" * 40                 # inside stage 0 (512 sites per stage)
    seqs[70][500:530] = b"N" * 30                # across the boundary of stages 0 and 1
    seqs[71][1024:1030] = b"-" * 6
    seqs[200] = bytearray(b"-" * L)              # a DENSE tip (every stage): p = 1 against everything
    seqs[201] = bytearray(b"-" * L)              # ... and useful = 0 against tip 200: NaN
    for k in range(0, 2000, 100):                # a tip with 20 listed-word candidates: dense too, in scattered stages
        seqs[150][k] = ord("N")
    for t in range(40):
        seqs[260 + t][(t * 57) % L] = ord("n")
    seqs[399][L - 1] = ord("-")                  # the last site
    seqs = [bytes(q) for q in seqs]
    packed = capi.pack4_many(seqs)
    got = {}
    for tag, env in (("fast", {}), ("plain", {"DPR_MSA_NO_FAST": "1", "DPR_MSA_NO_BAND": "1"})):
        os.environ.update(env)
        try:
            gpu.set_msa(packed, L)
            for dt in (capi.DIST_UNCORRECTED, capi.DIST_JC):
                gpu.dist_matrix(capi.SRC_MSA, dt)
                got[tag, dt] = gpu.matrix()
                if tag == "fast":
                    blk, _ = gpu.msa_dist_block(130, 200, 330, dist_type=dt)
                    blk_t, _ = gpu.msa_dist_block(130, 200, 330, dist_type=dt, transposed=True)
                    sub = got[tag, dt][130:330, :330].copy()
                    sub[np.arange(200), np.arange(130, 330)] = blk[np.arange(200), np.arange(130, 330)]      # (the hook has no diagonal rule)
                    assert np.array_equal(blk, sub, equal_nan=True) and np.array_equal(blk_t, sub.T, equal_nan=True)
        finally:
            for k in env:
                os.environ.pop(k, None)
    gpu.set_msa(packed, L)
    for dt in (capi.DIST_UNCORRECTED, capi.DIST_JC):
        assert np.array_equal(got["fast", dt], got["plain", dt], equal_nan=True)
        ref = orc.msa_dist_lower(packed, L, dt)
        lo = np.tril_indices(n, -1)
        a, b = got["fast", dt][lo], ref[lo]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        ok = np.isfinite(b)
        assert np.allclose(a[ok], b[ok], rtol=1e-11, atol=1e-300)
    u_ref, m_ref = orc.msa_counts(packed, L)
    assert (u_ref[np.tril_indices(n, -1)] == L).mean() > 0.7          # most pairs are clean: the band table's first row


def test_msa_short_alignment_table_equals_computed_epilogue(gpu):
    """alignments of at most 1 024 sites read the type-1 / type-2 distance of a (useful, match) count pair from a table filled with the
    epilogue function itself (msa.hip, msa_jc_table_kernel).  Appending columns of gaps leaves every count unchanged but takes the
    alignment over the limit, i.e. onto the computed epilogue: both matrices bit for bit, unknown bases and NaN / inf cells included."""
    from dipper_amd import capi
    n, L = 300, 1000
    rng = np.random.default_rng(5)
    seqs = _util.synth_alignment(rng, n, L, mean_bl=2e-2, lo=1e-4, hi=3e-1, invalid_frac=0.1)
    seqs = [bytes(q) for q in seqs]
    seqs[7] = b"-" * L                                  # a tip without a valid site: useful = 0
    seqs[9] = bytes(b"ACGT"[(k * 7) % 4] for k in range(L))   # far from everything: saturated pairs
    wide = [q + b"-" * 40 for q in seqs]
    out = {}
    for name, data, sites in (("table", seqs, L), ("computed", wide, L + 40)):
        gpu.set_msa(capi.pack4_many(data), sites)
        for dt in (capi.DIST_UNCORRECTED, capi.DIST_JC):
            gpu.dist_matrix(capi.SRC_MSA, dt)
            out[name, dt] = gpu.matrix().copy()
    for dt in (capi.DIST_UNCORRECTED, capi.DIST_JC):
        a, b = out["table", dt], out["computed", dt]
        assert a.view(np.uint64).tobytes() == b.view(np.uint64).tobytes(), dt
    assert not np.all(np.isfinite(out["table", capi.DIST_JC]))       # (the saturated / empty pairs are there)


@pytest.mark.parametrize("n,kind", [(700, "additive"), (1500, "additive"), (777, "ties"), (900, "noisy")])
def test_nj_epoch_rebuilds(orc, monkeypatch, n, kind):
    """pruned path with the position space rebuilt (compacted + re-sorted by the current row sums) every
    time the active size halves (DPR_NJ_EPOCH_MIN lowers the size threshold so that small inputs go
    through several epochs): same merge log as the oracle, also for runs interrupted between iterations,
    and U stays the active row sums."""
    import dipper_amd
    from dipper_amd import capi
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "48")
    rng = np.random.default_rng(n)
    if kind == "additive":
        D = _util.random_additive_matrix(rng, n)
    elif kind == "ties":
        D = _util.random_additive_matrix(rng, n, zero_frac=0.4)
        D = np.round(D, 1)
    else:
        D = np.round(rng.random((n, n)), 3)
        D = np.tril(D, -1) + np.tril(D, -1).T
    capi.set_nj_mode(1)
    d = dipper_amd.Dipper(0)
    try:
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        ref = orc.nj_run(np.tril(D, -1))
        # interrupted run: the pieces cross several epoch boundaries
        parts = [d.nj_run(max_iters=k) for k in (n // 2 + 7, n // 4, 5, -1)]
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(np.concatenate([p[key] for p in parts]), ref[key]), key
        assert parts[-1]["last_d"] == ref["last_d"]
    finally:
        d.close()
    d = dipper_amd.Dipper(0)
    try:
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        k = n // 2 + 30
        d.nj_run(max_iters=k)
        na = n - k
        part = orc.nj_run(np.tril(D, -1), max_iters=k)
        assert np.array_equal(d.row_sums()[:na], part["U"][:na])
        M = d.matrix()[:na, :na]
        assert np.array_equal(M, part["D"][:na, :na])
    finally:
        d.close()


@pytest.mark.parametrize("grid,n,kind", [(3, 1500, "additive"), (1, 900, "ties"), (7, 2100, "ties"), (5, 1200, "noisy")])
def test_nj_pruned_small_scan_grid(orc, monkeypatch, grid, n, kind):
    """The unit scan with very few blocks (DPR_NJP_GRID): every block walks many listed units (the wave keeps its
    best across units, pass 2 runs only for units that reach it), the number of listed units exceeds the grid,
    and the seed records are the first `grid` ones -- same merge log as the oracle."""
    import dipper_amd
    from dipper_amd import capi
    monkeypatch.setenv("DPR_NJP_GRID", str(grid))
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "256")
    rng = np.random.default_rng(1000 + n)
    if kind == "additive":
        D = _util.random_additive_matrix(rng, n)
    elif kind == "ties":
        D = np.round(_util.random_additive_matrix(rng, n, zero_frac=0.4), 1)
    else:
        D = np.round(rng.random((n, n)), 3)
        D = np.tril(D, -1) + np.tril(D, -1).T
    capi.set_nj_mode(1)
    d = dipper_amd.Dipper(0)
    try:
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        ref = orc.nj_run(np.tril(D, -1))
        res = d.nj_run()
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(res[key], ref[key]), key
        assert res["last_d"] == ref["last_d"]
    finally:
        d.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_nj_unit_sharded_virtual_ranks(orc, monkeypatch, world):
    """multi-GPU pruned NJ (every rank holds the matrix, the unit tests / scans are shared, one all-gather of
    block records per iteration) emulated on one GPU: each emulated rank tests and scans only the units it
    owns with its own list and counters; same merge log as the oracle, with epoch rebuilds in between."""
    import dipper_amd
    from dipper_amd import capi
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "200")
    capi.set_nj_mode(1)
    capi.set_nj_virtual_shards(world)
    try:
        for n, kind in ((900, "additive"), (777, "ties"), (40, "additive")):
            rng = np.random.default_rng(n + world)
            D = _util.random_additive_matrix(rng, n, zero_frac=0.4 if kind == "ties" else 0.0)
            if kind == "ties":
                D = np.round(D, 1)
            d = dipper_amd.Dipper(0)
            try:
                d.set_matrix_full(D)
                d.dist_matrix(capi.SRC_MATRIX)
                ref = orc.nj_run(np.tril(D, -1))
                parts = [d.nj_run(max_iters=k) for k in (n // 3, 7, -1)]
                for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
                    assert np.array_equal(np.concatenate([p[key] for p in parts]), ref[key]), (key, n, kind)
                assert parts[-1]["last_d"] == ref["last_d"]
            finally:
                d.close()
    finally:
        capi.set_nj_virtual_shards(1)


@pytest.mark.parametrize("world", [4, 8])
def test_nj_unit_sharded_virtual_ranks_many_units(world):
    """The same emulation at 6 000 tips (≈ 2 300 units in 12 strips, several epochs, zero-length branches and
    therefore tie-heavy Q values): the unit-sharded run must reproduce the single-rank pruned run bit for bit."""
    import dipper_amd
    from dipper_amd import capi
    n, L = 6000, 400
    seqs = _util.synth_alignment(np.random.default_rng(7), n, L, mean_bl=1e-3, lo=1e-4, hi=1e-2)
    packed = capi.pack4_many(seqs)
    capi.set_nj_mode(1)
    res = {}
    try:
        for w in (1, world):
            capi.set_nj_virtual_shards(w)
            d = dipper_amd.Dipper(0)
            try:
                d.set_msa(packed, L)
                d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
                res[w] = d.nj_run()
            finally:
                d.close()
    finally:
        capi.set_nj_virtual_shards(1)
    assert res[1]["iters"] == res[world]["iters"] == n - 2
    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
        assert np.array_equal(res[1][key], res[world][key]), key
    assert res[1]["last_d"] == res[world]["last_d"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", [1, 0], ids=["pruned", "stream"])
def test_nj_no_candidate_with_epochs_terminates(monkeypatch, mode):
    """A NaN distance makes every row sum NaN: no Q candidate at iteration 0.  The reference's behaviour is
    undefined there; the library must return DPR_ERR_NOCAND -- also when the run is long enough to reach an
    epoch boundary of the pruned path (the host loop used to spin there forever)."""
    import dipper_amd
    from dipper_amd import capi, DipperError
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "48")
    n = 300
    rng = np.random.default_rng(3)
    D = rng.random((n, n))
    D = np.tril(D, -1) + np.tril(D, -1).T
    D[7, 3] = D[3, 7] = np.nan
    D[:, 11] = np.nan
    D[11, :] = np.nan
    d = dipper_amd.Dipper(0)
    try:
        d.set_nj_mode(mode)
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        with pytest.raises(DipperError) as ei:
            d.nj_run()
        assert ei.value.code == -4
    finally:
        d.close()


@pytest.mark.parametrize("case", [0, 1, 2, 3], ids=["nan_pair", "inf_few", "nan_and_inf", "inf_row"])
def test_nj_nonfinite_distances_equal_oracle(gpu, orc, case):
    """NaN / +inf distances that do NOT end the run at iteration 0 (tests/_nonfinite.py): both single-GPU plans follow the
    oracle's log to the same end -- all n - 2 iterations, or the same iteration without a candidate (DPR_ERR_NOCAND)."""
    from tests import _nonfinite
    n = 700
    name, D = _nonfinite.matrices(n, 31)[case]
    done, code = _nonfinite.check(gpu, orc, D)
    assert done >= n - 10, (name, done, code)          # the premise: the run goes on for (nearly) all iterations


@pytest.mark.parametrize("case", [0, 1, 2, 3], ids=["nan_pair", "inf_few", "nan_and_inf", "inf_row"])
def test_nj_nonfinite_distances_with_epoch_rebuilds(orc, monkeypatch, case):
    """the same on the pruned path with many small epochs (a rebuild sorts positions by row sum: NaN sums last, +inf sums
    before them) and the run interrupted between iterations (the node in quarantine is materialised at every stop)"""
    import dipper_amd
    from tests import _nonfinite
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "48")
    n = 900
    name, D = _nonfinite.matrices(n, 32)[case]
    d = dipper_amd.Dipper(0)
    try:
        d.set_nj_mode(1)
        d.set_nj_adaptive(0)
        _nonfinite.check(d, orc, D, chunks=(n // 2 + 7, 1, n // 4, 5, 10 ** 9))
    finally:
        d.close()


def test_context_reuse_same_shape_and_per_context_plans(orc):
    """A context that builds a matrix of the same shape again keeps all its device buffers (no hipFree / hipMalloc of
    the N x N matrices): the second and third build must be as clean as the first -- every NJ result bit-identical to
    the oracle, for both algorithms chosen PER CONTEXT while another context of the process runs the other one."""
    import dipper_amd
    from dipper_amd import capi
    rng = np.random.default_rng(77)
    n = 700
    mats = [_util.random_additive_matrix(rng, n, zero_frac=z) for z in (0.0, 0.4, 0.0)]
    mats[2] = np.round(mats[2], 1)
    refs = [orc.nj_run(np.tril(D, -1)) for D in mats]
    a, b = dipper_amd.Dipper(0), dipper_amd.Dipper(0)
    try:
        a.set_nj_mode(1)
        b.set_nj_mode(0)
        for rnd in range(2):
            for D, ref in zip(mats, refs):
                for d in (a, b):
                    d.set_matrix_full(D)
                    d.dist_matrix(capi.SRC_MATRIX)
                    if rnd == 1:
                        d.dist_matrix(capi.SRC_MATRIX)      # twice on the same input (the packed triangle stays)
                    assert np.array_equal(d.matrix(), np.tril(D, -1) + np.tril(D, -1).T)
                    res = d.nj_run()
                    for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
                        assert np.array_equal(res[key], ref[key]), key
                    assert res["last_d"] == ref["last_d"]
        # the pruned context really ran the pruned path, the other one did not
        a.dist_matrix(capi.SRC_MATRIX)
        a.nj_run(max_iters=5)
        assert a.prune_stats()[1] > 0
        b.dist_matrix(capi.SRC_MATRIX)
        with pytest.raises(dipper_amd.DipperError):
            b.prune_stats()
    finally:
        a.close()
        b.close()


def test_context_reuse_msa_and_mash_sources(orc):
    """Reuse across sources: the Mash pair kernel writes only j < i and its mirror, so the diagonal of a reused
    buffer must be cleared by the library; an MSA build of the same shape in between leaves other data behind."""
    import dipper_amd
    from dipper_amd import capi
    rng = np.random.default_rng(5)
    n, L = 260, 1500
    seqs = _util.synth_alignment(rng, n, L, mean_bl=5e-3, lo=1e-4, hi=5e-2)
    packed = capi.pack4_many(seqs)
    d = dipper_amd.Dipper(0)
    fresh = dipper_amd.Dipper(0)
    try:
        d.set_msa(packed, L)
        d.set_reads(seqs)
        d.sketch(k=15, S=1000, fetch=False)
        fresh.set_reads(seqs)
        fresh.sketch(k=15, S=1000, fetch=False)
        fresh.dist_matrix(capi.SRC_MASH, 1, 15)
        M_fresh = fresh.matrix()
        for _ in range(2):
            d.dist_matrix(capi.SRC_MSA, capi.DIST_JC)
            M1 = d.matrix()
            d.nj_run()                                   # the pruned path's epochs overwrite both matrix buffers
            d.dist_matrix(capi.SRC_MASH, 1, 15)
            M2 = d.matrix()
            assert np.array_equal(M2, M_fresh) and np.all(np.diag(M2) == 0)
            res = d.nj_run()
            ref = orc.nj_run(np.tril(M2, -1))
            for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
                assert np.array_equal(res[key], ref[key]), key
            assert np.all(np.diag(M1) == 0)
    finally:
        d.close()
        fresh.close()


@pytest.mark.parametrize("kind", ["ties", "additive", "random"])
def test_nj_adaptive_switch_to_streaming_mid_run(monkeypatch, orc, kind):
    """Adaptive plan: with the switch forced (threshold 0: every epoch goes over to full streaming scans on the position-space
    matrix after its first graph of pruned iterations) and many small epochs, the merge log is still the oracle's -- pruned and
    streaming iterations alternate within ONE run, with epoch rebuilds and interrupted calls in between."""
    import dipper_amd
    from dipper_amd import capi
    monkeypatch.setenv("DPR_NJ_STREAM_FRAC", "0")
    monkeypatch.setenv("DPR_NJ_EPOCH_MIN", "64")
    monkeypatch.setenv("DPR_NJ_GRAPH_ITERS", "8")
    rng = np.random.default_rng(11)
    n = 900
    if kind == "ties":
        D = rng.integers(1, 4, size=(n, n)).astype(np.float64)
        D = np.tril(D, -1) + np.tril(D, -1).T
    elif kind == "additive":
        D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
    else:
        D = rng.random((n, n))
        D = np.tril(D, -1) + np.tril(D, -1).T
    ref = orc.nj_run(np.tril(D, -1))
    d = dipper_amd.Dipper(0)
    try:
        d.set_nj_mode(1)
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        got = {k: [] for k in ("merge_x", "merge_y", "bl_x", "bl_y")}
        done = 0
        for chunk in (5, 40, 1, 300, 10 ** 6):
            res = d.nj_run(max_iters=chunk)
            k = res["iters"]
            for key in got:
                got[key].append(res[key][:k])
            done += k
        assert done == n - 2
        for key in got:
            assert np.array_equal(np.concatenate(got[key]), ref[key]), key
        assert res["last_d"] == ref["last_d"]
        streamed, switched = d.nj_adaptive_stats()
        assert switched >= 3 and 0 < streamed < n - 2          # both kinds of iterations really ran
    finally:
        d.close()


def test_nj_adaptive_off_is_pruned_only(orc):
    import dipper_amd
    from dipper_amd import capi
    rng = np.random.default_rng(12)
    n = 500
    D = np.ones((n, n)) - np.eye(n)           # one global tie: every unit is listed every iteration
    d = dipper_amd.Dipper(0)
    try:
        d.set_nj_mode(1)
        d.set_nj_adaptive(0)
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        res = d.nj_run()
        assert d.nj_adaptive_stats() == (0, 0)
        ref = orc.nj_run(np.tril(D, -1))
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(res[key], ref[key]), key
    finally:
        d.close()
    d = dipper_amd.Dipper(0)
    try:
        d.set_nj_mode(1)                     # default: adaptive on -- this input switches by itself
        d.set_matrix_full(D)
        d.dist_matrix(capi.SRC_MATRIX)
        res2 = d.nj_run()
        assert d.nj_adaptive_stats()[0] > 0
        for key in ("merge_x", "merge_y", "bl_x", "bl_y"):
            assert np.array_equal(res2[key], ref[key]), key
    finally:
        d.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("post2,poison", [("1", None), ("0", None), ("1", "255")])
def test_large_shape_post_kernels(post2, poison):
    """The large launch shape of the pruned path's post kernel (used from 40 000 positions: 256 row groups x 4 strips per test
    block) forced at small sizes, with both kernels for it: njp_post2_kernel (producer blocks hand row / column maxima to the
    test blocks of the same launch) and the fused njp_post_kernel<256, 4> -- merge logs equal the oracle's.  Also a run
    without any Q candidate: the test blocks must leave their poll (status 1), not wait for the producers' tag."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DPR_NJ_BIG_P="1", DPR_NJP_POST2=post2, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if poison:
        env["DPR_SHAPE_POISON"] = poison          # (0xFF / 0x40 patterns in the memory every context is created on)
    r = subprocess.run([sys.executable, "-m", "tests._njp_shape_worker"], cwd=root, env=env, capture_output=True, text=True, timeout=800)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and line, r.stderr[-3000:]
    out = json.loads(line[0][7:])
    assert len(out) == 17 and all(c["ok"] for c in out), (out, r.stderr[-2000:])
