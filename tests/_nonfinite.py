"""Non-finite distances on the NJ loop (`useful == 0` gives a NaN distance, p >= 0.75 gives +inf under JC69,
src/MSA.cu:233-235).  The reference's loop (src/neighborJoining.cu:117-148,161-194) has no special case for them: a NaN entry
makes the row sums of ITS two rows NaN -- those rows never win (every Q they take part in is NaN, `<` is false) but stay
active, are folded into every new node's row, and are what is left when no candidate remains (the reference then merges slots
(0,0): undefined; the library returns DPR_ERR_NOCAND); a +inf entry makes two row sums +inf, whose pairs have Q = -inf and win
at once, and the merge turns the partner's row sum into NaN (inf - inf).  The pruned path uses NaN row sums as its own marker
of dead positions and of the node in quarantine, so these are the inputs where it could part from the streaming loop.

Shared by tests/test_gpu_nj.py (natural small launch shape) and tests/_njp_shape_worker.py (large shape forced)."""
import numpy as np


def matrices(n, seed):
    """(name, D) pairs: symmetric, zero diagonal, distances ~ U(0.05, 1) rounded so that exact Q ties occur"""
    rng = np.random.default_rng(seed)
    base = np.round(rng.random((n, n)) * 0.95 + 0.05, 2)
    base = np.tril(base, -1) + np.tril(base, -1).T
    out = []

    def put(D, pairs, v):
        for a, b in pairs:
            D[a, b] = D[b, a] = v

    D = base.copy()
    put(D, [(n // 3, n // 7)], np.nan)                                   # one isolated NaN pair: two live rows with NaN sums to the end
    out.append(("nan_pair", D))
    D = base.copy()
    put(D, [(5, 2), (n - 1, 40), (n // 2, n // 2 - 1), (n - 2, 3), (77, 76)], np.inf)      # a handful of saturated JC69 pairs
    out.append(("inf_few", D))
    D = base.copy()
    put(D, [(n // 5, 9), (n - 3, n // 2)], np.nan)
    put(D, [(17, 4), (n - 1, n - 2), (n // 2 + 5, 11)], np.inf)
    put(D, [(n // 5, 17)], np.inf)                                       # a row that holds a NaN AND an inf
    out.append(("nan_and_inf", D))
    D = base.copy()
    a = n // 4
    D[a, :] = np.inf                                                     # one tip saturated against everybody
    D[:, a] = np.inf
    D[a, a] = 0.0
    out.append(("inf_row", D))
    return out


def check(d, orc, D, chunks=(10 ** 9,), threads=1):
    """runs the context's NJ (in pieces: `chunks` of max_iters) on D and compares with the oracle: the same log up to the same
    end -- either all n - 2 iterations or the same iteration without a candidate.  Returns (iterations, code)."""
    from dipper_amd import capi
    n = D.shape[0]
    ref = orc.nj_run(np.tril(D, -1), threads=threads)
    d.set_matrix_full(D)
    d.dist_matrix(capi.SRC_MATRIX)
    got = {k: [] for k in ("merge_x", "merge_y", "bl_x", "bl_y")}
    done, code, last = 0, 0, None
    for c in chunks:
        res = d.nj_run_partial(max_iters=c)
        for k in got:
            got[k].append(res[k])
        done += res["iters"]
        code, last = res["code"], res["last_d"]
        if code != 0 or done >= n - 2:
            break
    assert done == ref["done"], (done, ref["done"], code, ref["iters"])
    assert (code == -4) == (ref["iters"] == -1), (code, ref["iters"])
    for k in got:
        g = np.concatenate(got[k])
        r = ref[k][:done]
        if not np.array_equal(g, r, equal_nan=True):
            bad = int(np.flatnonzero(~((g == r) | ((g != g) & (r != r))))[0])
            raise AssertionError(f"{k} differs first at iteration {bad} of {done}: {g[bad]} vs {r[bad]}")
    if code == 0:
        assert (last == ref["last_d"]) or (last != last and ref["last_d"] != ref["last_d"]), (last, ref["last_d"])
    return done, code
