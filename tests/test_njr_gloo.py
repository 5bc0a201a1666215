"""CPU test of the ROW-SHARDED PRUNED NJ protocol of dipper_amd/csrc/njr.hip (world_size 2 and 3, gloo): a numpy model of what
the ranks exchange and own, driven by the PRODUCT's ownership helpers (dpr_njr_owner / dpr_njr_local_row / dpr_njr_global_pos /
dpr_njr_rows_cap, dpr_nj_key, dpr_record_reduce):

  * the matrix lives in POSITION space (nodes sorted by row sum), dealt to the ranks in chunks of 1 024 positions; a rank scans
    only the pairs of its OWN rows (a pair belongs to the owner of its higher position);
  * one all-gather of the ranks' records; every rank reduces them to the winner and compares the header words (the row sum of
    the previous merge's node as each rank derived it);
  * every rank extracts COLUMNS px and py from its own rows; the all-gathered column slices, indexed through the ownership
    helpers, must BE rows px and py (the matrix is symmetric bit for bit) -- no rank reads another rank's rows in the loop;
  * the whole update replicated; the new node's column stored into own rows only, its row stored by the owner of px;
  * one epoch rebuild on the way (positions compacted and re-sorted, rows re-dealt).

The pruning itself (unit bounds) decides only WHICH pairs a rank evaluates, never the result (every GPU test runs it against the
streaming loop); the model scans all own pairs.  Merge log: the single-rank oracle's, bit for bit (src/neighborJoining.cu:117-249)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_sharded_gloo import _tree256


def _keys(i, j, n):
    """vectorised (band(i), j mod 256, j, i) key of the reference's findMinDist + min_element (= dpr_nj_key, spot-checked below)"""
    i = i.astype(np.uint64); j = j.astype(np.uint64)
    sz0, rem = np.uint64(n // 256), np.uint64(n % 256)
    thr = (sz0 + np.uint64(1)) * rem
    band = np.where(i < thr, i // (sz0 + np.uint64(1)), rem + (i - np.minimum(thr, i)) // np.maximum(sz0, np.uint64(1)))
    return (band << np.uint64(56)) | ((j & np.uint64(255)) << np.uint64(48)) | (j << np.uint64(24)) | i


def _worker(rank, world, port, D, iters, rebuild_at, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dipper_amd import capi
    L = capi.load_library()
    N = D.shape[0]
    Dsym = np.tril(D, -1) + np.tril(D, -1).T

    def allgather(vec):
        t = torch.from_numpy(np.ascontiguousarray(vec))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [o.numpy() for o in outs]

    def row_sums(M):
        U = np.zeros(M.shape[0])
        for i in range(M.shape[0]):
            c = np.zeros(256)
            for t in range(256):
                idx = np.arange(t, M.shape[0], 256)
                idx = idx[idx != i]
                s = 0.0
                for v in M[i, idx]:
                    s += v
                c[t] = s
            U[i] = _tree256(c)
        return U

    def deal(full, P):
        """this rank's chunks of a P x P position-space matrix: local row storage of dpr_njr_rows_cap rows"""
        cap = L.dpr_njr_rows_cap(P, world)
        loc = np.zeros((cap, P))
        mine = [p for p in range(P) if L.dpr_njr_owner(p, world) == rank]
        for p in mine:
            loc[L.dpr_njr_local_row(p, world)] = full[p]
            assert L.dpr_njr_global_pos(L.dpr_njr_local_row(p, world), rank, world) == p
        return loc, np.array(mine, dtype=np.int64)

    # epoch 0: positions = tips sorted by row sum (stable, as the product's host sort)
    U_tip = row_sums(Dsym)
    perm = np.argsort(U_tip, kind="stable")
    P = N
    slot_of_pos = perm.copy()                       # position -> reference slot
    pos_of_slot = np.empty(N, dtype=np.int64); pos_of_slot[perm] = np.arange(N)
    U = U_tip[perm].copy()                          # by position; NaN = dead
    loc, mine = deal(Dsym[np.ix_(perm, perm)], P)
    alive = np.ones(P, dtype=bool)
    merges = []
    ux_prev = 0.0
    n = N
    for it in range(iters):
        n = N - it
        if it == rebuild_at:
            # ---- epoch rebuild: live positions compacted and re-sorted by their current row sums; every rank pulls the rows its
            # new chunks need (model: all-gather of the old rows, then the product's dealing rule)
            allrows = allgather(loc)
            old = np.zeros((P, P))
            for p in range(P):
                old[p] = allrows[L.dpr_njr_owner(p, world)][L.dpr_njr_local_row(p, world)]
            live = np.flatnonzero(alive)
            order = live[np.argsort(U[live], kind="stable")]
            slot_of_pos = slot_of_pos[order]
            U = U[order]
            P = len(order)
            pos_of_slot = np.full(N, -1, dtype=np.int64); pos_of_slot[slot_of_pos] = np.arange(P)
            loc, mine = deal(old[np.ix_(order, order)], P)
            alive = np.ones(P, dtype=bool)
        r = float(n - 2)
        Ur = U / r
        # ---- SCAN: the pairs (a, b), b < a, of the own rows a -- both orders of the reference's full-square scan
        best = (10000.0, np.uint64(2**64 - 1), 0.0, -1, -1)
        for a in mine:
            if not alive[a] or a == 0:
                continue
            b = np.flatnonzero(alive[:a])
            if len(b) == 0:
                continue
            d = loc[L.dpr_njr_local_row(int(a), world)][b]
            sa = np.full(len(b), slot_of_pos[a]); sb = slot_of_pos[b]
            for (q, k, pi, pj) in (((d - Ur[a]) - Ur[b], _keys(sa, sb, n), np.full(len(b), a), b),
                                   ((d - Ur[b]) - Ur[a], _keys(sb, sa, n), b, np.full(len(b), a))):
                m = np.nanmin(q) if np.any(q == q) else np.inf
                if m > best[0]:
                    continue
                hit = np.flatnonzero(q == m)
                kk = hit[np.argmin(k[hit])]
                if m < best[0] or k[kk] < best[1]:
                    best = (float(m), k[kk], float(d[kk]), int(pi[kk]), int(pj[kk]))
        rec = np.array([best[0], 0.0, best[2], 0.0, ux_prev], dtype=np.float64)
        rec[1:2].view(np.uint64)[0] = best[1]
        rec[3:4].view(np.uint64)[0] = np.uint64(best[3] & 0xffffffff) | (np.uint64(best[4] & 0xffffffff) << np.uint64(32))
        recs = allgather(rec)
        # ---- EXTRACT: the replicated state agrees; the winner; the own slices of columns px / py
        assert len({float(x[4]) for x in recs}) == 1, ("replicated row sums differ", it)
        r32 = np.zeros(world, dtype=np.dtype([("q", "f8"), ("key", "u8"), ("d", "f8"), ("pad", "u8")]))
        for w_, x in enumerate(recs):
            r32["q"][w_] = x[0]; r32["key"][w_] = x[1:2].view(np.uint64)[0]; r32["d"][w_] = x[2]; r32["pad"][w_] = x[3:4].view(np.uint64)[0]
        w = L.dpr_record_reduce(r32.ctypes.data, world)
        assert w >= 0
        key, d, pad = int(r32["key"][w]), float(r32["d"][w]), int(r32["pad"][w])
        ki, kj = key & 0xFFFFFF, (key >> 24) & 0xFFFFFF
        pi, pj = pad & 0xffffffff, pad >> 32
        assert key == L.dpr_nj_key(ki, kj, n)
        x, y = min(ki, kj), max(ki, kj)
        px, py = (pi, pj) if ki < kj else (pj, pi)
        assert slot_of_pos[px] == x and slot_of_pos[py] == y
        cap = loc.shape[0]
        sl = allgather(np.concatenate([loc[:, px], loc[:, py]]))
        rowx = np.array([sl[L.dpr_njr_owner(p, world)][L.dpr_njr_local_row(p, world)] for p in range(P)])
        rowy = np.array([sl[L.dpr_njr_owner(p, world)][cap + L.dpr_njr_local_row(p, world)] for p in range(P)])
        # the columns over all ranks' rows ARE the rows (checked against the owner's storage)
        for pz, rowz in ((px, rowx), (py, rowy)):
            own = torch.zeros(P, dtype=torch.float64)
            if L.dpr_njr_owner(pz, world) == rank:
                own = torch.from_numpy(loc[L.dpr_njr_local_row(pz, world)].copy())
            dist.broadcast(own, src=L.dpr_njr_owner(pz, world))
            live = alive.copy(); live[pz] = False
            assert np.array_equal(own.numpy()[live].view(np.uint64), rowz[live].view(np.uint64)), ("column != row", it, pz)
        # ---- POST: the reference's host part and update (src/neighborJoining.cu:161-194,227-239), replicated
        blX = (d + U[px] / r - U[py] / r) * 0.5
        blY = d - blX
        if blX < 0:
            blY += blX; blX = 0.0
        if blY < 0:
            blX += blY; blY = 0.0
        merges.append((x, y, blX, blY))
        last = n - 1
        plast = int(pos_of_slot[last])
        val = np.zeros(P)
        Unew = U.copy()
        nchunk = (n + 255) // 256
        cs = np.zeros(nchunk)
        for c in range(nchunk):
            v = np.zeros(256)
            for t in range(256):
                i = c * 256 + t
                if i < n and i != x and i != y:
                    p = int(pos_of_slot[i])
                    v[t] = (rowx[p] + rowy[p] - d) * 0.5
                    val[p] = v[t]
                    Unew[p] = U[p] + (-rowx[p] - rowy[p] + v[t])
            cs[c] = _tree256(v)
        p256 = np.zeros(256)
        for t in range(256):
            s = 0.0
            for c in range(t, nchunk, 256):
                s += cs[c]
            p256[t] = s
        Unew[px] = _tree256(p256)
        Unew[py] = np.nan
        ux_prev = float(Unew[px])
        # the new node's column into the OWN rows, its row by the owner of px; position py dies; the last slot's node is slot y now
        for p in mine:
            if alive[p] and p != px and p != py:
                loc[L.dpr_njr_local_row(int(p), world)][px] = val[p]
        if L.dpr_njr_owner(px, world) == rank:
            newrow = val.copy(); newrow[px] = 0.0
            loc[L.dpr_njr_local_row(px, world)][:P] = newrow
        alive[py] = False
        if last != y:
            slot_of_pos[plast] = y
            pos_of_slot[y] = plast
        pos_of_slot[last] = -1
        slot_of_pos[py] = -1 if plast != py else slot_of_pos[py]
        U = Unew
    if rank == 0:
        out_q.put(merges)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,iters,rebuild_at", [(2, 1400, 120, 60), (3, 2200, 60, 30)])
def test_row_sharded_pruned_protocol_model(orc, world, n, iters, rebuild_at):
    """2 / 3 ranks, two / three ownership chunks (with 3 ranks one chunk each; with 2 ranks rank 0 holds chunk 0, rank 1 the
    partial chunk 1), tie-heavy distances, one epoch rebuild inside the timed iterations"""
    rng = np.random.default_rng(n + world)
    D = np.round(rng.random((n, n)) * 0.9 + 0.1, 2)
    D = np.tril(D, -1) + np.tril(D, -1).T
    ref = orc.nj_run(np.tril(D, -1), max_iters=iters, threads=4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35100 + (os.getpid() + n + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, D, iters, rebuild_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    merges = q.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [m[0] for m in merges] == ref["merge_x"][:iters].tolist()
    assert [m[1] for m in merges] == ref["merge_y"][:iters].tolist()
    assert [m[2] for m in merges] == ref["bl_x"][:iters].tolist()
    assert [m[3] for m in merges] == ref["bl_y"][:iters].tolist()


def test_ownership_helpers_partition_the_positions():
    from dipper_amd import capi
    L = capi.load_library()
    for world in (1, 2, 3, 8):
        for P in (1, 1023, 1024, 1025, 5000, 30000):
            seen = 0
            for rank in range(world):
                cap = L.dpr_njr_rows_cap(P, world)
                assert cap % 1024 == 0
                mine = [p for p in range(0, P, 97) if L.dpr_njr_owner(p, world) == rank]
                for p in mine:
                    l = L.dpr_njr_local_row(p, world)
                    assert 0 <= l < cap and L.dpr_njr_global_pos(l, rank, world) == p
                seen += len(mine)
            assert seen == len(range(0, P, 97))
