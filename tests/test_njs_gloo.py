"""CPU test of the ONE-EXCHANGE row-sharded NJ protocol of dipper_amd/csrc/njs.hip (world_size 2 and 3, gloo): a numpy
model of an iteration -- scan of the own rows through ROW VIEWS, all-gather of the rank records, PULLS of rows x / y / n-1
from their owners, the update with deferred row buffers R[it & 1] and the flush of the previous merge's buffers -- driven
by the PRODUCT's host helpers (dpr_shard_*, dpr_nj_key, dpr_record_reduce).  It must reproduce the single-rank oracle's
merge log bit for bit, AND the storage a pull reads must be untouched by the owner's own update of the same iteration:
every pull is repeated after the owner has finished its update and has to return the same bits (that is what makes one
exchange per iteration enough; see the header of njs.hip).
Round 4: the rank records are the 64-byte NjsRec of the product -- besides (q, key, d) a sequence word, the bits of the row sum
U[x] of the previous merge AS THIS RANK DERIVED IT from the rows it pulled, and the rank's status; every rank compares the
words of all ranks before it updates.  A rank that used a wrong pulled value (the fault hook) is found out one iteration
later BY EVERY RANK, which is what turns a silently different merge log into DPR_ERR_COMM."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import _util
from tests.test_sharded_gloo import _tree256


def _worker(rank, world, port, D, out_q, fault=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dipper_amd import capi
    L = capi.load_library()
    N = D.shape[0]
    Dsym = np.tril(D, -1) + np.tril(D, -1).T
    owner = lambda i: L.dpr_shard_owner(i, world)
    owned = [i for i in range(N) if owner(i) == rank]
    loc = {i: Dsym[i].copy() for i in owned}           # matrix rows of this rank
    R = [[np.zeros(N), np.zeros(N)], [np.zeros(N), np.zeros(N)]]      # R[parity][0 = x row, 1 = y row] (window of this rank)
    pend = (-1, -1)

    def allgather(vec):
        t = torch.from_numpy(np.ascontiguousarray(vec))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [o.numpy() for o in outs]

    def view(slot, it, xp, yp):
        """storage of slot's row on its owner as it stands before merge `it` (call on the owner only)"""
        if slot == xp:
            return R[(it - 1) & 1][0]
        if slot == yp:
            return R[(it - 1) & 1][1]
        return loc[slot]

    def pull(slot, it, xp, yp):
        buf = torch.zeros(N, dtype=torch.float64)
        if owner(slot) == rank:
            buf = torch.from_numpy(view(slot, it, xp, yp).copy())
        dist.broadcast(buf, src=owner(slot))
        return buf.numpy()

    # initial row sums (canonical 256-class tree), all-gathered
    slice_len = ((N + 63) // 64 + world - 1) // world * 64
    Uloc = np.zeros(slice_len)
    for i in owned:
        c = np.zeros(256)
        for t in range(256):
            s = 0.0
            for j in range(t, N, 256):
                if j != i:
                    s += loc[i][j]
            c[t] = s
        Uloc[L.dpr_shard_local_row(i, world)] = _tree256(c)
    g = allgather(Uloc)
    U = np.array([g[owner(i)][L.dpr_shard_local_row(i, world)] for i in range(N)])

    rec_t = np.dtype([("q", "f8"), ("key", "u8"), ("d", "f8"), ("seq", "u8"), ("ux", "u8"), ("status", "u8"), ("pad0", "u8"), ("pad1", "u8")])
    assert rec_t.itemsize == 64
    merges = []
    x_prev = -1
    detected = None
    for it in range(N - 2):
        n = N - it
        r = float(n - 2)
        Ur = U / r
        xp, yp = pend
        # ---- SCAN(it): own rows through the row view
        best = (10000.0, 2**64 - 1, 0.0)
        for a in owned:
            if a >= n:
                continue
            row = view(a, it, xp, yp)
            for b in range(a):
                d = row[b]
                for (i, j, q) in ((a, b, (d - Ur[a]) - Ur[b]), (b, a, (d - Ur[b]) - Ur[a])):
                    k = L.dpr_nj_key(i, j, n)
                    if q < best[0] or (q == best[0] and k < best[1]):
                        best = (q, k, d)
        rec = np.zeros(1, dtype=rec_t)
        rec["q"], rec["key"], rec["d"] = best
        rec["seq"] = (1 << 32) | (it + 1)
        rec["ux"] = np.float64(U[x_prev] if x_prev >= 0 else 0.0).view(np.uint64)      # this rank's view of the replicated state
        recs = np.concatenate([np.frombuffer(x.tobytes(), dtype=rec_t) for x in allgather(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy())])
        # ---- POST(it): the exchange happened (sequence words) and the ranks agree on the replicated row sums
        assert np.all(recs["seq"] == ((1 << 32) | (it + 1)))
        if len(set(recs["ux"].tolist())) != 1:
            detected = it
            break
        # (the winner is reduced from the first 32 bytes of every record, the product's NjRecord prefix)
        recs32 = np.zeros(world, dtype=np.dtype([("q", "f8"), ("key", "u8"), ("d", "f8"), ("pad", "u8")]))
        recs32["q"], recs32["key"], recs32["d"] = recs["q"], recs["key"], recs["d"]
        w = L.dpr_record_reduce(recs32.ctypes.data, world)
        assert w >= 0
        key = int(recs["key"][w]); d = float(recs["d"][w])
        i, j = key & 0xFFFFFF, (key >> 24) & 0xFFFFFF
        x, y = min(i, j), max(i, j)
        last = n - 1
        rowx, rowy, rowl = pull(x, it, xp, yp), pull(y, it, xp, yp), pull(last, it, xp, yp)
        pulled = (rowx.copy(), rowy.copy(), rowl.copy())
        if fault is not None and fault == (it, rank):
            k = next(i for i in range(n) if i not in (x, y))
            rowx[k] = rowx[k] * 0.5 + 1.0e-3          # "a stale pull": this rank alone uses a wrong element of row x
        blX = (d + U[x] / r - U[y] / r) * 0.5
        blY = d - blX
        if blX < 0:
            blY += blX; blX = 0.0
        if blY < 0:
            blX += blY; blY = 0.0
        merges.append((x, y, blX, blY))
        own_x, own_y = owner(x) == rank, owner(y) == rank
        RXn, RYn = R[it & 1][0], R[it & 1][1]
        nchunk = (n + 255) // 256
        cs = np.zeros(nchunk)
        Unew = U.copy()
        for c in range(nchunk):
            v = np.zeros(256)
            for t in range(256):
                i = c * 256 + t
                if i < n and i != x and i != y:
                    dxi, dyi = rowx[i], rowy[i]
                    val = (dxi + dyi - d) * 0.5
                    v[t] = val
                    if i != last:
                        Unew[i] = U[i] + (-dxi - dyi + val)
                        if own_x: RXn[i] = val
                        if own_y: RYn[i] = rowl[i]
                        if i in loc and i != xp and i != yp:
                            far = loc[i][last]
                            loc[i][x] = val; loc[i][y] = far
                    else:
                        Unew[y] = U[last] + (-dxi - dyi + val)
                        if own_x: RXn[y] = val
                        if own_y: RYn[x] = val
                elif i == x:
                    if own_x: RXn[x] = 0.0
                elif i == y:
                    if own_y: RYn[y] = 0.0
            cs[c] = _tree256(v)
        # flush of the previous merge's row buffers (owner; not a row consumed by this merge, not the dying slot)
        for wch, p in ((0, xp), (1, yp)):
            if p < 0 or p in (x, y) or p >= last or owner(p) != rank:
                continue
            Rp = R[(it - 1) & 1][wch]
            new = Rp[:n].copy()
            new[x] = (rowx[p] + rowy[p] - d) * 0.5
            new[y] = Rp[last]
            loc[p][:n] = new
        p256 = np.zeros(256)
        for t in range(256):
            s = 0.0
            for c in range(t, nchunk, 256):
                s += cs[c]
            p256[t] = s
        Unew[x] = _tree256(p256)
        U = Unew
        # ---- the owner's update of THIS iteration must not have touched what the other ranks pull in it
        for slot, first in ((x, pulled[0]), (y, pulled[1]), (last, pulled[2])):
            again = pull(slot, it, xp, yp)
            assert np.array_equal(again[:n].view(np.uint64), first[:n].view(np.uint64)), ("pulled storage modified", it, slot)
        pend = (x, y)
        x_prev = x
    if fault is not None:
        out_q.put((rank, detected))
        dist.barrier()
        dist.destroy_process_group()
        return
    # after the loop: flush the last buffers (njs_finish_kernel), then the final distance from rank owner(1)
    n = 2
    for wch, p in ((0, pend[0]), (1, pend[1])):
        if 0 <= p < n and owner(p) == rank:
            loc[p][:n + 1] = R[(N - 3) & 1][wch][:n + 1]
    last_d = np.zeros(1)
    if 1 in loc:
        last_d[0] = loc[1][0]
    last_d = allgather(last_d)[owner(1)][0]
    if rank == 0:
        out_q.put((merges, float(last_d)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 9), (2, 150), (3, 200)])
def test_one_exchange_protocol_model(orc, world, n):
    rng = np.random.default_rng(n + world)
    if n == 150:      # tie-heavy: the new node is merged again at once, pulls then come from the row buffers
        D = rng.integers(1, 4, size=(n, n)).astype(np.float64)
        D = np.tril(D, -1) + np.tril(D, -1).T
    else:
        D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
    ref = orc.nj_run(np.tril(D, -1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + n + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    merges, last_d = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [m[0] for m in merges] == ref["merge_x"].tolist()
    assert [m[1] for m in merges] == ref["merge_y"].tolist()
    assert [m[2] for m in merges] == ref["bl_x"].tolist()
    assert [m[3] for m in merges] == ref["bl_y"].tolist()
    assert last_d == ref["last_d"]


def test_one_exchange_protocol_model_detects_a_stale_pull():
    """rank 1 of 3 uses a wrong element of a pulled row at iteration 12: its U[x] of that merge differs, the records of
    iteration 13 carry three words of which one differs, and EVERY rank stops there (the product: DPR_ERR_COMM on every rank)"""
    n, world, fault = 60, 3, (12, 1)
    D = _util.random_additive_matrix(np.random.default_rng(5), n, zero_frac=0.2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33600 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, D, q, fault)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: 13, 1: 13, 2: 13}
