"""CPU test of the N>1 host logic (world_size 2, gloo): a numpy model of the sharded NJ iteration
(local argmin over owned rows -> all-gather of records -> commit -> all-gather of the column slices
of x, y, n-1 -> replicated update), driven by the PRODUCT's host helpers (dpr_shard_*, dpr_nj_key,
dpr_record_reduce), must reproduce the single-rank oracle's merge log bit for bit."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import _util


def _tree256(v):
    c = np.array(v, dtype=np.float64)
    s = 128
    while s > 0:
        c[:s] = c[:s] + c[s:2 * s]
        s //= 2
    return c[0]


def _worker(rank, world, port, D, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dipper_amd import capi
    L = capi.load_library()
    N = D.shape[0]
    Dsym = np.tril(D, -1) + np.tril(D, -1).T
    owned = [i for i in range(N) if L.dpr_shard_owner(i, world) == rank]
    assert len(owned) == L.dpr_shard_rows(N, rank, world)
    loc = {i: Dsym[i].copy() for i in owned}          # owned rows at full width
    slice_len = ((N + 63) // 64 + world - 1) // world * 64

    def allgather(vec):
        t = torch.from_numpy(np.ascontiguousarray(vec))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [o.numpy() for o in outs]

    # initial row sums: owned rows, all-gathered (canonical 256-class tree)
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
    U = np.array([g[L.dpr_shard_owner(i, world)][L.dpr_shard_local_row(i, world)] for i in range(N)])

    rec_t = np.dtype([("q", "f8"), ("key", "u8"), ("d", "f8"), ("pad", "u8")])
    merges = []
    for it in range(N - 2):
        n = N - it
        r = float(n - 2)
        Ur = U / r
        best = (10000.0, 2**64 - 1, 0.0)
        for a in owned:
            if a >= n:
                continue
            for b in range(a):
                d = loc[a][b]
                for (i, j, q) in ((a, b, (d - Ur[a]) - Ur[b]), (b, a, (d - Ur[b]) - Ur[a])):
                    k = L.dpr_nj_key(i, j, n)
                    if q < best[0] or (q == best[0] and k < best[1]):
                        best = (q, k, d)
        rec = np.zeros(1, dtype=rec_t)
        rec["q"], rec["key"], rec["d"] = best
        recs = np.concatenate([np.frombuffer(x.tobytes(), dtype=rec_t) for x in allgather(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy())])
        w = L.dpr_record_reduce(recs.ctypes.data, world)
        assert w >= 0
        key = int(recs["key"][w]); d = float(recs["d"][w])
        i, j = key & 0xFFFFFF, (key >> 24) & 0xFFFFFF
        x, y = min(i, j), max(i, j)
        blX = (d + U[x] / r - U[y] / r) * 0.5
        blY = d - blX
        if blX < 0:
            blY += blX; blX = 0.0
        if blY < 0:
            blX += blY; blY = 0.0
        merges.append((x, y, blX, blY))
        last = n - 1
        sl = np.zeros(3 * slice_len)
        for a in owned:
            if a < n:
                li = L.dpr_shard_local_row(a, world)
                sl[li], sl[slice_len + li], sl[2 * slice_len + li] = loc[a][x], loc[a][y], loc[a][last]
        g = allgather(sl)
        r1 = float(n - 3)
        nchunk = (n + 255) // 256
        cs = np.zeros(nchunk)
        for c in range(nchunk):
            v = np.zeros(256)
            for t in range(256):
                i = c * 256 + t
                if i < n and i != x and i != y:
                    ro, li = L.dpr_shard_owner(i, world), L.dpr_shard_local_row(i, world)
                    dxi, dyi, far = g[ro][li], g[ro][slice_len + li], g[ro][2 * slice_len + li]
                    val = (dxi + dyi - d) * 0.5
                    v[t] = val
                    if i != last:
                        U[i] = U[i] + (-dxi - dyi + val)
                        if x in loc: loc[x][i] = val
                        if y in loc: loc[y][i] = far
                        if i in loc:
                            loc[i][x] = val; loc[i][y] = far
                    else:
                        U[y] = U[last] + (-dxi - dyi + val)
                        if x in loc: loc[x][y] = val
                        if y in loc: loc[y][x] = val
            cs[c] = _tree256(v)
        p = np.zeros(256)
        for t in range(256):
            s = 0.0
            for c in range(t, nchunk, 256):
                s += cs[c]
            p[t] = s
        U[x] = _tree256(p)
    last_d = np.zeros(1)
    if 1 in loc:
        last_d[0] = loc[1][0]
    last_d = allgather(last_d)[L.dpr_shard_owner(1, world)][0]
    if rank == 0:
        out_q.put((merges, float(last_d)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [9, 70, 150])
def test_sharded_host_logic_world2(orc, n):
    rng = np.random.default_rng(n)
    D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
    ref = orc.nj_run(np.tril(D, -1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, D, q)) for r in range(2)]
    for p in procs:
        p.start()
    merges, last_d = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [m[0] for m in merges] == ref["merge_x"].tolist()
    assert [m[1] for m in merges] == ref["merge_y"].tolist()
    assert [m[2] for m in merges] == ref["bl_x"].tolist()
    assert [m[3] for m in merges] == ref["bl_y"].tolist()
    assert last_d == ref["last_d"]
