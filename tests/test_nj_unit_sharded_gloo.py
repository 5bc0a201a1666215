"""CPU test of the multi-GPU pruned-NJ host logic (world_size 2, gloo).  Every rank holds the whole matrix;
a unit (16 rows x 512 columns of the strict lower triangle) belongs to the rank the PRODUCT's rule gives
(dpr_njp_unit_owner); each rank scans only its own units (numpy, exhaustively -- pruning only skips units
that cannot win), the per-rank best records are all-gathered, the winner is picked with the product's
dpr_record_reduce / dpr_nj_key, and the merge + update run replicated.  The merge log must be the
single-rank oracle's, bit for bit."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import _util


def _tree256(c):
    c = np.array(c, dtype=np.float64)
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
    M = np.tril(D, -1) + np.tril(D, -1).T          # every rank keeps the whole matrix (slot space here)
    # units of the strict lower triangle owned by this rank
    mine = []
    seen = 0
    for strip in range((N + 511) // 512):
        for group in range((N + 15) // 16):
            o = L.dpr_njp_unit_owner(strip, group, N, world)
            if o >= 0:
                seen += 1
                if o == rank:
                    mine.append((strip, group))
    # initial row sums (canonical 256-class tree), replicated
    U = np.zeros(N)
    for i in range(N):
        c = np.zeros(256)
        for t in range(256):
            idx = np.arange(t, N, 256)
            idx = idx[idx != i]
            s = 0.0
            for j in idx:
                s += M[i, j]
            c[t] = s
        U[i] = _tree256(c)
    rec_t = np.dtype([("q", "f8"), ("key", "u8"), ("d", "f8"), ("pad", "u8")])
    merges = []
    for it in range(N - 2):
        n = N - it
        Ur = U / float(n - 2)
        best = (10000.0, 2**64 - 1, 0.0)
        for (strip, group) in mine:
            a0, a1 = 16 * group, min(16 * group + 16, n)
            b0, b1 = 512 * strip, min(512 * strip + 512, n)
            if a0 >= a1 or b0 >= b1 or b0 >= a1 - 1 + 1:
                continue
            A = np.arange(a0, a1)[:, None]
            B = np.arange(b0, b1)[None, :]
            valid = B < A
            if not valid.any():
                continue
            d = M[a0:a1, b0:b1]
            q1 = (d - Ur[a0:a1, None]) - Ur[None, b0:b1]          # (i = a, j = b)
            q2 = (d - Ur[None, b0:b1]) - Ur[a0:a1, None]          # (i = b, j = a)
            q1 = np.where(valid, q1, np.inf)
            q2 = np.where(valid, q2, np.inf)
            m = min(q1.min(), q2.min())
            if m > best[0]:
                continue
            for qq, swap in ((q1, False), (q2, True)):
                for ia, ib in zip(*np.nonzero(qq == m)):
                    a, b = a0 + int(ia), b0 + int(ib)
                    i, j = (b, a) if swap else (a, b)
                    k = L.dpr_nj_key(i, j, n)
                    if m < best[0] or (m == best[0] and k < best[1]):
                        best = (float(m), k, float(d[ia, ib]))
        rec = np.zeros(1, dtype=rec_t)
        rec["q"], rec["key"], rec["d"] = best
        t = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).copy())
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        recs = np.concatenate([np.frombuffer(o.numpy().tobytes(), dtype=rec_t) for o in outs])
        w = L.dpr_record_reduce(recs.ctypes.data, world)
        assert w >= 0
        key, d = int(recs["key"][w]), float(recs["d"][w])
        i, j = key & 0xFFFFFF, (key >> 24) & 0xFFFFFF
        x, y = min(i, j), max(i, j)
        r = float(n - 2)
        blx = (d + U[x] / r - U[y] / r) * 0.5
        bly = d - blx
        if blx < 0:
            bly += blx; blx = 0.0
        if bly < 0:
            blx += bly; bly = 0.0
        merges.append((x, y, blx, bly))
        # replicated update (src/neighborJoining.cu:161-194), canonical U[x]
        last = n - 1
        vals = np.zeros(((n + 255) // 256) * 256)
        for m in range(n):
            if m == x or m == y:
                continue
            dxm, dym = M[x, m], M[y, m]
            v = (dxm + dym - d) * 0.5
            vals[m] = v
            if m == last:
                U[y] = U[last] + (-dxm - dym + v)
            else:
                U[m] = U[m] + (-dxm - dym + v)
        newx = vals[:n].copy()
        cs = [_tree256(vals[c:c + 256]) for c in range(0, len(vals), 256)]
        p = np.zeros(256)
        for c, v in enumerate(cs):
            p[c % 256] += v
        U[x] = _tree256(p)
        if y != last:
            M[y, :] = M[last, :]; M[:, y] = M[:, last]
        for m in range(n):
            if m not in (x, y):
                tgt = y if m == last else m
                M[x, tgt] = M[tgt, x] = newx[m]
        M[x, x] = 0.0; M[y, y] = 0.0
    out_q.put((rank, merges, len(mine), seen))
    dist.barrier()
    dist.destroy_process_group()


def test_unit_sharded_nj_two_ranks_gloo(orc):
    rng = np.random.default_rng(23)
    n = 530           # two column strips -> two test blocks of the product's ownership rule, one per rank
    D = _util.random_additive_matrix(rng, n, zero_frac=0.3)
    D = np.round(D, 2)                      # ties: the key order decides
    ref = orc.nj_run(np.tril(D, -1))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, D, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sum(cnt for _, _, cnt, _ in res)
    assert owned == res[0][3] and all(cnt > 0 for _, _, cnt, _ in res)     # a partition of the units
    for _, merges, _, _ in res:
        assert [m[0] for m in merges] == ref["merge_x"].tolist()
        assert [m[1] for m in merges] == ref["merge_y"].tolist()
        assert np.array_equal(np.array([m[2] for m in merges]), ref["bl_x"])
        assert np.array_equal(np.array([m[3] for m in merges]), ref["bl_y"])


_OWNER_CHECK = r"""
import sys
from dipper_amd import capi
L = capi.load_library()
P, world = int(sys.argv[1]), int(sys.argv[2])
G16, S = (P + 15) // 16, (P + 511) // 512
owners, per_rank = {}, [0] * world
for strip in range(S + 1):
    for group in range(G16 + 2):
        o = L.dpr_njp_unit_owner(strip, group, P, world)
        valid = group < G16 and group >= 32 * strip and strip * 512 < P - 1
        assert (o >= 0) == valid, (strip, group, o)
        if valid:
            assert 0 <= o < world
            owners[(strip, group)] = o
            per_rank[o] += 1
assert sum(per_rank) == len(owners) and min(per_rank) > 0, per_rank
print(len(owners), max(per_rank) / (sum(per_rank) / world))
"""


def test_unit_owner_rule_both_shapes():
    """dpr_njp_unit_owner covers every valid unit of the lower triangle exactly once in both test-block shapes (the large
    one -- four strips x 256 row groups per block -- forced through DPR_NJ_BIG_P), nothing outside it, no idle rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for big_p in ("1000000000", "1"):
        for P, world in ((5000, 2), (20000, 8), (9001, 3)):
            env = dict(os.environ, DPR_NJ_BIG_P=big_p, PYTHONPATH=root)
            r = subprocess.run([sys.executable, "-c", _OWNER_CHECK, str(P), str(world)], env=env, capture_output=True, text=True, cwd=root)
            assert r.returncode == 0, r.stderr[-2000:]
            units, imbalance = r.stdout.split()
            assert int(units) > 0 and float(imbalance) < 1.6, (big_p, P, world, r.stdout)
