"""CPU test of the multi-GPU divide-and-conquer host logic (world_size 2, gloo).  Each rank takes the
share of the query tips and of the clusters that the PRODUCT's helpers give it (dpr_dc_query_share,
dpr_dc_deal_clusters), applies only its own clusters' changes (taken from the oracle's sequential run) to
the backbone state, and the ranks merge by all-reducing (new - old) in wrap-around 64-bit integers --
the arithmetic of dc_delta_sub / ncclAllReduce(uint64 sum) / dc_delta_add.  The merged state must be the
oracle's full state bit for bit (doubles as bit patterns, int32 arrays packed two per word)."""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import _util

KEYS = ("head", "e", "nxt", "belong", "len", "cid", "cdis")


def _words(a):
    """array -> uint64 words exactly as the device code sees them"""
    b = np.ascontiguousarray(a)
    if b.dtype.itemsize == 4 and b.size % 2:
        b = np.concatenate([b, np.zeros(1, b.dtype)])
    return b.view(np.uint64)


def _worker(rank, world, port, D, B, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dipper_amd import capi
    from tests import _orc
    L = capi.load_library()
    orc = _orc.load()
    n = D.shape[0]
    ref = orc.dc_run(D, B)                          # sequential reference: final state
    old = orc.dc_run(D, B, backbone_only=True)      # state every rank holds after the (replicated) backbone
    assert np.array_equal(ref["cluster_id"], old["cluster_id"])

    # ---- query shares: each rank contributes the ids of its share, zeros elsewhere, summed
    q0, q1 = C.c_int64(), C.c_int64()
    assert L.dpr_dc_query_share(n, B, rank, world, C.byref(q0), C.byref(q1)) == 0
    mine = np.zeros(n, dtype=np.int32)
    mine[q0.value:q1.value] = ref["cluster_id"][q0.value:q1.value]
    t = torch.from_numpy(mine)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    cl = t.numpy().copy()
    cl[:B] = -1
    assert np.array_equal(cl, ref["cluster_id"])

    # ---- clusters in the order dpr_dc_run builds them: size descending, ties by ascending slot
    slots, sizes = np.unique(cl[B:], return_counts=True)
    order = np.lexsort((slots, -sizes))
    slots, sizes = slots[order], sizes[order].astype(np.int64)
    owner = np.zeros(len(slots), dtype=np.int32)
    assert L.dpr_dc_deal_clusters(sizes.ctypes.data_as(C.POINTER(C.c_int64)), len(sizes), world,
                                  owner.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    assert set(owner.tolist()) == set(range(world))
    # what a cluster touches: its edge slot j, the reverse slot, its new slots and new nodes
    lim = 4 * B - 4
    start = {}
    acc = 0
    for s in range(lim):
        start[s] = acc
        acc += int(np.sum(cl[B:] == s))
    own_slot = np.zeros(8 * n, dtype=bool)
    own_node = np.zeros(2 * n, dtype=bool)
    for s, m, o in zip(slots.tolist(), sizes.tolist(), owner.tolist()):
        if o != rank:
            continue
        x, y = int(old["belong"][s]), int(old["e"][s])
        r = int(old["head"][y])
        while int(old["e"][r]) != x:
            r = int(old["nxt"][r])
        own_slot[[s, r]] = True
        b0 = lim + 4 * start[s]
        own_slot[b0:b0 + 4 * m] = True
        own_node[n + B + start[s] - 1:n + B + start[s] - 1 + m] = True          # middle nodes
        own_node[np.nonzero(cl == s)[0]] = True                                   # the member tips
    merged = {}
    for key in KEYS:
        per = 5 if key in ("cid", "cdis") else 1
        mask = own_node if key == "head" else np.repeat(own_slot, per)
        new = np.where(mask[:len(ref[key])], ref[key], old[key])
        delta = _words(new) - _words(old[key])                  # wrap-around uint64
        t = torch.from_numpy(delta.view(np.int64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                # two's complement sum == uint64 sum
        merged[key] = (_words(old[key]) + t.numpy().view(np.uint64)).view(ref[key].dtype)[:len(ref[key])]
    ok = all(np.array_equal(merged[k].view(np.uint8), np.ascontiguousarray(ref[k]).view(np.uint8)) for k in KEYS)
    out_q.put((rank, ok, int(own_slot.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_dc_two_ranks_gloo():
    rng = np.random.default_rng(17)
    n, B = 420, 90
    D = _util.random_additive_matrix(rng, n, zero_frac=0.2)
    D *= 0.9 / D.max()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, D, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(cnt > 0 for _, _, cnt in res)        # both ranks built clusters
