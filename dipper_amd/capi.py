"""ctypes binding of include/dipper_hip.h.  Thin: every method is one C-ABI call (plus numpy
buffer plumbing).  There is no Python or CPU fallback -- if the library or the GPU is missing the
call raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SRC_MSA, SRC_MASH, SRC_MATRIX = 1, 2, 3
DIST_UNCORRECTED, DIST_JC = 1, 2

c_i32p = C.POINTER(C.c_int32)
c_u64p = C.POINTER(C.c_uint64)
c_f64p = C.POINTER(C.c_double)


class DipperError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dipper_hip error {code}: {msg}")
        self.code = code


def library_path():
    # (DPR_LIB: another build of the library -- kernel-variant experiments under profiles/)
    return os.environ.get("DPR_LIB") or os.path.join(_HERE, "libdipper_hip.so")


def load_library():
    """Loads libdipper_hip.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(path)
    L.dpr_last_error.restype = C.c_char_p
    L.dpr_abi_version.restype = C.c_int
    L.dpr_pack4.argtypes = [C.c_char_p, C.c_uint64, c_u64p]
    L.dpr_pack2.argtypes = [C.c_char_p, C.c_uint64, c_u64p]
    L.dpr_njr_owner.argtypes = [C.c_int64, C.c_int]
    L.dpr_njr_local_row.argtypes = [C.c_int64, C.c_int]
    L.dpr_njr_local_row.restype = C.c_int64
    L.dpr_njr_global_pos.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.dpr_njr_global_pos.restype = C.c_int64
    L.dpr_njr_rows_cap.argtypes = [C.c_int64, C.c_int]
    L.dpr_njr_rows_cap.restype = C.c_int64
    L.dpr_shard_owner.argtypes = [C.c_int64, C.c_int]
    L.dpr_shard_local_row.argtypes = [C.c_int64, C.c_int]
    L.dpr_shard_local_row.restype = C.c_int64
    L.dpr_shard_rows.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.dpr_shard_rows.restype = C.c_int64
    L.dpr_shard_global_row.argtypes = [C.c_int64, C.c_int, C.c_int]
    L.dpr_shard_global_row.restype = C.c_int64
    L.dpr_record_reduce.argtypes = [C.c_void_p, C.c_int]
    L.dpr_nj_key.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    L.dpr_nj_key.restype = C.c_uint64
    L.dpr_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.dpr_create_virtual.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
    L.dpr_destroy.argtypes = [C.c_void_p]
    L.dpr_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.dpr_comm_unique_id.argtypes = [C.c_void_p]
    L.dpr_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.dpr_comm_selftest.argtypes = [C.c_void_p]
    L.dpr_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dpr_comm_init_local.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.dpr_comm_init_shared.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_int]
    L.dpr_comm_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.dpr_shared_abort.argtypes = [C.c_void_p]
    L.dpr_shared_failed.argtypes = [C.c_void_p]
    L.dpr_shared_barrier.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_int]
    L.dpr_shared_gather.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.dpr_peer_export.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.dpr_peer_attach.argtypes = [C.c_void_p, C.c_void_p]
    L.dpr_ctx_set_nj_exchange.argtypes = [C.c_void_p, C.c_int]
    L.dpr_get_nj_exchange_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_char_p, C.c_int]
    L.dpr_ctx_set_poll_limit_ms.argtypes = [C.c_void_p, C.c_int]
    L.dpr_ctx_set_nj_adaptive.argtypes = [C.c_void_p, C.c_int]
    L.dpr_get_nj_adaptive_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.dpr_scan_tune.argtypes = [C.c_int, C.c_int, C.c_int]
    L.dpr_set_nj_mode.argtypes = [C.c_int]
    L.dpr_set_nj_multi_plan.argtypes = [C.c_int]
    L.dpr_nj_is_unit_sharded.argtypes = [C.c_void_p]
    L.dpr_get_nj_multi_info.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.dpr_ctx_set_nj_mode.argtypes = [C.c_void_p, C.c_int]
    L.dpr_ctx_set_nj_multi_plan.argtypes = [C.c_void_p, C.c_int]
    L.dpr_ctx_set_nj_virtual_shards.argtypes = [C.c_void_p, C.c_int]
    L.dpr_ctx_set_nj_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    L.dpr_get_nj_kernel_timing.argtypes = [C.c_void_p, C.POINTER(C.c_int), c_f64p, C.POINTER(C.c_int64)]
    L.dpr_nj_kernel_name.argtypes = [C.c_int]
    L.dpr_nj_kernel_name.restype = C.c_char_p
    L.dpr_get_place_timing.argtypes = [C.c_void_p, c_f64p, c_f64p]
    L.dpr_get_place_overlap.argtypes = [C.c_void_p, C.POINTER(C.c_int), c_f64p]
    L.dpr_get_place_policy.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.dpr_get_prune_stats.argtypes = [C.c_void_p, c_u64p, c_u64p]
    L.dpr_ctx_set_debug_fault.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.dpr_get_nj_progress.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.dpr_get_njp_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dpr_bw_probe.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.dpr_set_msa.argtypes = [C.c_void_p, c_u64p, C.c_int64, C.c_int64]
    L.dpr_set_reads.argtypes = [C.c_void_p, c_u64p, c_u64p, c_u64p, C.c_int64]
    L.dpr_set_matrix_lower.argtypes = [C.c_void_p, c_f64p, C.c_int64]
    L.dpr_sketch.argtypes = [C.c_void_p, C.c_int, C.c_int, c_u64p]
    L.dpr_dist_matrix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.dpr_reserve_nj.argtypes = [C.c_void_p, C.c_int64]
    L.dpr_warm_graphs.argtypes = [C.c_void_p]
    L.dpr_nj_run.argtypes = [C.c_void_p, C.c_int64, c_i32p, c_i32p, c_f64p, c_f64p, c_f64p]
    L.dpr_nj_run.restype = C.c_int64
    L.dpr_argmin_once.argtypes = [C.c_void_p, C.c_int, c_i32p, c_i32p, c_f64p, C.POINTER(C.c_float)]
    L.dpr_n_active.argtypes = [C.c_void_p]
    L.dpr_n_active.restype = C.c_int64
    L.dpr_n_total.argtypes = [C.c_void_p]
    L.dpr_n_total.restype = C.c_int64
    L.dpr_get_matrix_row.argtypes = [C.c_void_p, C.c_int64, c_f64p]
    L.dpr_get_row_sums.argtypes = [C.c_void_p, c_f64p]
    L.dpr_get_msa_counts.argtypes = [C.c_void_p, C.c_int64, c_i32p, c_i32p]
    L.dpr_msa_dist_block.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, c_f64p, C.c_int, C.POINTER(C.c_float)]
    L.dpr_get_kmer_hashes.argtypes = [C.c_void_p, C.c_int64, C.c_int, c_u64p, c_u64p, c_u64p]
    L.dpr_get_place_state.argtypes = [C.c_void_p, c_i32p, c_f64p, c_f64p]
    L.dpr_get_place_walks.argtypes = [C.c_void_p, c_i32p, C.POINTER(C.c_int64)]
    L.dpr_get_timing.argtypes = [C.c_void_p, c_f64p, c_f64p]
    L.dpr_place_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                c_i32p, c_i32p, c_i32p, c_i32p, c_f64p]
    L.dpr_place_exact_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64,
                                      c_i32p, c_i32p, c_i32p, c_i32p, c_f64p]
    L.dpr_get_exact_state.argtypes = [C.c_void_p, c_i32p, c_i32p]
    L.dpr_dc_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int,
                             c_i32p, c_i32p, c_i32p, c_i32p, c_f64p, c_i32p]
    L.dpr_njp_unit_owner.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int]
    L.dpr_dc_query_share.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.dpr_dc_deal_clusters.argtypes = [C.POINTER(C.c_int64), C.c_int64, C.c_int, c_i32p]
    L.dpr_get_dc_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), c_f64p]
    _LIB = L
    return L


DC_EXACT_LAST = 1
COMM_SHARED_BYTES = 65536
TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_IPC = 0, 1, 2


class SharedRegion:
    """DPR_COMM_SHARED_BYTES of shared memory for dpr_comm_init_shared, backed by a file under /dev/shm: the process that
    passes create=True makes it (zero-filled) BEFORE the ranks start, the ranks map it by name.  Test / bench plumbing: the
    `dipper` command forks its ranks around an anonymous shared mapping instead."""

    def __init__(self, name, create=False):
        import mmap
        self.path = os.path.join("/dev/shm", name)
        if create:
            with open(self.path, "wb") as f:
                f.write(b"\0" * COMM_SHARED_BYTES)
        self._f = open(self.path, "r+b")
        self._m = mmap.mmap(self._f.fileno(), COMM_SHARED_BYTES)
        self._buf = (C.c_char * COMM_SHARED_BYTES).from_buffer(self._m)
        self.address = C.addressof(self._buf)
        self.size = COMM_SHARED_BYTES

    def abort(self):
        load_library().dpr_shared_abort(self.address)

    def failed(self):
        return bool(load_library().dpr_shared_failed(self.address))

    def barrier(self, world, sense, timeout_ms=10000):
        L = load_library()
        _chk(L, L.dpr_shared_barrier(self.address, world, C.byref(sense), timeout_ms))

    def gather(self, rank, world, sense, mine: bytes, timeout_ms=10000):
        L = load_library()
        out = (C.c_char * (len(mine) * world))()
        _chk(L, L.dpr_shared_gather(self.address, rank, world, C.byref(sense), timeout_ms, mine, out, len(mine)))
        return [bytes(out[r * len(mine):(r + 1) * len(mine)]) for r in range(world)]

    def unlink(self):
        try:
            os.unlink(self.path)
        except OSError:
            pass


def dc_virtual_ranks(w):
    """flags value emulating w ranks of the multi-GPU divide-and-conquer path on one GPU"""
    return (w & 0xff) << 8


def set_nj_virtual_shards(w):
    """the next dist_matrix of a single-rank context emulates w unit-sharded ranks of the pruned NJ"""
    L = load_library()
    _chk(L, L.dpr_set_nj_virtual_shards(w))


def set_nj_mode(mode):
    """0 = full streaming scan every iteration, 1 = exact pruned scan (default)."""
    L = load_library()
    _chk(L, L.dpr_set_nj_mode(mode))


def set_nj_multi_plan(plan):
    """Several ranks, pruned NJ: 0 = auto (unit-sharded from 65 536 tips on), 1 = always unit-sharded, 2 = every rank
    runs the single-GPU plan."""
    L = load_library()
    _chk(L, L.dpr_set_nj_multi_plan(plan))


def _chk(L, rc):
    if rc < 0:
        raise DipperError(rc, (L.dpr_last_error() or b"").decode())
    return rc


def _p(a, t):
    return a.ctypes.data_as(t)


def pack4(seq: bytes):
    L = load_library()
    out = np.zeros((len(seq) + 15) // 16, dtype=np.uint64)
    _chk(L, L.dpr_pack4(seq, len(seq), _p(out, c_u64p)))
    return out


def pack2(seq: bytes):
    L = load_library()
    out = np.zeros((len(seq) + 31) // 32, dtype=np.uint64)
    _chk(L, L.dpr_pack2(seq, len(seq), _p(out, c_u64p)))
    return out


def pack4_many(seqs):
    """[n][ceil(L/16)] words, L = len(seqs[0]) (src/MSA.cu:19: every row uses sequence 0's length)."""
    Ls = len(seqs[0])
    W = (Ls + 15) // 16
    out = np.zeros((len(seqs), W), dtype=np.uint64)
    for i, s in enumerate(seqs):
        w = pack4(s)
        out[i, : min(W, len(w))] = w[:W]
    return out


class Dipper:
    """One GPU context (dpr_ctx)."""

    def __init__(self, device=0, virtual_world=0):
        self.L = load_library()
        h = C.c_void_p()
        if virtual_world:
            _chk(self.L, self.L.dpr_create_virtual(C.byref(h), device, virtual_world))
        else:
            _chk(self.L, self.L.dpr_create(C.byref(h), device))
        self.h = h

    def close(self):
        if self.h:
            self.L.dpr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_name(self):
        buf = C.create_string_buffer(256)
        _chk(self.L, self.L.dpr_device_name(self.h, buf, 256))
        return buf.value.decode()

    # ---- multi-GPU ------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        _chk(self.L, self.L.dpr_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank, world, uid: bytes):
        buf = (C.c_char * 128).from_buffer_copy(uid) if uid else None
        _chk(self.L, self.L.dpr_comm_init(self.h, rank, world, buf))

    def comm_selftest(self):
        _chk(self.L, self.L.dpr_comm_selftest(self.h))

    def comm_init_local(self, rank, world):
        _chk(self.L, self.L.dpr_comm_init_local(self.h, rank, world))

    def comm_init_shared(self, rank, world, region, transport=0):
        """join `world` ranks through a shared host region (SharedRegion below); transport 0 auto, 1 RCCL, 2 ipc windows"""
        self._region = region          # (the mapping must outlive the context)
        _chk(self.L, self.L.dpr_comm_init_shared(self.h, rank, world, region.address, region.size, transport))

    def comm_stats(self):
        """(transport: 'none' | 'rccl' | 'ipc' | 'local', device collectives so far)"""
        t = C.c_int()
        n = C.c_int64()
        _chk(self.L, self.L.dpr_comm_stats(self.h, C.byref(t), C.byref(n)))
        return ["none", "rccl", "ipc", "local"][t.value], int(n.value)

    def peer_export(self, n_tips):
        buf = (C.c_char * 192)()
        _chk(self.L, self.L.dpr_peer_export(self.h, n_tips, buf))
        return bytes(buf)

    def peer_attach(self, blobs):
        data = b"".join(blobs)
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        _chk(self.L, self.L.dpr_peer_attach(self.h, buf))

    def set_nj_exchange(self, plan):
        """row-sharded NJ loop: 0 legacy, 1 peer, 2 mailbox, -1 default"""
        _chk(self.L, self.L.dpr_ctx_set_nj_exchange(self.h, plan))

    def nj_exchange_info(self):
        p = C.c_int()
        nl = C.c_int64()
        nc = C.c_int64()
        note = C.create_string_buffer(256)
        _chk(self.L, self.L.dpr_get_nj_exchange_info(self.h, C.byref(p), C.byref(nl), C.byref(nc), note, 256))
        return {"plan": {0: "legacy", 1: "peer", 2: "mailbox"}.get(p.value, str(p.value)), "launches": int(nl.value),
                "collectives": int(nc.value), "note": note.value.decode()}

    def set_nj_adaptive(self, on):
        _chk(self.L, self.L.dpr_ctx_set_nj_adaptive(self.h, on))

    def nj_adaptive_stats(self):
        """(iterations run as streaming scans, epochs that switched) since the matrix was built"""
        a = C.c_int64()
        b = C.c_int64()
        _chk(self.L, self.L.dpr_get_nj_adaptive_stats(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def set_debug_fault(self, iteration, rank):
        """test hook of the one-exchange loops' cross-check: `rank` corrupts one pulled element at `iteration` (-1, -1: off)"""
        _chk(self.L, self.L.dpr_ctx_set_debug_fault(self.h, iteration, rank))

    def set_poll_limit_ms(self, ms):
        _chk(self.L, self.L.dpr_ctx_set_poll_limit_ms(self.h, ms))

    def comm_info(self):
        """(rank, ranks) as the RCCL communicator of this context reports them; (0, 1) without one"""
        r = C.c_int()
        n = C.c_int()
        _chk(self.L, self.L.dpr_comm_info(self.h, C.byref(r), C.byref(n)))
        return r.value, n.value

    # ---- inputs -----------------------------------------------------------------------------------
    def set_msa(self, packed4, L):
        p = np.ascontiguousarray(packed4, dtype=np.uint64)
        _chk(self.L, self.L.dpr_set_msa(self.h, _p(p, c_u64p), p.shape[0], L))

    def set_matrix_lower(self, rows, n):
        r = np.ascontiguousarray(rows, dtype=np.float64)
        assert r.size == n * (n - 1) // 2
        _chk(self.L, self.L.dpr_set_matrix_lower(self.h, _p(r, c_f64p), n))

    def set_matrix_full(self, D):
        """Convenience: takes an (n,n) array and hands over its strict lower triangle row by row."""
        D = np.asarray(D, dtype=np.float64)
        n = D.shape[0]
        rows = np.concatenate([D[i, :i] for i in range(n)]) if n > 1 else np.zeros(0)
        self.set_matrix_lower(rows, n)

    def set_reads(self, seqs):
        """seqs: list of byte strings (unaligned reads); packs with the 2-bit encoder."""
        words = [pack2(s) for s in seqs]
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        nw = np.array([len(w) for w in words], dtype=np.uint64)
        off = np.zeros(len(seqs), dtype=np.uint64)
        off[1:] = np.cumsum(nw)[:-1]
        flat = np.concatenate(words) if len(words) else np.zeros(0, np.uint64)
        if flat.size == 0:
            flat = np.zeros(1, np.uint64)
        self._reads = (np.ascontiguousarray(flat), off, lens)
        _chk(self.L, self.L.dpr_set_reads(self.h, _p(self._reads[0], c_u64p), _p(off, c_u64p), _p(lens, c_u64p), len(seqs)))

    def set_reads_packed(self, flat, off, lens):
        """already 2-bit packed reads: the three arrays of dpr_set_reads (twoBitCompressor layout)"""
        flat = np.ascontiguousarray(flat, dtype=np.uint64)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        self._reads = (flat, off, lens)
        _chk(self.L, self.L.dpr_set_reads(self.h, _p(flat, c_u64p), _p(off, c_u64p), _p(lens, c_u64p), len(lens)))

    def sketch(self, k=15, S=1000, fetch=True):
        n = len(self._reads[2])
        out = np.zeros((n, S), dtype=np.uint64) if fetch else None
        _chk(self.L, self.L.dpr_sketch(self.h, k, S, _p(out, c_u64p) if fetch else None))
        return out

    def kmer_hashes(self, seq, k):
        _, off, lens = self._reads
        nk = max(int(lens[seq]) - k + 1, 0)
        out = np.zeros(max(nk, 1), dtype=np.uint64)
        _chk(self.L, self.L.dpr_get_kmer_hashes(self.h, seq, k, _p(off, c_u64p), _p(lens, c_u64p), _p(out, c_u64p)))
        return out[:nk]

    def place_run(self, source, n, first=2, dist_type=1, k=15, state=None):
        st = state or dict(head=np.full(2 * n, -1, np.int32), e=np.full(8 * n, -1, np.int32),
                           nxt=np.full(8 * n, -1, np.int32), belong=np.full(8 * n, -1, np.int32),
                           len=np.full(8 * n, 2.0, np.float64))
        _chk(self.L, self.L.dpr_place_run(self.h, source, dist_type, k, first, n, _p(st["head"], c_i32p),
                                          _p(st["e"], c_i32p), _p(st["nxt"], c_i32p), _p(st["belong"], c_i32p),
                                          _p(st["len"], c_f64p)))
        cid = np.zeros(40 * n, np.int32)
        cdis = np.zeros(40 * n, np.float64)
        trace = np.zeros(3 * n, np.float64)
        _chk(self.L, self.L.dpr_get_place_state(self.h, _p(cid, c_i32p), _p(cdis, c_f64p), _p(trace, c_f64p)))
        st.update(cid=cid, cdis=cdis, trace=trace.reshape(n, 3))
        return st

    def place_exact_run(self, source, n, dist_type=1, k=15):
        """Exact placement mode (dpr_place_exact_run); returns adjacency, rev, dep, trace."""
        st = dict(head=np.full(2 * n, -1, np.int32), e=np.full(8 * n, -1, np.int32),
                  nxt=np.full(8 * n, -1, np.int32), belong=np.full(8 * n, -1, np.int32),
                  len=np.full(8 * n, 2.0, np.float64))
        _chk(self.L, self.L.dpr_place_exact_run(self.h, source, dist_type, k, n, _p(st["head"], c_i32p),
                                                _p(st["e"], c_i32p), _p(st["nxt"], c_i32p),
                                                _p(st["belong"], c_i32p), _p(st["len"], c_f64p)))
        rev = np.zeros(8 * n, np.int32)
        dep = np.zeros(2 * n, np.int32)
        trace = np.zeros(3 * n, np.float64)
        _chk(self.L, self.L.dpr_get_exact_state(self.h, _p(rev, c_i32p), _p(dep, c_i32p)))
        _chk(self.L, self.L.dpr_get_place_state(self.h, None, None, _p(trace, c_f64p)))
        st.update(rev=rev, dep=dep, trace=trace.reshape(n, 3))
        return st

    def dc_run(self, source, n, backbone, dist_type=1, k=15, flags=0):
        """Divide-and-conquer mode (dpr_dc_run); returns the adjacency, closest lists, trace, cluster ids."""
        st = dict(head=np.full(2 * n, -1, np.int32), e=np.full(8 * n, -1, np.int32),
                  nxt=np.full(8 * n, -1, np.int32), belong=np.full(8 * n, -1, np.int32),
                  len=np.full(8 * n, 2.0, np.float64))
        cl = np.full(n, -1, np.int32)
        _chk(self.L, self.L.dpr_dc_run(self.h, source, dist_type, k, n, backbone, flags, _p(st["head"], c_i32p),
                                       _p(st["e"], c_i32p), _p(st["nxt"], c_i32p), _p(st["belong"], c_i32p),
                                       _p(st["len"], c_f64p), _p(cl, c_i32p)))
        cid = np.zeros(40 * n, np.int32)
        cdis = np.zeros(40 * n, np.float64)
        trace = np.zeros(3 * n, np.float64)
        _chk(self.L, self.L.dpr_get_place_state(self.h, _p(cid, c_i32p), _p(cdis, c_f64p), _p(trace, c_f64p)))
        counts = np.zeros(5, np.int64)
        ms = np.zeros(3, np.float64)
        _chk(self.L, self.L.dpr_get_dc_stats(self.h, counts.ctypes.data_as(C.POINTER(C.c_int64)), _p(ms, c_f64p)))
        st.update(cid=cid, cdis=cdis, trace=trace.reshape(n, 3), cluster_id=cl,
                  stats=dict(clusters=int(counts[0]), max_cluster=int(counts[1]), pairs=int(counts[2]),
                             groups=int(counts[3]), jobs=int(counts[4]), backbone_ms=float(ms[0]),
                             assign_ms=float(ms[1]), cluster_ms=float(ms[2])))
        return st

    def warm_graphs(self):
        _chk(self.L, self.L.dpr_warm_graphs(self.h))

    def reserve_nj(self, n):
        _chk(self.L, self.L.dpr_reserve_nj(self.h, n))

    def dist_matrix(self, source, dist_type=1, k=15):
        _chk(self.L, self.L.dpr_dist_matrix(self.h, source, dist_type, k))

    # ---- NJ ---------------------------------------------------------------------------------------
    def nj_run(self, max_iters=-1):
        N = self.L.dpr_n_total(self.h)
        k = max(N - 2, 1)
        mx = np.zeros(k, dtype=np.int32)
        my = np.zeros(k, dtype=np.int32)
        bx = np.zeros(k, dtype=np.float64)
        by = np.zeros(k, dtype=np.float64)
        last = C.c_double(0.0)
        done = _chk(self.L, self.L.dpr_nj_run(self.h, max_iters, _p(mx, c_i32p), _p(my, c_i32p),
                                              _p(bx, c_f64p), _p(by, c_f64p), C.byref(last)))
        return dict(iters=done, merge_x=mx[:done], merge_y=my[:done], bl_x=bx[:done], bl_y=by[:done],
                    last_d=last.value)

    def nj_run_partial(self, max_iters=-1):
        """like nj_run, but a run that ends without a Q candidate (DPR_ERR_NOCAND: the reference's undefined (0,0) merge) returns
        the log up to there instead of raising: dict(..., code=0 or -4)"""
        N = self.L.dpr_n_total(self.h)
        k = max(N - 2, 1)
        mx = np.zeros(k, dtype=np.int32)
        my = np.zeros(k, dtype=np.int32)
        bx = np.zeros(k, dtype=np.float64)
        by = np.zeros(k, dtype=np.float64)
        last = C.c_double(0.0)
        it0, _ = self.nj_progress()
        rc = self.L.dpr_nj_run(self.h, max_iters, _p(mx, c_i32p), _p(my, c_i32p), _p(bx, c_f64p), _p(by, c_f64p), C.byref(last))
        if rc < 0 and rc != -4:
            _chk(self.L, rc)
        done = self.nj_progress()[0] - it0
        return dict(iters=done, code=0 if rc >= 0 else int(rc), merge_x=mx[:done], merge_y=my[:done], bl_x=bx[:done], bl_y=by[:done],
                    last_d=last.value)

    def nj_progress(self):
        """(iterations done since dist_matrix, active size)"""
        a = C.c_int64()
        b = C.c_int64()
        _chk(self.L, self.L.dpr_get_nj_progress(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def njp_shape(self):
        """launch shape of the pruned path's current epoch: dict(positions, row_groups, strips, post2, scan_grid)"""
        P = C.c_int64()
        v = [C.c_int() for _ in range(4)]
        _chk(self.L, self.L.dpr_get_njp_shape(self.h, C.byref(P), *[C.byref(x) for x in v]))
        return dict(positions=int(P.value), row_groups=v[0].value, strips=v[1].value, post2=bool(v[2].value), scan_grid=v[3].value)

    def nj_multi_info(self):
        buf = C.create_string_buffer(256)
        _chk(self.L, self.L.dpr_get_nj_multi_info(self.h, buf, 256))
        return buf.value.decode()

    def nj_is_unit_sharded(self):
        return bool(self.L.dpr_nj_is_unit_sharded(self.h))

    # plan knobs of THIS context (-1 = follow the process-wide default); effective at the next dist_matrix
    def set_nj_mode(self, mode):
        _chk(self.L, self.L.dpr_ctx_set_nj_mode(self.h, mode))

    def set_nj_multi_plan(self, plan):
        _chk(self.L, self.L.dpr_ctx_set_nj_multi_plan(self.h, plan))

    def set_nj_virtual_shards(self, w):
        _chk(self.L, self.L.dpr_ctx_set_nj_virtual_shards(self.h, w))

    def argmin_once(self, reps=1):
        i = C.c_int32()
        j = C.c_int32()
        q = C.c_double()
        ms = C.c_float()
        _chk(self.L, self.L.dpr_argmin_once(self.h, reps, C.byref(i), C.byref(j), C.byref(q), C.byref(ms)))
        return i.value, j.value, q.value, ms.value

    def set_nj_kernel_timing(self, stride):
        _chk(self.L, self.L.dpr_ctx_set_nj_kernel_timing(self.h, stride))

    def nj_kernel_timing(self):
        """per-kernel averages of the last timed NJ run (set_nj_kernel_timing): launch order, microseconds"""
        nk = C.c_int()
        ns = C.c_int64()
        us = np.zeros(8, np.float64)
        _chk(self.L, self.L.dpr_get_nj_kernel_timing(self.h, C.byref(nk), _p(us, c_f64p), C.byref(ns)))
        names = [(self.L.dpr_nj_kernel_name(i) or b"").decode() for i in range(nk.value)]
        rec = {"kernels_per_iteration": sum(1 for nm in names if not nm.startswith("(")), "sampled_iterations": int(ns.value),
               "kernel_us_avg": {nm: float(us[i]) for i, nm in enumerate(names)}}
        for i, nm in enumerate(names):
            if "scan" in nm:
                rec["scan_kernel"], rec["scan_us_avg"] = nm, float(us[i])
        return rec

    def place_walks(self):
        """(reached[n], dict(spilled, largest, total, degree_walks)) of the last placement run's closest-list walks"""
        st = (C.c_int64 * 6)()
        _chk(self.L, self.L.dpr_get_place_walks(self.h, None, st))
        return dict(spilled=int(st[0]), largest=int(st[1]), total=int(st[2]), degree_walks=int(st[3]), full_eval_dirty=int(st[4]), full_eval_rescans=int(st[5]))

    def place_walks_per_tip(self, n):
        out = np.zeros(n, dtype=np.int32)
        st = (C.c_int64 * 6)()
        _chk(self.L, self.L.dpr_get_place_walks(self.h, _p(out, c_i32p), st))
        return out

    def place_timing(self):
        a = C.c_double()
        b = C.c_double()
        _chk(self.L, self.L.dpr_get_place_timing(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def place_overlap(self):
        """(overlapped, dist_busy_ms): whether the last placement run computed its distance rows beside the tree kernels,
        and for how long those batches were in flight (not a summand of the wall time)"""
        o = C.c_int()
        b = C.c_double()
        _chk(self.L, self.L.dpr_get_place_overlap(self.h, C.byref(o), C.byref(b)))
        return bool(o.value), b.value

    def place_policy(self):
        """(batches, batches produced beside the previous batch's tree kernels) of the last placement run"""
        a = C.c_int64()
        b = C.c_int64()
        _chk(self.L, self.L.dpr_get_place_policy(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # ---- hooks ------------------------------------------------------------------------------------
    def n_active(self):
        return _chk(self.L, self.L.dpr_n_active(self.h))

    def n_total(self):
        return self.L.dpr_n_total(self.h)

    def matrix_row(self, i):
        out = np.zeros(self.n_total(), dtype=np.float64)
        _chk(self.L, self.L.dpr_get_matrix_row(self.h, i, _p(out, c_f64p)))
        return out

    def matrix(self):
        n = self.n_total()
        out = np.zeros((n, n), dtype=np.float64)
        for i in range(n):
            _chk(self.L, self.L.dpr_get_matrix_row(self.h, i, _p(out[i], c_f64p)))
        return out

    def row_sums(self):
        out = np.zeros(self.n_total(), dtype=np.float64)
        _chk(self.L, self.L.dpr_get_row_sums(self.h, _p(out, c_f64p)))
        return out

    def msa_dist_block(self, row0, nrows, ncols, dist_type=2, transposed=False, fetch=True, reps=1):
        """(block or None, average milliseconds per launch)"""
        out = np.zeros((ncols, nrows) if transposed else (nrows, ncols), dtype=np.float64) if fetch else None
        ms = C.c_float()
        _chk(self.L, self.L.dpr_msa_dist_block(self.h, row0, nrows, ncols, dist_type, 1 if transposed else 0,
                                               _p(out, c_f64p) if fetch else None, reps, C.byref(ms)))
        return out, ms.value

    def msa_counts(self, row):
        u = np.zeros(max(row, 1), dtype=np.int32)
        m = np.zeros(max(row, 1), dtype=np.int32)
        _chk(self.L, self.L.dpr_get_msa_counts(self.h, row, _p(u, c_i32p), _p(m, c_i32p)))
        return u[:row], m[:row]

    def prune_stats(self):
        a = C.c_uint64()
        b = C.c_uint64()
        _chk(self.L, self.L.dpr_get_prune_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timing(self):
        a = C.c_double()
        b = C.c_double()
        _chk(self.L, self.L.dpr_get_timing(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value
