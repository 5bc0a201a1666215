"""ctypes loader for the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os as _os
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

c_i32p = C.POINTER(C.c_int32)
c_u64p = C.POINTER(C.c_uint64)
c_f64p = C.POINTER(C.c_double)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_pack4.argtypes = [C.c_char_p, C.c_uint64, c_u64p]
        L.orc_pack2.argtypes = [C.c_char_p, C.c_uint64, c_u64p]
        L.orc_msa_counts.argtypes = [c_u64p, C.c_int64, C.c_int64, c_i32p, c_i32p]
        L.orc_msa_dist_lower.argtypes = [c_u64p, C.c_int64, C.c_int64, C.c_int, c_f64p, C.c_int64]
        L.orc_msa_dist_row.argtypes = [c_u64p, C.c_int64, C.c_int, C.c_int64, C.c_int64, c_f64p]
        L.orc_fill_symmetric.argtypes = [c_f64p, C.c_int64, C.c_int64]
        L.orc_row_sums.argtypes = [c_f64p, C.c_int64, C.c_int64, c_f64p]
        L.orc_nj_argmin.argtypes = [c_f64p, C.c_int64, C.c_int64, c_f64p, C.c_int, c_i32p, c_i32p, c_f64p]
        L.orc_nj_argmin.restype = C.c_int
        L.orc_nj_run.argtypes = [c_f64p, C.c_int64, C.c_int64, C.c_int, C.c_int64, c_i32p, c_i32p,
                                 c_f64p, c_f64p, c_f64p, c_f64p]
        L.orc_nj_run.restype = C.c_int64
        L.orc_nj_last_iterations.restype = C.c_int64
        L.orc_murmur3_x64_128.argtypes = [C.c_char_p, C.c_int, C.c_uint32, c_u64p]
        L.orc_kmer_hash.argtypes = [c_u64p, C.c_uint64, C.c_int]
        L.orc_kmer_hash.restype = C.c_uint64
        L.orc_sketch.argtypes = [c_u64p, C.c_uint64, C.c_int, C.c_int, c_u64p]
        L.orc_mash_dist.argtypes = [c_u64p, c_u64p, C.c_int, C.c_int]
        L.orc_mash_dist.restype = C.c_double
        L.orc_mash_dist_row.argtypes = [c_u64p, C.c_int, C.c_int, C.c_int64, C.c_int64, c_f64p]
        L.orc_place_run.argtypes = [C.c_int64, C.c_int64, c_f64p, C.c_int64, c_i32p, c_i32p, c_i32p,
                                    c_i32p, c_f64p, c_i32p, c_f64p, c_f64p]
        L.orc_place_run.restype = C.c_int
        L.orc_place_init_lists.argtypes = [C.c_int64, C.c_int64, c_i32p, c_i32p, c_i32p, c_i32p,
                                           c_f64p, c_i32p, c_f64p]
        L.orc_dc_run.argtypes = [C.c_int64, C.c_int64, c_f64p, C.c_int64, C.c_int, c_i32p, c_i32p, c_i32p,
                                 c_i32p, c_f64p, c_i32p, c_f64p, c_i32p, c_f64p]
        L.orc_dc_run.restype = C.c_int
        L.orc_dc_run_backbone.argtypes = L.orc_dc_run.argtypes
        L.orc_dc_run_backbone.restype = C.c_int
        L.orc_place_exact_run.argtypes = [C.c_int64, c_f64p, C.c_int64, c_i32p, c_i32p, c_i32p, c_i32p, c_f64p,
                                          c_i32p, c_i32p, c_f64p]
        L.orc_place_exact_run.restype = C.c_int
        L.orc_rapidnj_run.argtypes = [c_f64p, C.c_int64, C.c_int64, C.c_int, c_i32p, c_i32p, c_f64p, c_f64p,
                                      c_i32p, c_f64p]
        L.orc_rapidnj_run.restype = C.c_int64
        L.orc_phylip_value.argtypes = [C.c_char_p]
        L.orc_phylip_value.restype = C.c_double

    # ---- encoders -------------------------------------------------------------------------
    def pack4(self, seq: bytes):
        out = np.zeros((len(seq) + 15) // 16, dtype=np.uint64)
        self.lib.orc_pack4(seq, len(seq), _p(out, c_u64p))
        return out

    def pack2(self, seq: bytes):
        out = np.zeros((len(seq) + 31) // 32, dtype=np.uint64)
        self.lib.orc_pack2(seq, len(seq), _p(out, c_u64p))
        return out

    def pack4_many(self, seqs):
        L = len(seqs[0])
        W = (L + 15) // 16
        out = np.zeros((len(seqs), W), dtype=np.uint64)
        for i, s in enumerate(seqs):
            out[i, : (len(s) + 15) // 16] = self.pack4(s)[:W]
        return out

    # ---- MSA distances ----------------------------------------------------------------------
    def msa_counts(self, packed4, L):
        n = packed4.shape[0]
        u = np.zeros((n, n), dtype=np.int32)
        m = np.zeros((n, n), dtype=np.int32)
        p = np.ascontiguousarray(packed4)
        self.lib.orc_msa_counts(_p(p, c_u64p), n, L, _p(u, c_i32p), _p(m, c_i32p))
        return u, m

    def msa_dist_lower(self, packed4, L, dist_type, ld=None):
        n = packed4.shape[0]
        ld = ld or n
        D = np.zeros((n, ld), dtype=np.float64)
        p = np.ascontiguousarray(packed4)
        self.lib.orc_msa_dist_lower(_p(p, c_u64p), n, L, dist_type, _p(D, c_f64p), ld)
        return D

    # ---- NJ ---------------------------------------------------------------------------------
    def row_sums(self, D):
        n, ld = D.shape
        U = np.zeros(n, dtype=np.float64)
        self.lib.orc_row_sums(_p(D, c_f64p), n, ld, _p(U, c_f64p))
        return U

    def nj_argmin(self, D, n, U, threads=1):
        i = C.c_int32()
        j = C.c_int32()
        q = C.c_double()
        rc = self.lib.orc_nj_argmin(_p(D, c_f64p), n, D.shape[1], _p(U, c_f64p), threads,
                                    C.byref(i), C.byref(j), C.byref(q))
        return rc, i.value, j.value, q.value

    def nj_run(self, D_lower, threads=None, max_iters=-1):
        """D_lower: (N, ld) float64, strict lower triangle valid.  Works on a copy.  threads=None: the host's cores (at most 16) from 600
        tips on -- the result does not depend on the thread count (per-thread minima are combined with the reference's key)."""
        D = np.array(D_lower, dtype=np.float64, order="C", copy=True)
        N, ld = D.shape
        if threads is None:
            threads = max(1, min(16, _os.cpu_count() or 1)) if N >= 600 else 1
        k = max(N - 2, 0)
        mx = np.zeros(k, dtype=np.int32)
        my = np.zeros(k, dtype=np.int32)
        bx = np.zeros(k, dtype=np.float64)
        by = np.zeros(k, dtype=np.float64)
        last = C.c_double(0.0)
        U = np.zeros(N, dtype=np.float64)
        it = self.lib.orc_nj_run(_p(D, c_f64p), N, ld, threads, max_iters, _p(mx, c_i32p),
                                 _p(my, c_i32p), _p(bx, c_f64p), _p(by, c_f64p), C.byref(last),
                                 _p(U, c_f64p))
        # iters = -1: no candidate left (the reference's undefined (0,0) merge); `done` entries of the log are valid either way
        return dict(iters=it, done=int(self.lib.orc_nj_last_iterations()), merge_x=mx, merge_y=my, bl_x=bx, bl_y=by, last_d=last.value, U=U, D=D)

    # ---- Mash -------------------------------------------------------------------------------
    def murmur(self, data: bytes, seed: int):
        out = np.zeros(2, dtype=np.uint64)
        self.lib.orc_murmur3_x64_128(data, len(data), seed, _p(out, c_u64p))
        return int(out[0]), int(out[1])

    def sketch(self, packed2, length, k=15, S=1000):
        out = np.zeros(S, dtype=np.uint64)
        p = np.ascontiguousarray(packed2)
        self.lib.orc_sketch(_p(p, c_u64p), length, k, S, _p(out, c_u64p))
        return out

    def mash_dist(self, A, B, k=15):
        A = np.ascontiguousarray(A, dtype=np.uint64)
        B = np.ascontiguousarray(B, dtype=np.uint64)
        return self.lib.orc_mash_dist(_p(A, c_u64p), _p(B, c_u64p), len(A), k)

    def mash_dist_row(self, sketches, k, row, ncols):
        out = np.zeros(ncols, dtype=np.float64)
        s = np.ascontiguousarray(sketches, dtype=np.uint64)
        self.lib.orc_mash_dist_row(_p(s, c_u64p), s.shape[1], k, row, ncols, _p(out, c_f64p))
        return out

    # ---- placement --------------------------------------------------------------------------
    def place_alloc(self, N):
        return dict(
            head=np.full(2 * N, -1, dtype=np.int32),
            e=np.full(8 * N, -1, dtype=np.int32),
            nxt=np.full(8 * N, -1, dtype=np.int32),
            belong=np.full(8 * N, -1, dtype=np.int32),
            len=np.full(8 * N, 2.0, dtype=np.float64),
            cid=np.full(40 * N, -1, dtype=np.int32),
            cdis=np.full(40 * N, 2.0, dtype=np.float64),
        )

    def place_run(self, dist_rows, first=2, state=None):
        D = np.ascontiguousarray(dist_rows, dtype=np.float64)
        N, ld = D.shape
        st = state or self.place_alloc(N)
        trace = np.zeros(3 * N, dtype=np.float64)
        nxt_slot = self.lib.orc_place_run(N, first, _p(D, c_f64p), ld, _p(st["head"], c_i32p),
                                          _p(st["e"], c_i32p), _p(st["nxt"], c_i32p),
                                          _p(st["belong"], c_i32p), _p(st["len"], c_f64p),
                                          _p(st["cid"], c_i32p), _p(st["cdis"], c_f64p),
                                          _p(trace, c_f64p))
        st["next_slot"] = nxt_slot
        st["trace"] = trace.reshape(N, 3)
        return st

    def place_init_lists(self, N, m, st):
        self.lib.orc_place_init_lists(N, m, _p(st["head"], c_i32p), _p(st["e"], c_i32p),
                                      _p(st["nxt"], c_i32p), _p(st["belong"], c_i32p),
                                      _p(st["len"], c_f64p), _p(st["cid"], c_i32p),
                                      _p(st["cdis"], c_f64p))

    def dc_run(self, dist_rows, B, skip_last_backbone=0, backbone_only=False):
        """Divide-and-conquer mode on a dense distance matrix (entry (i,j), j<i, read).
        backbone_only: stop after the backbone tree and the cluster assignment."""
        D = np.ascontiguousarray(dist_rows, dtype=np.float64)
        N, ld = D.shape
        st = self.place_alloc(N)
        trace = np.zeros(3 * N, dtype=np.float64)
        cl = np.full(N, -1, dtype=np.int32)
        fn = self.lib.orc_dc_run_backbone if backbone_only else self.lib.orc_dc_run
        rc = fn(N, B, _p(D, c_f64p), ld, skip_last_backbone, _p(st["head"], c_i32p),
                                 _p(st["e"], c_i32p), _p(st["nxt"], c_i32p), _p(st["belong"], c_i32p),
                                 _p(st["len"], c_f64p), _p(st["cid"], c_i32p), _p(st["cdis"], c_f64p),
                                 _p(cl, c_i32p), _p(trace, c_f64p))
        st["next_slot"] = rc
        st["cluster_id"] = cl
        st["trace"] = trace.reshape(N, 3)
        return st

    def place_exact_run(self, dist_rows):
        """Exact placement mode (src/placement.cu) on a dense matrix (entry (i,j), j<i, read)."""
        D = np.ascontiguousarray(dist_rows, dtype=np.float64)
        N, ld = D.shape
        st = self.place_alloc(N)
        st["rev"] = np.full(8 * N, -1, dtype=np.int32)
        st["dep"] = np.full(2 * N, -1, dtype=np.int32)
        trace = np.zeros(3 * N, dtype=np.float64)
        st["next_slot"] = self.lib.orc_place_exact_run(N, _p(D, c_f64p), ld, _p(st["head"], c_i32p),
                                                       _p(st["e"], c_i32p), _p(st["nxt"], c_i32p),
                                                       _p(st["belong"], c_i32p), _p(st["len"], c_f64p),
                                                       _p(st["rev"], c_i32p), _p(st["dep"], c_i32p),
                                                       _p(trace, c_f64p))
        st["trace"] = trace.reshape(N, 3)
        return st

    def rapidnj_run(self, D_full, threads=0):
        """CPU baseline (oracle/rapidnj_baseline.c): RapidNJ-style exact NJ on a full symmetric matrix.
        Returns the join log in node ids (tips 0..N-1, join t creates node N+t)."""
        D = np.array(D_full, dtype=np.float64, order="C", copy=True)
        N, ld = D.shape
        ca = np.zeros(N, np.int32); cb = np.zeros(N, np.int32)
        la = np.zeros(N); lb = np.zeros(N)
        last = np.zeros(2, np.int32); last_d = C.c_double()
        joins = self.lib.orc_rapidnj_run(_p(D, c_f64p), N, ld, threads, _p(ca, c_i32p), _p(cb, c_i32p),
                                         _p(la, c_f64p), _p(lb, c_f64p), _p(last, c_i32p), C.byref(last_d))
        return dict(joins=joins, child_a=ca[:max(joins, 0)], child_b=cb[:max(joins, 0)], bl_a=la[:max(joins, 0)],
                    bl_b=lb[:max(joins, 0)], last_pair=last, last_d=last_d.value)

    def phylip_value(self, tok: str):
        return self.lib.orc_phylip_value(tok.encode())


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, capture_output=True)


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("dipper_oracle.c", "rapidnj_baseline.c")]
        if os.environ.get("DPR_ORACLE_LIB"):       # the sanitizer build (`make -C oracle asan`; tests/test_tools.py runs it in a child with libasan preloaded)
            path = os.environ["DPR_ORACLE_LIB"]
        elif not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs):
            build()
        _LIB = Oracle(C.CDLL(path))
    return _LIB
