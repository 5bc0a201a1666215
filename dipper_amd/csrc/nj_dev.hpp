// Device helpers shared by the NJ translation units (nj.hip: streaming path, njp.hip: pruned path).
#pragma once
#include "dpr_internal.hpp"

namespace dpr {

typedef double v2d __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// nj_key_a on the device.  nj_band (dpr_internal.hpp) divides 64-bit integers by a run-time divisor -- two software divisions
// of ~80 instructions each, in every thread of every update role of the NJ loops, which run one wave per SIMD with nothing to
// hide instruction latency (round 4).  Slot numbers and sizes are below 2^24 (the key gives them 24 bits) and a band index is
// below 256, so: 32-bit operands, quotient from a float reciprocal, one exact correction step.  Same value as nj_band, always.
__device__ __forceinline__ uint64_t nj_key_a_dev(int64_t i64, int64_t n64)
{
    const uint32_t i = (uint32_t)i64, n = (uint32_t)n64;
    const uint32_t sz0 = n >> 8, rem = n & 255u, thr = (sz0 + 1u) * rem;
    const bool lo = i < thr;
    const uint32_t num = lo ? i : i - thr;
    uint32_t den = lo ? sz0 + 1u : sz0;
    den = den ? den : 1u;                                   // (only for slots >= n, whose keys nobody uses)
    uint32_t q = (uint32_t)((float)num * __builtin_amdgcn_rcpf((float)den));
    int32_t r = (int32_t)(num - q * den);
    if (r < 0) { --q; r += (int32_t)den; }
    if (r >= (int32_t)den) { ++q; }
    const uint32_t band = lo ? q : rem + q;
    return ((uint64_t)band << 56) | (uint64_t)i64;
}

__device__ __forceinline__ void best_update(double& bq, uint64_t& bk, double q, uint64_t k)
{
    // strict '<' on q (NaN never wins), ties resolved by the reference's visiting order (key)
    const bool take = (q < bq) | ((q == bq) & (k < bk));
    bq = take ? q : bq;
    bk = take ? k : bk;
}

__device__ __forceinline__ void wave_best(double& bq, uint64_t& bk)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oq = __shfl_down(bq, off, 64);
        const uint64_t ok = __shfl_down((unsigned long long)bk, off, 64);
        best_update(bq, bk, oq, ok);
    }
}

// block-wide (256 threads) lexicographic minimum; result valid in thread 0
__device__ __forceinline__ void block_best(double& bq, uint64_t& bk, double* sq, uint64_t* sk)
{
    wave_best(bq, bk);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sq[w] = bq; sk[w] = bk; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < kThreads / 64; ++i) best_update(bq, bk, sq[i], sk[i]);
    }
}

// ---- wave-wide reductions on the DPP path --------------------------------------------------------
// The latency-bound kernels of the pruned path reduce a handful of values per launch; a butterfly of
// __shfl (ds_bpermute, ~100+ cycles per dependent step) was a visible part of their critical path.
// Here: two quad permutes + two mirrors (VALU latency) give every lane its row-of-16 result, four
// v_readlane combine the rows.  Results are wave-uniform.
template <int CTRL> __device__ __forceinline__ uint64_t dpp_u64(uint64_t x)
{
    const int lo = (int)(uint32_t)x, hi = (int)(uint32_t)(x >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return ((uint64_t)(uint32_t)hi2 << 32) | (uint64_t)(uint32_t)lo2;
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double x)
{
    return __longlong_as_double((long long)dpp_u64<CTRL>((uint64_t)__double_as_longlong(x)));
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t x, int lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(x), lane));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

// minimum / maximum over the wave, NaN operands ignored (fmin / fmax semantics)
__device__ __forceinline__ double wave_fmin(double x)
{
    x = fmin(x, dpp_f64<kDppXor1>(x));
    x = fmin(x, dpp_f64<kDppXor2>(x));
    x = fmin(x, dpp_f64<kDppHalfMirror>(x));
    x = fmin(x, dpp_f64<kDppMirror>(x));
    return fmin(fmin(readlane_f64(x, 0), readlane_f64(x, 16)), fmin(readlane_f64(x, 32), readlane_f64(x, 48)));
}
__device__ __forceinline__ double wave_fmax(double x)
{
    x = fmax(x, dpp_f64<kDppXor1>(x));
    x = fmax(x, dpp_f64<kDppXor2>(x));
    x = fmax(x, dpp_f64<kDppHalfMirror>(x));
    x = fmax(x, dpp_f64<kDppMirror>(x));
    return fmax(fmax(readlane_f64(x, 0), readlane_f64(x, 16)), fmax(readlane_f64(x, 32), readlane_f64(x, 48)));
}
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t wave_umin64(uint64_t x)
{
    x = umin64(x, dpp_u64<kDppXor1>(x));
    x = umin64(x, dpp_u64<kDppXor2>(x));
    x = umin64(x, dpp_u64<kDppHalfMirror>(x));
    x = umin64(x, dpp_u64<kDppMirror>(x));
    return umin64(umin64(readlane_u64(x, 0), readlane_u64(x, 16)), umin64(readlane_u64(x, 32), readlane_u64(x, 48)));
}

// wave-wide winner of (q, key, positions, d) under "smaller q, then smaller key" (the lanes' q are never NaN:
// they start at the reference's 10000 and only take smaller values); result wave-uniform
__device__ __forceinline__ void wave_best4(double& bq, uint64_t& bk, uint64_t& bp, double& bd)
{
    const double wq = wave_fmin(bq);
    const uint64_t wk = wave_umin64(bq == wq ? bk : ~0ull);
    const unsigned long long own = __builtin_amdgcn_ballot_w64((bq == wq) & (bk == wk));
    const int src = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(own));   // own != 0: the lane holding wq has bk >= wk
    bp = readlane_u64(bp, src);
    bd = readlane_f64(bd, src);
    bq = wq; bk = wk;
}

// pairwise tree over 256 values, c[t] += c[t+s] for s = 128..1 (canonical order)
__device__ __forceinline__ double block_tree256(double v, double* s)
{
    const int t = threadIdx.x;
    s[t] = v;
    __syncthreads();
#pragma unroll
    for (int st = 128; st > 0; st >>= 1) {
        if (t < st) s[t] = s[t] + s[t + st];
        __syncthreads();
    }
    return s[0];
}

// The same tree (same operand pairs at every level, so the same bits) with three barriers instead of nine: the
// levels 128 and 64 go through LDS, the levels 32 .. 1 stay inside wave 0 (lane t adds lane t + st).  All 256
// threads must call it; the sum is valid in thread 0 only.
constexpr int kDppRowShl1 = 0x101, kDppRowShl2 = 0x102, kDppRowShl4 = 0x104, kDppRowShl8 = 0x108;
__device__ __forceinline__ double block_tree256_lane0(double v, double* s)
{
    const int t = threadIdx.x;
    s[t] = v;
    __syncthreads();
    if (t < 128) s[t] = s[t] + s[t + 128];
    __syncthreads();
    double r = 0.0;
    if (t < 64) {                                   // wave 0, all lanes active
        r = s[t] + s[t + 64];
        r = r + __shfl_down(r, 32, 64);
        r = r + __shfl_down(r, 16, 64);
        r = r + dpp_f64<kDppRowShl8>(r);            // lane t of a row reads lane t + 8 of the same row
        r = r + dpp_f64<kDppRowShl4>(r);
        r = r + dpp_f64<kDppRowShl2>(r);
        r = r + dpp_f64<kDppRowShl1>(r);
    }
    return r;
}

// canonical U[x] from the chunk partials of the previous update (n_prev = n + 1 slots)
__device__ __forceinline__ double finish_ux(const double* __restrict__ xpart, int64_t n_prev, double* s)
{
    const int64_t nchunk = (n_prev + kThreads - 1) / kThreads;
    double acc = 0.0;
    for (int64_t c = threadIdx.x; c < nchunk; c += kThreads) acc += xpart[c];
    return block_tree256(acc, s);
}
// The same value in every thread with three barriers instead of nine (the scan kernels run this prologue in each of their
// 1 024 - 2 048 blocks): block_tree256_lane0 adds the same operand pairs at every level, thread 0 hands the sum out through
// s[255], which nobody reads inside the tree after its second barrier.  `s` may be reused after the call.
__device__ __forceinline__ double finish_ux_bcast(const double* __restrict__ xpart, int64_t n_prev, double* s)
{
    const int64_t nchunk = (n_prev + kThreads - 1) / kThreads;
    double acc = 0.0;
    for (int64_t c = threadIdx.x; c < nchunk; c += kThreads) acc += xpart[c];
    const double r = block_tree256_lane0(acc, s);
    if (threadIdx.x == 0) s[kThreads - 1] = r;
    __syncthreads();
    return s[kThreads - 1];
}

// the same with the thread's first chunk partial already loaded (xp0 = xpart[threadIdx.x], issued by the caller together with
// its other first loads; same additions in the same order)
__device__ __forceinline__ double finish_ux_bcast_pre(double xp0, const double* __restrict__ xpart, int64_t n_prev, double* s)
{
    const int64_t nchunk = (n_prev + kThreads - 1) / kThreads;
    double acc = 0.0;
    if ((int64_t)threadIdx.x < nchunk) acc += xp0;
    for (int64_t c = threadIdx.x + kThreads; c < nchunk; c += kThreads) acc += xpart[c];
    const double r = block_tree256_lane0(acc, s);
    if (threadIdx.x == 0) s[kThreads - 1] = r;
    __syncthreads();
    return s[kThreads - 1];
}

// block-wide reduction of `cnt` records to the winner (q, key, d); result in every thread
__device__ __forceinline__ void reduce_records(const NjRecord* __restrict__ recs, int cnt, double& bq,
                                               uint64_t& bk, double& bd, double* sq, uint64_t* sk,
                                               double* sdd)
{
    bq = 10000.0; bk = ~0ull; bd = 0.0;
    for (int i = threadIdx.x; i < cnt; i += kThreads) {
        const double q = recs[i].q;
        const uint64_t k = recs[i].key;
        if ((q < bq) | ((q == bq) & (k < bk))) { bq = q; bk = k; bd = recs[i].d; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oq = __shfl_down(bq, off, 64);
        const uint64_t ok = __shfl_down((unsigned long long)bk, off, 64);
        const double od = __shfl_down(bd, off, 64);
        if ((oq < bq) | ((oq == bq) & (ok < bk))) { bq = oq; bk = ok; bd = od; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sq[w] = bq; sk[w] = bk; sdd[w] = bd; }
    __syncthreads();
    bq = sq[0]; bk = sk[0]; bd = sdd[0];
#pragma unroll
    for (int i = 1; i < kThreads / 64; ++i)
        if ((sq[i] < bq) | ((sq[i] == bq) & (sk[i] < bk))) { bq = sq[i]; bk = sk[i]; bd = sdd[i]; }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Q-argmin scan of a group of rows over one 512-column strip (shared by nj.hip and njs.hip; see the scan kernels)
// ------------------------------------------------------------------------------------------------
// Row view of the one-exchange sharded path (njs.hip): the rows of the two slots rewritten by the previous merge live in
// row buffers (xrow, yrow) until the next update flushes them into the matrix; VIEW = false: every row is in D.
struct RowView {
    int64_t xp = -1, yp = -1;                 // slots whose authoritative row is xrow / yrow (-1: none)
    const double* xrow = nullptr;
    const double* yrow = nullptr;
};

template <bool DIAG, bool NT, bool FILT, bool VIEW>
__device__ __forceinline__ void scan_rows(const double* __restrict__ D, int64_t ld,
                                          const double* __restrict__ Ur,
                                          const uint64_t* __restrict__ KA, const uint64_t* __restrict__ KB,
                                          int64_t a0, int64_t l0,
                                          int nrows, int64_t c0, int64_t xprev, double urx, double ub0,
                                          double ub1, uint64_t ka0, uint64_t ka1, uint64_t kb0,
                                          uint64_t kb1, double& bq, uint64_t& bk, const RowView rv)
{
    const int tid = threadIdx.x;
    const int64_t b0 = c0 + 2 * tid, b1 = b0 + 1;
    const v2d* base = reinterpret_cast<const v2d*>(D + l0 * ld + c0) + tid;
    const int64_t ld2 = ld >> 1;
    for (int r = 0; r < nrows; r += 8) {
        v2d v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rr = min(r + u, nrows - 1);  // clamp: duplicates are idempotent
            const v2d* p = base + (int64_t)rr * ld2;
            if (VIEW) {      // (wave-uniform: the row index does not depend on the lane)
                const int64_t ar = a0 + rr;
                if (ar == rv.xp) p = reinterpret_cast<const v2d*>(rv.xrow + c0) + tid;
                else if (ar == rv.yp) p = reinterpret_cast<const v2d*>(rv.yrow + c0) + tid;
            }
            if (DIAG) p = (b0 < a0 + rr) ? p : p - tid;  // masked lanes share one line
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t a = a0 + min(r + u, nrows - 1);
            const double ua = (a == xprev) ? urx : Ur[a];
            const uint64_t kaa = KA[a];
            const uint64_t kba = KB ? KB[a] : nj_key_b(a);   // position-space matrices carry their slot keys
            double d0 = v[u].x, d1 = v[u].y;
            if (DIAG) {
                d0 = (b0 < a) ? d0 : __builtin_nan("");
                d1 = (b1 < a) ? d1 : __builtin_nan("");
            }
            const double q0a = (d0 - ua) - ub0, q0b = (d0 - ub0) - ua;
            const double q1a = (d1 - ua) - ub1, q1b = (d1 - ub1) - ua;
            if (FILT) {
                // a lane's best improves O(log m) times over m elements: test once per row, update rarely
                const double m = fmin(fmin(q0a, q0b), fmin(q1a, q1b));  // fmin drops NaN (masked / invalid)
                if (!(m <= bq)) continue;
            }
            best_update(bq, bk, q0a, kaa | kb0);
            best_update(bq, bk, q0b, ka0 | kba);
            best_update(bq, bk, q1a, kaa | kb1);
            best_update(bq, bk, q1b, ka1 | kba);
        }
    }
}

// strip geometry for active size n on (rank, world): first owned local row that can see column c0,
// rounded down to a row group, and the number of row groups below it
template <int RG>
__device__ __forceinline__ void strip_geom(int64_t cb, int64_t n, int64_t nloc, int rank, int world,
                                           int64_t& lstart, int& cnt)
{
    const int64_t c0 = cb * kTileCols;
    const int64_t lmin = shard_rows(min(c0 + 1, n), rank, world);  // owned rows with global index <= c0
    lstart = lmin / RG * RG;
    cnt = nloc > lstart ? (int)((nloc - lstart + RG - 1) / RG) : 0;
}

// Exclusive prefix of the strips' unit counts into LDS, pref[0 .. nstrips], pref[nstrips] = total: ONE barrier.  Wave 0
// alone does it (lane l: strips l * per .. l * per + per - 1, a wave-level scan of the lanes' sums); round 3's version was a
// 256-thread Hillis-Steele scan through LDS -- 17 barriers in every block of the scan grid, ~10 % of a rank's scan time
// once a rank streams only an eighth of the triangle.
template <int RG>
__device__ __forceinline__ void strip_prefix_lds(int32_t* pref, int nstrips, int64_t n, int64_t nloc, int rank, int world)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int per = (nstrips + 63) / 64;
        int mysum = 0;
        for (int k = 0; k < per; ++k) {
            const int cb = lane * per + k;
            if (cb < nstrips) { int64_t ls; int c; strip_geom<RG>(cb, n, nloc, rank, world, ls, c); mysum += c; }
        }
        int incl = mysum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        int run = incl - mysum;
        for (int k = 0; k < per; ++k) {
            const int cb = lane * per + k;
            if (cb < nstrips) { int64_t ls; int c; strip_geom<RG>(cb, n, nloc, rank, world, ls, c); pref[cb] = run; run += c; }
        }
        if (lane == 63) pref[nstrips] = incl;
    }
    __syncthreads();
}


// host part of the reference's loop (src/neighborJoining.cu:219-239): branch lengths, merge log, state
__device__ __forceinline__ void commit_merge(NjState* __restrict__ st, const double* __restrict__ U,
                                             int64_t n, int64_t it, int64_t x, int64_t y, double d, double q,
                                             int32_t* __restrict__ log_x, int32_t* __restrict__ log_y,
                                             double* __restrict__ log_bx, double* __restrict__ log_by)
{
    const double r = (double)(n - 2);
    double blX = (d + U[x] / r - U[y] / r) * 0.5;
    double blY = d - blX;
    if (blX < 0) { blY += blX; blX = 0; }
    if (blY < 0) { blX += blY; blY = 0; }
    log_x[it] = (int32_t)x; log_y[it] = (int32_t)y; log_bx[it] = blX; log_by[it] = blY;
    st->x = (int32_t)x; st->y = (int32_t)y; st->d = d; st->q = q;
    st->n = n - 1; st->it = it + 1;
}


}  // namespace dpr
