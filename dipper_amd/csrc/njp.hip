// Exact PRUNED neighbor joining (single GPU).  Same arithmetic, tie-breaking and merge log as the
// streaming path of nj.hip (and therefore as the reference), but the per-iteration Q-argmin no
// longer streams the whole triangle:
//
//  * the matrix lives in POSITION space: tips are permuted by ascending initial row sum, rows never
//    move afterwards (the reference's "last slot moves into y" becomes a relabel of slot_of_pos /
//    pos_of_slot); every key uses the reference's slot numbers, so ties resolve as in the reference;
//  * the triangle is cut into units of 16 rows x 512 columns with a lazily maintained lower bound
//    umin[unit] <= min D over the unit (exact whenever the unit is scanned, min-updated when the new
//    node's row/column is written);
//  * for a unit, every candidate obeys  q = fl(fl(D - Ur_a) - Ur_b) >= fl(fl(umin - rmax) - cmax)
//    (and the other association order) because fl(x - y) is monotone in x and -y; units whose bound
//    exceeds a known upper bound of the optimum (the best of the previous iteration's per-block
//    winners re-evaluated with the current row sums) cannot contain the winner NOR A TIE and are
//    skipped.  Sorting by row sum makes Ur homogeneous inside units, which is what makes the bound
//    tight.
//
// Per iteration: njp_post_kernel (select + merge + update, indexed by reference slot so that the
// canonical U[x] summation order is unchanged) -> njp_bounds_kernel (finish U[x], group maxima,
// seed bound) -> njp_scan_kernel (test all units, scan survivors, refresh their umin).
#include "nj_dev.hpp"

#include <algorithm>
#include <numeric>

namespace dpr {

constexpr int kUR = 16;  // rows per unit

__device__ __forceinline__ uint64_t enc_f64(double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_f64(uint64_t k)
{
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
static uint64_t enc_f64_host(double x)
{
    union { double d; uint64_t u; } c;
    c.d = x;
    return (c.u >> 63) ? ~c.u : (c.u | 0x8000000000000000ull);
}

// valid units: strip cb holds groups g >= 32*cb (row a = 16g.. can see column 512cb iff a > 512cb)
__host__ __device__ inline int64_t unit_prefix(int64_t cb, int64_t G16) { return cb * G16 - 16 * cb * (cb - 1); }
__host__ __device__ inline int64_t unit_total(int64_t P)
{
    const int64_t G16 = (P + kUR - 1) / kUR;
    int64_t S = (P - 1 + kTileCols - 1) / kTileCols;       // strips with at least one valid column
    while (S > 0 && G16 - 32 * (S - 1) <= 0) --S;
    return S > 0 ? unit_prefix(S, G16) : 0;
}

// ------------------------------------------------------------------------------------------------
// epoch build: B[a][b] = A[perm[a]][perm[b]]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void njp_permute_kernel(const double* __restrict__ A, int64_t lda,
                                                               double* __restrict__ B, int64_t ldb,
                                                               const int32_t* __restrict__ perm, int64_t P)
{
    for (int64_t a = blockIdx.y; a < P; a += gridDim.y) {
        const double* row = A + (int64_t)perm[a] * lda;
        for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < P; b += (int64_t)gridDim.x * kThreads)
            B[a * ldb + b] = row[perm[b]];
    }
}

__global__ __launch_bounds__(kThreads) void njp_init_vectors_kernel(const double* __restrict__ U_src,
                                                                    const int32_t* __restrict__ perm,
                                                                    const int32_t* __restrict__ slot_src,
                                                                    int64_t P, int64_t n, double* __restrict__ U,
                                                                    double* __restrict__ Ur, uint64_t* __restrict__ KA,
                                                                    uint64_t* __restrict__ KB,
                                                                    int32_t* __restrict__ slot_of_pos,
                                                                    int32_t* __restrict__ pos_of_slot)
{
    const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (p >= P) return;
    const int32_t src = perm[p];                    // position (or tip) in the previous epoch
    const int32_t slot = slot_src ? slot_src[src] : src;
    const double u = U_src[src];
    U[p] = u;
    Ur[p] = u / (double)(n - 2);
    KA[p] = nj_key_a(slot, n);
    KB[p] = nj_key_b(slot);
    slot_of_pos[p] = slot;
    pos_of_slot[slot] = (int32_t)p;
}

__global__ void njp_fill_u64_kernel(uint64_t* __restrict__ a, int64_t cnt, uint64_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) a[i] = v;
}

// ------------------------------------------------------------------------------------------------
// bounds: finish U[px] (canonical sum of the chunk partials), per-16-row and per-256 maxima of Ur,
// seed bound = best current q among the previous iteration's per-block winners (positions in pad)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void njp_bounds_kernel(const double* __restrict__ D, int64_t ld,
                                                              const NjState* __restrict__ st, double* __restrict__ U,
                                                              double* __restrict__ Ur, const double* __restrict__ xpart,
                                                              const NjRecord* __restrict__ partials, int nparts,
                                                              int64_t P, int64_t n, int64_t it,
                                                              double* __restrict__ gmax, double* __restrict__ bmax,
                                                              unsigned long long* __restrict__ seed)
{
    __shared__ double s[kThreads];
    __shared__ double smx[kThreads / 64];
    if (st->status != 0) return;
    int64_t px = -1;
    double ux = 0.0, urx = 0.0;
    if (it > 0) {
        px = st->pad;  // position of the node created by the previous merge
        ux = finish_ux(xpart, n + 1, s);
        urx = ux / (double)(n - 2);
    }
    const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    double ur = -__builtin_inf();
    if (p < P) {
        double v = (p == px) ? urx : Ur[p];
        if (p == px) { U[p] = ux; Ur[p] = urx; }
        if (v == v) ur = v;  // dead positions carry NaN
    }
    double m = ur;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 15) == 0 && p < P) gmax[p / kUR] = m;
#pragma unroll
    for (int off = 32; off >= 16; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) bmax[blockIdx.x] = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));

    // seed: one candidate per thread
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    double q = __builtin_inf();
    if (t < nparts && partials[t].key != ~0ull) {
        const uint64_t pp = partials[t].pad;
        const int64_t pa = (int64_t)(pp & 0xffffffffull), pb = (int64_t)(pp >> 32);  // pa > pb
        if (pa < P && pb < pa) {
        const double ua = (pa == px) ? urx : Ur[pa], ub = (pb == px) ? urx : Ur[pb];
        const double d = D[pa * ld + pb];
        const double q1 = (d - ua) - ub, q2 = (d - ub) - ua;
        q = fmin(q1, q2);        // NaN (dead) and inf drop out
        if (!(q == q)) q = __builtin_inf();
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q = fmin(q, __shfl_xor(q, off, 64));
    if ((threadIdx.x & 63) == 0 && q < __builtin_inf()) atomicMin(seed, (unsigned long long)enc_f64(q));
}

// ------------------------------------------------------------------------------------------------
// pruned scan.  Linear valid-unit index t -> (strip cb, group g); block b handles t = b, b+G, ...
// (interleaved, so that clustered survivors spread over the grid).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void unit_of(int64_t t, int64_t G16, int& cb, int64_t& g)
{
    // largest cb with unit_prefix(cb) <= t; prefix is concave increasing while counts stay positive
    int lo = 0, hi = (int)((G16 + 31) / 32);
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (unit_prefix(mid, G16) <= t && G16 - 32 * (int64_t)(mid - 1) > 0) lo = mid; else hi = mid - 1;
    }
    cb = lo;
    g = 32 * (int64_t)lo + (t - unit_prefix(lo, G16));
}

template <bool FULL>
__global__ __launch_bounds__(kThreads) void njp_scan_kernel(const double* __restrict__ D, int64_t ld,
                                                            const NjState* __restrict__ st,
                                                            const double* __restrict__ Ur,
                                                            const uint64_t* __restrict__ KA,
                                                            const uint64_t* __restrict__ KB,
                                                            const int32_t* __restrict__ pos_of_slot,
                                                            unsigned long long* __restrict__ umin,
                                                            const double* __restrict__ gmax,
                                                            const double* __restrict__ bmax,
                                                            const unsigned long long* __restrict__ seed,
                                                            int64_t P, int64_t utot,
                                                            NjRecord* __restrict__ partials,
                                                            unsigned long long* __restrict__ counters)
{
    __shared__ int32_t s_list[kThreads];
    __shared__ int32_t s_cnt;
    __shared__ double sq[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64];
    __shared__ double smin[kThreads / 64];

    const int tid = threadIdx.x;
    const int64_t G16 = (P + kUR - 1) / kUR;
    const int64_t nb256 = (P + kThreads - 1) / kThreads;
    double bq = 10000.0;  // the reference's init value
    uint64_t bk = ~0ull;
    const bool dead = st->status != 0;
    const unsigned long long seedv = *seed;
    const double bound = (FULL || seedv == ~0ull) ? __builtin_inf() : dec_f64(seedv);
    unsigned long long scanned = 0;

    for (int64_t base = blockIdx.x; base < utot && !dead; base += (int64_t)gridDim.x * kThreads) {
        // ---- test up to 256 units of this block, one per lane
        const int64_t t = base + (int64_t)tid * gridDim.x;
        bool keep = false;
        if (t < utot) {
            int cb; int64_t g;
            unit_of(t, G16, cb, g);
            const double u = dec_f64(umin[(int64_t)cb * G16 + g]);
            const double rm = gmax[g];
            const int64_t b2 = 2 * (int64_t)cb;
            const double cm = fmax(bmax[b2], b2 + 1 < nb256 ? bmax[b2 + 1] : -__builtin_inf());
            const double lb = fmin((u - rm) - cm, (u - cm) - rm);
            keep = FULL ? (rm > -__builtin_inf() && cm > -__builtin_inf()) : (lb <= bound);
        }
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        if (keep) { const int slot = atomicAdd(&s_cnt, 1); s_list[slot] = tid; }
        __syncthreads();
        const int cnt = s_cnt;
        scanned += (unsigned long long)cnt;
        // ---- scan the survivors (order inside the block is irrelevant for the result)
        for (int e = 0; e < cnt; ++e) {
            const int64_t tu = base + (int64_t)s_list[e] * gridDim.x;
            int cb; int64_t g;
            unit_of(tu, G16, cb, g);
            cb = __builtin_amdgcn_readfirstlane(cb);
            const int64_t g_s = (int64_t)__builtin_amdgcn_readfirstlane((int)g);
            const int64_t c0 = (int64_t)cb * kTileCols, a0 = g_s * kUR;
            const int nrows = (int)min((int64_t)kUR, P - a0);
            const int64_t b0 = c0 + 2 * tid, b1 = b0 + 1;
            const double ub0 = Ur[b0], ub1 = Ur[b1];
            const uint64_t ka0 = KA[b0], ka1 = KA[b1], kb0 = KB[b0], kb1 = KB[b1];
            const v2d* basep = reinterpret_cast<const v2d*>(D + a0 * ld + c0) + tid;
            const int64_t ld2 = ld >> 1;
            const bool diag = a0 < c0 + kTileCols;
            double m = __builtin_inf();
            for (int r = 0; r < nrows; r += 8) {
                v2d v[8];
#pragma unroll
                for (int u8 = 0; u8 < 8; ++u8) {
                    const int rr = min(r + u8, nrows - 1);
                    const v2d* pp = basep + (int64_t)rr * ld2;
                    if (diag) pp = (b0 < a0 + rr) ? pp : pp - tid;
                    v[u8] = __builtin_nontemporal_load(pp);
                }
#pragma unroll
                for (int u8 = 0; u8 < 8; ++u8) {
                    const int64_t a = a0 + min(r + u8, nrows - 1);
                    const double ua = Ur[a];
                    const uint64_t kaa = KA[a], kba = KB[a];
                    double d0 = v[u8].x, d1 = v[u8].y;
                    if (diag) {
                        d0 = (b0 < a) ? d0 : __builtin_nan("");
                        d1 = (b1 < a) ? d1 : __builtin_nan("");
                    }
                    m = fmin(m, fmin(d0, d1));  // fmin drops the NaN of masked entries; dead entries hold +inf
                    best_update(bq, bk, (d0 - ua) - ub0, kaa | kb0);
                    best_update(bq, bk, (d0 - ub0) - ua, ka0 | kba);
                    best_update(bq, bk, (d1 - ua) - ub1, kaa | kb1);
                    best_update(bq, bk, (d1 - ub1) - ua, ka1 | kba);
                }
            }
            // exact unit minimum -> umin
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_xor(m, off, 64));
            if ((tid & 63) == 0) smin[tid >> 6] = m;
            __syncthreads();
            if (tid == 0) umin[(int64_t)cb * G16 + g_s] = enc_f64(fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3])));
            __syncthreads();
        }
    }

    block_best(bq, bk, sq, sk);
    if (tid == 0) {
        NjRecord rec;
        rec.q = bq; rec.key = bk; rec.d = 0.0; rec.pad = 0;
        if (bk != ~0ull) {
            const int64_t i = (int64_t)(bk & 0xFFFFFFull), j = (int64_t)((bk >> 24) & 0xFFFFFFull);
            const int64_t pi = pos_of_slot[i], pj = pos_of_slot[j];
            const int64_t pa = pi > pj ? pi : pj, pb = pi > pj ? pj : pi;
            rec.d = D[pa * ld + pb];
            rec.pad = (uint64_t)pa | ((uint64_t)pb << 32);
        }
        partials[blockIdx.x] = rec;
        if (counters && scanned) atomicAdd(counters, scanned);
    }
}

// ------------------------------------------------------------------------------------------------
// select + merge + update, indexed by REFERENCE slot i (so the chunk sums of U[x] keep the canonical
// order).  Position space: rows never move; the node of slot n-1 is relabelled to slot y.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void njp_post_kernel(double* __restrict__ D, int64_t ld,
                                                            NjState* __restrict__ st, double* __restrict__ U,
                                                            double* __restrict__ Ur, uint64_t* __restrict__ KA,
                                                            uint64_t* __restrict__ KB,
                                                            int32_t* __restrict__ slot_of_pos,
                                                            int32_t* __restrict__ pos_of_slot,
                                                            unsigned long long* __restrict__ umin,
                                                            unsigned long long* __restrict__ seed,
                                                            double* __restrict__ xpart,
                                                            const NjRecord* __restrict__ partials, int nparts,
                                                            int64_t P, int64_t n, int64_t it,
                                                            int32_t* __restrict__ log_x, int32_t* __restrict__ log_y,
                                                            double* __restrict__ log_bx, double* __restrict__ log_by)
{
    __shared__ double s[kThreads];
    __shared__ double sq[kThreads / 64], sdd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64];
    if (st->status != 0) return;
    double bq, d; uint64_t bk;
    reduce_records(partials, nparts, bq, bk, d, sq, sk, sdd);
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t last = n - 1;
    if (bk == ~0ull) {
        if (i == last) st->status = 1;
        return;
    }
    const int64_t ki = (int64_t)(bk & 0xFFFFFFull), kj = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t x = ki < kj ? ki : kj, y = ki < kj ? kj : ki;
    const int64_t px = pos_of_slot[x], py = pos_of_slot[y];
    const int64_t n1 = n - 1;
    const double r1 = (double)(n1 - 2);
    const int64_t G16 = (P + kUR - 1) / kUR;
    const double INF = __builtin_inf();

    double val = 0.0;
    if (i < n) {
        const int64_t p = pos_of_slot[i];
        if (i == last) {
            // single writer of the log and the state (reads U[px], U[py] before anything rewrites them)
            const double r = (double)(n - 2);
            double blX = (d + U[px] / r - U[py] / r) * 0.5;
            double blY = d - blX;
            if (blX < 0) { blY += blX; blX = 0; }
            if (blY < 0) { blX += blY; blY = 0; }
            log_x[it] = (int32_t)x; log_y[it] = (int32_t)y; log_bx[it] = blX; log_by[it] = blY;
            st->x = (int32_t)x; st->y = (int32_t)y; st->d = d; st->q = bq;
            st->n = n1; st->it = it + 1; st->pad = (int32_t)px;
            *seed = ~0ull;  // +inf-most encoding: no bound until njp_bounds_kernel finds one
        }
        int64_t new_slot = i;
        if (i != x && i != y) {
            const double dxi = D[px * ld + p], dyi = D[py * ld + p];
            val = (dxi + dyi - d) * 0.5;
            const double u = U[p] + (-dxi - dyi + val);   // i == last: "U[y] = U[last] + ..." of the reference's tail
            U[p] = u;
            Ur[p] = u / r1;
            D[px * ld + p] = val;
            D[p * ld + px] = val;
            D[py * ld + p] = INF;
            D[p * ld + py] = INF;
            // the unit holding the pair (px, p) may have a new minimum
            const int64_t pa = p > px ? p : px, pb = p > px ? px : p;
            atomicMin(&umin[(pb / kTileCols) * G16 + pa / kUR], (unsigned long long)enc_f64(val));
            if (i == last) {           // relabel: the node of the last slot now lives in slot y
                new_slot = y;
                slot_of_pos[p] = (int32_t)y;
                pos_of_slot[y] = (int32_t)p;
            }
        } else if (i == y) {
            Ur[p] = __builtin_nan("");   // dead; U[p] is left alone (the writer thread may still be reading it)
            D[px * ld + py] = INF;
            D[py * ld + px] = INF;
            if (y != last) slot_of_pos[p] = -1;
            new_slot = -1;
        }
        if (new_slot >= 0) { KA[p] = nj_key_a(new_slot, n1); KB[p] = nj_key_b(new_slot); }
    }
    const double cs = block_tree256(val, s);
    if (threadIdx.x == 0) xpart[blockIdx.x] = cs;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int g_njp_grid = 2048;
int njp_scan_grid() { return g_njp_grid; }

static int64_t round_up16(int64_t v) { return (v + 15) / 16 * 16; }

int njp_build(NjBuffers& b, hipStream_t s)
{
    // b.D / b.U hold the matrix and the row sums in tip order (world == 1).  Sort by U ascending.
    const int64_t N = b.N;
    std::vector<double> hU((size_t)N);
    DPR_HIP(hipMemcpyAsync(hU.data(), b.U, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, s));
    DPR_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> perm((size_t)N);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t c) {
        const double ua = hU[(size_t)a], uc = hU[(size_t)c];
        if (ua != ua) return false;      // NaN last
        if (uc != uc) return true;
        return ua < uc;
    });
    NjPruned& q = b.pr;
    q.P = N;
    q.ld = round_up16(N);
    const int64_t rows_alloc = (N + kUR - 1) / kUR * kUR + kUR;
    const size_t dbytes = (size_t)(rows_alloc * q.ld + kTileCols + 16) * sizeof(double);
    DPR_HIP(hipMalloc(&q.D, dbytes));
    DPR_HIP(hipMemsetAsync(q.D, 0, dbytes, s));
    const size_t vec = (size_t)(N + kTileCols + 16);
    DPR_HIP(hipMalloc(&q.U, vec * sizeof(double)));
    DPR_HIP(hipMalloc(&q.Ur, vec * sizeof(double)));
    DPR_HIP(hipMalloc(&q.KA, vec * sizeof(uint64_t)));
    DPR_HIP(hipMalloc(&q.KB, vec * sizeof(uint64_t)));
    DPR_HIP(hipMemsetAsync(q.U, 0, vec * sizeof(double), s));
    DPR_HIP(hipMemsetAsync(q.Ur, 0xff, vec * sizeof(double), s));   // NaN beyond P
    DPR_HIP(hipMemsetAsync(q.KA, 0, vec * sizeof(uint64_t), s));
    DPR_HIP(hipMemsetAsync(q.KB, 0, vec * sizeof(uint64_t), s));
    DPR_HIP(hipMalloc(&q.slot_of_pos, sizeof(int32_t) * vec));
    DPR_HIP(hipMalloc(&q.pos_of_slot, sizeof(int32_t) * vec));
    DPR_HIP(hipMalloc(&q.perm, sizeof(int32_t) * vec));
    DPR_HIP(hipMemcpyAsync(q.perm, perm.data(), sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, s));
    const int64_t G16 = (N + kUR - 1) / kUR, S = (N + kTileCols - 1) / kTileCols + 1;
    q.nunits_alloc = S * G16;
    DPR_HIP(hipMalloc(&q.umin, sizeof(uint64_t) * (size_t)q.nunits_alloc));
    DPR_HIP(hipMalloc(&q.gmax, sizeof(double) * (size_t)(G16 + 16)));
    DPR_HIP(hipMalloc(&q.bmax, sizeof(double) * (size_t)((N + kThreads - 1) / kThreads + 2)));
    DPR_HIP(hipMalloc(&q.seed, sizeof(uint64_t) * 2));
    q.counters = q.seed + 1;
    const uint64_t init_seed[2] = { ~0ull, 0ull };
    DPR_HIP(hipMemcpyAsync(q.seed, init_seed, sizeof(init_seed), hipMemcpyHostToDevice, s));
    q.utot = unit_total(N);

    dim3 grid((unsigned)((N + kThreads - 1) / kThreads > 64 ? 64 : (N + kThreads - 1) / kThreads), (unsigned)(N < 32768 ? N : 32768));
    hipLaunchKernelGGL(njp_permute_kernel, grid, dim3(kThreads), 0, s, b.D, b.ld, q.D, q.ld, q.perm, N);
    hipLaunchKernelGGL(njp_init_vectors_kernel, dim3((unsigned)((N + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       b.U, q.perm, (const int32_t*)nullptr, N, N, q.U, q.Ur, q.KA, q.KB, q.slot_of_pos, q.pos_of_slot);
    hipLaunchKernelGGL(njp_fill_u64_kernel, dim3(256), dim3(256), 0, s, (uint64_t*)q.umin, q.nunits_alloc,
                       enc_f64_host(-__builtin_inf()));
    DPR_HIP(hipGetLastError());
    DPR_HIP(hipStreamSynchronize(s));
    // the tip-order matrix is no longer needed
    (void)hipFree(b.D);
    b.D = nullptr;
    q.active = true;
    return DPR_OK;
}

void njp_free(NjPruned& q)
{
    void* ptrs[] = { q.D, q.U, q.Ur, q.KA, q.KB, q.slot_of_pos, q.pos_of_slot, q.perm, q.umin, q.gmax, q.bmax, q.seed };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    q = NjPruned();
}

int njp_launch_bounds(NjBuffers& b, int64_t n, int64_t it, hipStream_t s)
{
    NjPruned& q = b.pr;
    const unsigned grid = (unsigned)((q.P + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(njp_bounds_kernel, dim3(grid), dim3(kThreads), 0, s, q.D, q.ld, b.st, q.U, q.Ur, b.xpart,
                       b.partials, njp_scan_grid(), q.P, n, it, q.gmax, q.bmax, (unsigned long long*)q.seed);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}


int njp_launch_scan(NjBuffers& b, bool full, hipStream_t s)
{
    NjPruned& q = b.pr;
    if (full)
        hipLaunchKernelGGL(njp_scan_kernel<true>, dim3(g_njp_grid), dim3(kThreads), 0, s, q.D, q.ld, b.st, q.Ur, q.KA,
                           q.KB, q.pos_of_slot, (unsigned long long*)q.umin, q.gmax, q.bmax,
                           (const unsigned long long*)q.seed, q.P, q.utot, b.partials, (unsigned long long*)q.counters);
    else
        hipLaunchKernelGGL(njp_scan_kernel<false>, dim3(g_njp_grid), dim3(kThreads), 0, s, q.D, q.ld, b.st, q.Ur, q.KA,
                           q.KB, q.pos_of_slot, (unsigned long long*)q.umin, q.gmax, q.bmax,
                           (const unsigned long long*)q.seed, q.P, q.utot, b.partials, (unsigned long long*)q.counters);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int njp_launch_post(NjBuffers& b, int64_t n, int64_t it, hipStream_t s)
{
    NjPruned& q = b.pr;
    const unsigned grid = (unsigned)((n + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(njp_post_kernel, dim3(grid), dim3(kThreads), 0, s, q.D, q.ld, b.st, q.U, q.Ur, q.KA, q.KB,
                       q.slot_of_pos, q.pos_of_slot, (unsigned long long*)q.umin, (unsigned long long*)q.seed, b.xpart,
                       b.partials, njp_scan_grid(), q.P, n, it, b.log_x, b.log_y, b.log_bx, b.log_by);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

}  // namespace dpr
