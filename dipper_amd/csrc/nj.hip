// Neighbor joining on gfx950: row sums, per-iteration Q-argmin over the strict lower triangle,
// matrix/row-sum update.  Replaces src/neighborJoining.cu (calculateU :94-115, findMinDist
// :117-148, thrust::min_element :214, host bookkeeping :219-239, updateDisMatrix :161-194) of the
// reference with a device-resident loop: no host round trip per iteration.
//
// Data layout in HBM: D is row-major fp64, full symmetric, row stride ld (multiple of 16 doubles =
// 128 B).  A rank stores the rows it owns (block-cyclic, 64 rows per block) at full width.
// The argmin kernel streams only columns j < i of every owned row (algorithmic bytes 4n^2+4n).
//
// All fp64 expressions are written exactly as the reference associates them; this TU is compiled
// with -ffp-contract=off.
#include "dpr_internal.hpp"

namespace dpr {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------------
// matrix source: packed lower triangle (MatrixReader, src/matrix_reader.cu:23-45 + fillDismatrix
// src/neighborJoining.cu:20-32).  Row i of the packed array starts at i(i-1)/2.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void nj_expand_lower_kernel(const double* __restrict__ P,
                                                                   double* __restrict__ D,
                                                                   int64_t ld, int64_t N,
                                                                   int64_t rows_local, int rank,
                                                                   int world)
{
    const int64_t li = blockIdx.y;
    if (li >= rows_local) return;
    const int64_t i = shard_global_row(li, rank, world);
    for (int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x; j < N;
         j += (int64_t)gridDim.x * kThreads) {
        double v;
        if (j == i) v = 0.0;
        else if (j < i) v = P[i * (i - 1) / 2 + j];
        else v = P[j * (j - 1) / 2 + i];
        D[li * ld + j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// calculateU (src/neighborJoining.cu:94-115): 256 strided class partials (j == t mod 256,
// ascending j, j != i) combined by the canonical pairwise tree.  One block per owned row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void nj_row_sums_kernel(const double* __restrict__ D,
                                                               int64_t ld, int64_t N,
                                                               int64_t rows_local, int rank,
                                                               int world, double* __restrict__ U,
                                                               int local_index)
{
    __shared__ double s[kThreads];
    for (int64_t li = blockIdx.x; li < rows_local; li += gridDim.x) {
        const int64_t i = shard_global_row(li, rank, world);
        const double* row = D + li * ld;
        double acc = 0.0;
        for (int64_t j = threadIdx.x; j < N; j += kThreads)
            if (j != i) acc += row[j];
        const double tot = block_tree256(acc, s);
        if (threadIdx.x == 0) U[local_index ? li : i] = tot;
        __syncthreads();
    }
}

__global__ void nj_state_init_kernel(NjState* st, int64_t N)
{
    st->n = N; st->it = 0; st->x = 0; st->y = 0; st->d = 0.0; st->q = 0.0; st->status = 0; st->pad = 0;
}

// Ur[i] = U[i]/(n-2) (plain division, src/neighborJoining.cu:130,137) and the i-part of the key
__global__ __launch_bounds__(kThreads) void nj_prepare_kernel(const NjState* __restrict__ st,
                                                              const double* __restrict__ U,
                                                              double* __restrict__ Ur,
                                                              uint64_t* __restrict__ KA)
{
    const int64_t n = st->n;
    const double r = (double)(n - 2);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads) {
        Ur[i] = U[i] / r;
        KA[i] = nj_key_a(i, n);
    }
}

// ------------------------------------------------------------------------------------------------
// Q-argmin scan (findMinDist, src/neighborJoining.cu:117-148).
// Work unit = tile of 64 owned rows x 512 columns of the strict lower triangle; a lane owns two
// adjacent columns (one 16-byte load per row), eight rows in flight.  Each loaded D[a][b] (b<a)
// yields both ordered candidates of the reference:
//   (i=a,j=b): q = (D - Ur[a]) - Ur[b], key = KA[a] | KB[b]
//   (i=b,j=a): q = (D - Ur[b]) - Ur[a], key = KA[b] | KB[a]
// ------------------------------------------------------------------------------------------------
template <bool DIAG>
__device__ __forceinline__ void scan_tile(const double* __restrict__ D, int64_t ld,
                                          const double* __restrict__ Ur,
                                          const uint64_t* __restrict__ KA, int64_t g0, int64_t l0,
                                          int nrows, int64_t c0, double& bq, uint64_t& bk)
{
    const int tid = threadIdx.x;
    const int64_t b0 = c0 + 2 * tid, b1 = b0 + 1;
    const double ub0 = Ur[b0], ub1 = Ur[b1];
    const uint64_t ka0 = KA[b0], ka1 = KA[b1];
    const uint64_t kb0 = nj_key_b(b0), kb1 = nj_key_b(b1);
    const double2* base = reinterpret_cast<const double2*>(D + l0 * ld + c0) + tid;
    const int64_t ld2 = ld >> 1;

    for (int r = 0; r < nrows; r += 8) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rr = min(r + u, nrows - 1);  // clamp: duplicates are idempotent
            v[u] = base[(int64_t)rr * ld2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t a = g0 + min(r + u, nrows - 1);
            const double ua = Ur[a];
            const uint64_t kaa = KA[a];
            const uint64_t kba = nj_key_b(a);
            double d0 = v[u].x, d1 = v[u].y;
            if (DIAG) {
                d0 = (b0 < a) ? d0 : __builtin_nan("");
                d1 = (b1 < a) ? d1 : __builtin_nan("");
            }
            best_update(bq, bk, (d0 - ua) - ub0, kaa | kb0);
            best_update(bq, bk, (d0 - ub0) - ua, ka0 | kba);
            best_update(bq, bk, (d1 - ua) - ub1, kaa | kb1);
            best_update(bq, bk, (d1 - ub1) - ua, ka1 | kba);
        }
    }
}

template <bool PROBE>
__global__ __launch_bounds__(kThreads) void nj_scan_kernel(
    const double* __restrict__ D, int64_t ld, const NjState* __restrict__ st,
    const double* __restrict__ Ur, const uint64_t* __restrict__ KA,
    const int32_t* __restrict__ tile_start, int nlrb, int rank, int world,
    NjRecord* __restrict__ partials)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* ts = reinterpret_cast<int32_t*>(smem);
    __shared__ double sq[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64];

    const int tid = threadIdx.x;
    const int64_t n = st->n;
    double bq = 10000.0;  // the reference's init value (src/neighborJoining.cu:134)
    uint64_t bk = ~0ull;

    const int64_t nrb_glob = (n + kRowBlock - 1) / kRowBlock;
    int nact = nrb_glob > rank ? (int)((nrb_glob - rank + world - 1) / world) : 0;
    if (nact > nlrb) nact = nlrb;
    if (st->status != 0) nact = 0;
    for (int i = tid; i < nact; i += kThreads) ts[i] = tile_start[i];
    __syncthreads();

    int ntiles = 0;
    if (nact > 0) {
        const int64_t g0L = ((int64_t)(nact - 1) * world + rank) * kRowBlock;
        const int64_t gEndL = min(g0L + (int64_t)kRowBlock, n);
        ntiles = ts[nact - 1] + (int)((gEndL - 1 + kTileCols - 1) / kTileCols);
    }

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int lo = 0, hi = nact - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (ts[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const int lrb = __builtin_amdgcn_readfirstlane(lo);
        const int cb = __builtin_amdgcn_readfirstlane(t - ts[lo]);
        const int64_t g0 = ((int64_t)lrb * world + rank) * kRowBlock;
        const int64_t l0 = (int64_t)lrb * kRowBlock;
        const int nrows = (int)min((int64_t)kRowBlock, n - g0);
        const int64_t c0 = (int64_t)cb * kTileCols;
        if (c0 + kTileCols > g0)
            scan_tile<true>(D, ld, Ur, KA, g0, l0, nrows, c0, bq, bk);
        else
            scan_tile<false>(D, ld, Ur, KA, g0, l0, nrows, c0, bq, bk);
    }

    block_best(bq, bk, sq, sk);
    if (tid == 0) {
        NjRecord rec;
        rec.q = bq; rec.key = bk; rec.d = 0.0; rec.pad = 0;
        partials[blockIdx.x] = rec;
    }
}

// ------------------------------------------------------------------------------------------------
// select: reduce the per-block partials (thrust::min_element, src/neighborJoining.cu:214), fetch
// d = D[x][y] from the owned row, and (COMMIT) do the host part of the reference's loop
// (:219-239): branch lengths, merge log, state.
// ------------------------------------------------------------------------------------------------
template <bool COMMIT>
__global__ __launch_bounds__(kThreads) void nj_select_kernel(
    const double* __restrict__ D, int64_t ld, NjState* __restrict__ st,
    const double* __restrict__ U, const NjRecord* __restrict__ partials, int nparts, int rank,
    int world, int32_t* __restrict__ log_x, int32_t* __restrict__ log_y,
    double* __restrict__ log_bx, double* __restrict__ log_by, NjRecord* __restrict__ out)
{
    __shared__ double sq[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64];
    double bq = 10000.0;
    uint64_t bk = ~0ull;
    for (int i = threadIdx.x; i < nparts; i += kThreads) best_update(bq, bk, partials[i].q, partials[i].key);
    block_best(bq, bk, sq, sk);
    if (threadIdx.x != 0) return;
    if (st->status != 0) {
        if (out) { out->q = 10000.0; out->key = ~0ull; out->d = 0.0; out->pad = 0; }
        return;
    }
    if (bk == ~0ull) {
        if (COMMIT) st->status = 1;
        if (out) { out->q = bq; out->key = bk; out->d = 0.0; out->pad = 0; }
        return;
    }
    const int64_t i = (int64_t)(bk & 0xFFFFFFull), j = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t x = i < j ? i : j, y = i < j ? j : i;
    // the scan only visits owned rows a > b, so row y is local
    const double d = D[shard_local_row(y, world) * ld + x];
    if (out) { out->q = bq; out->key = bk; out->d = d; out->pad = 0; }
    if (COMMIT) {
        const int64_t n = st->n;
        const double r = (double)(n - 2);
        double blX = (d + U[x] / r - U[y] / r) * 0.5;
        double blY = d - blX;
        if (blX < 0) { blY += blX; blX = 0; }
        if (blY < 0) { blX += blY; blY = 0; }
        const int64_t it = st->it;
        log_x[it] = (int32_t)x; log_y[it] = (int32_t)y; log_bx[it] = blX; log_by[it] = blY;
        st->x = (int32_t)x; st->y = (int32_t)y; st->d = d; st->q = bq;
    }
}

// ------------------------------------------------------------------------------------------------
// updateDisMatrix (src/neighborJoining.cu:161-194), one thread per active slot i.
// world == 1: the three source vectors are rows x, y, n-1 of D.
// Also prepares Ur/KA of the next iteration (n' = n-1) and the 256-chunk partial sums of U[x].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void nj_update_kernel(double* __restrict__ D, int64_t ld,
                                                             const NjState* __restrict__ st,
                                                             double* __restrict__ U,
                                                             double* __restrict__ Ur,
                                                             uint64_t* __restrict__ KA,
                                                             double* __restrict__ xpart)
{
    __shared__ double s[kThreads];
    const int64_t n = st->n;
    if (st->status != 0) return;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if ((int64_t)blockIdx.x * kThreads >= n) return;  // whole block idle
    const int64_t x = st->x, y = st->y, last = n - 1;
    const double d = st->d;
    const int64_t n1 = n - 1;
    const double r1 = (double)(n1 - 2);

    double val = 0.0;
    if (i < n && i != x && i != y) {
        const double dxi = D[x * ld + i], dyi = D[y * ld + i];
        val = (dxi + dyi - d) * 0.5;
        if (i != last) {
            const double far = D[last * ld + i];
            const double u = U[i] + (-dxi - dyi + val);
            U[i] = u;
            Ur[i] = u / r1;
            D[x * ld + i] = val;
            D[i * ld + x] = val;
            D[y * ld + i] = far;
            D[i * ld + y] = far;
        } else {
            // tail of the reference (thread (0,0), :184-193)
            const double uy = U[last] + (-dxi - dyi + val);
            U[y] = uy;
            Ur[y] = uy / r1;
            D[x * ld + y] = val;
            D[y * ld + x] = val;
        }
    }
    if (i < n1) KA[i] = nj_key_a(i, n1);
    const double cs = block_tree256(val, s);
    if (threadIdx.x == 0) xpart[blockIdx.x] = cs;
}

// ------------------------------------------------------------------------------------------------
// world > 1.  Per iteration: local record -> all-gather -> commit (identical on every rank) ->
// column slices of x, y, n-1 for the owned rows -> all-gather -> update.
// ------------------------------------------------------------------------------------------------
__global__ void nj_commit_kernel(NjState* __restrict__ st, const double* __restrict__ U,
                                 const NjRecord* __restrict__ recs, int world,
                                 int32_t* __restrict__ log_x, int32_t* __restrict__ log_y,
                                 double* __restrict__ log_bx, double* __restrict__ log_by)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->status != 0) return;
    double bq = 10000.0, d = 0.0;
    uint64_t bk = ~0ull;
    for (int r = 0; r < world; ++r) {
        const double q = recs[r].q;
        const uint64_t k = recs[r].key;
        if ((q < bq) | ((q == bq) & (k < bk))) { bq = q; bk = k; d = recs[r].d; }
    }
    if (bk == ~0ull) { st->status = 1; return; }
    const int64_t i = (int64_t)(bk & 0xFFFFFFull), j = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t x = i < j ? i : j, y = i < j ? j : i;
    const int64_t n = st->n;
    const double r = (double)(n - 2);
    double blX = (d + U[x] / r - U[y] / r) * 0.5;
    double blY = d - blX;
    if (blX < 0) { blY += blX; blX = 0; }
    if (blY < 0) { blX += blY; blY = 0; }
    const int64_t it = st->it;
    log_x[it] = (int32_t)x; log_y[it] = (int32_t)y; log_bx[it] = blX; log_by[it] = blY;
    st->x = (int32_t)x; st->y = (int32_t)y; st->d = d; st->q = bq;
}

// slice[v][li] = D[li][c_v] for c = (x, y, n-1), owned rows with global index < n
__global__ __launch_bounds__(kThreads) void nj_extract_kernel(const double* __restrict__ D, int64_t ld,
                                                              const NjState* __restrict__ st,
                                                              double* __restrict__ slice,
                                                              int64_t slice_len, int64_t rows_local,
                                                              int rank, int world)
{
    if (st->status != 0) return;
    const int64_t li = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (li >= rows_local) return;
    const int64_t n = st->n;
    const int64_t i = shard_global_row(li, rank, world);
    if (i >= n) return;
    const double* row = D + li * ld;
    slice[0 * slice_len + li] = row[st->x];
    slice[1 * slice_len + li] = row[st->y];
    slice[2 * slice_len + li] = row[n - 1];
}

// gathered layout: gath[(r*3 + v)*slice_len + li]
__global__ __launch_bounds__(kThreads) void nj_update_sharded_kernel(
    double* __restrict__ D, int64_t ld, const NjState* __restrict__ st, double* __restrict__ U,
    double* __restrict__ Ur, uint64_t* __restrict__ KA, double* __restrict__ xpart,
    const double* __restrict__ gath, int64_t slice_len, int rank, int world)
{
    __shared__ double s[kThreads];
    const int64_t n = st->n;
    if (st->status != 0) return;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if ((int64_t)blockIdx.x * kThreads >= n) return;
    const int64_t x = st->x, y = st->y, last = n - 1;
    const double d = st->d;
    const int64_t n1 = n - 1;
    const double r1 = (double)(n1 - 2);
    const bool own_x = shard_owner(x, world) == rank, own_y = shard_owner(y, world) == rank;
    const int64_t lx = shard_local_row(x, world), ly = shard_local_row(y, world);

    double val = 0.0;
    if (i < n && i != x && i != y) {
        const int ro = shard_owner(i, world);
        const int64_t li = shard_local_row(i, world);
        const double* g = gath + (int64_t)ro * 3 * slice_len + li;
        const double dxi = g[0], dyi = g[slice_len];
        val = (dxi + dyi - d) * 0.5;
        if (i != last) {
            const double far = g[2 * slice_len];
            const double u = U[i] + (-dxi - dyi + val);
            U[i] = u;
            Ur[i] = u / r1;
            if (own_x) D[lx * ld + i] = val;
            if (own_y) D[ly * ld + i] = far;
            if (ro == rank) { D[li * ld + x] = val; D[li * ld + y] = far; }
        } else {
            const double uy = U[last] + (-dxi - dyi + val);
            U[y] = uy;
            Ur[y] = uy / r1;
            if (own_x) D[lx * ld + y] = val;
            if (own_y) D[ly * ld + x] = val;
        }
    }
    if (i < n1) KA[i] = nj_key_a(i, n1);
    const double cs = block_tree256(val, s);
    if (threadIdx.x == 0) xpart[blockIdx.x] = cs;
}

// initial row sums: Uloc[li] for owned rows (gathered by the caller), then U[i] = gathU[owner][li]
__global__ __launch_bounds__(kThreads) void nj_unpack_u_kernel(const double* __restrict__ gathU,
                                                               int64_t slice_len, int64_t N, int world,
                                                               double* __restrict__ U)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
    U[i] = gathU[(int64_t)shard_owner(i, world) * slice_len + shard_local_row(i, world)];
}

// U[x] = canonical sum of the chunk sums; advance the state to n-1.
__global__ __launch_bounds__(kThreads) void nj_finalize_kernel(NjState* __restrict__ st,
                                                               double* __restrict__ U,
                                                               double* __restrict__ Ur,
                                                               const double* __restrict__ xpart)
{
    __shared__ double s[kThreads];
    if (st->status != 0) return;
    const int64_t n = st->n;
    const int64_t nchunk = (n + kThreads - 1) / kThreads;
    double acc = 0.0;
    for (int64_t c = threadIdx.x; c < nchunk; c += kThreads) acc += xpart[c];
    const double ux = block_tree256(acc, s);
    if (threadIdx.x == 0) {
        const int64_t x = st->x;
        const int64_t n1 = n - 1;
        U[x] = ux;
        Ur[x] = ux / (double)(n1 - 2);
        st->n = n1;
        st->it = st->it + 1;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

int nj_alloc(NjBuffers& b, int64_t N, int rank, int world)
{
    nj_free(b);
    b.N = N; b.rank = rank; b.world = world;
    b.ld = round_up(N, 16);
    b.rows_local = shard_rows(N, rank, world);
    const int64_t rows_alloc = round_up(b.rows_local > 0 ? b.rows_local : 1, kRowBlock);
    const size_t dbytes = (size_t)(rows_alloc * b.ld + kTileCols + 16) * sizeof(double);
    DPR_HIP(hipMalloc(&b.D, dbytes));
    DPR_HIP(hipMemset(b.D, 0, dbytes));
    const size_t vec = (size_t)(N + kTileCols + 16);
    DPR_HIP(hipMalloc(&b.U, vec * sizeof(double)));
    DPR_HIP(hipMalloc(&b.Ur, vec * sizeof(double)));
    DPR_HIP(hipMalloc(&b.KA, vec * sizeof(uint64_t)));
    DPR_HIP(hipMemset(b.U, 0, vec * sizeof(double)));
    DPR_HIP(hipMemset(b.Ur, 0, vec * sizeof(double)));
    DPR_HIP(hipMemset(b.KA, 0, vec * sizeof(uint64_t)));
    DPR_HIP(hipMalloc(&b.partials, sizeof(NjRecord) * kScanBlocks));
    DPR_HIP(hipMalloc(&b.recs, sizeof(NjRecord) * (size_t)(world > 1 ? world : 1)));
    DPR_HIP(hipMalloc(&b.xpart, sizeof(double) * (size_t)((N + kThreads - 1) / kThreads + 1)));
    {
        // uniform slice length: local rows of rank 0 at n = N, padded to whole ownership blocks
        const int64_t nblk = (N + kRowBlock - 1) / kRowBlock;
        b.slice_len = ((nblk + world - 1) / world) * kRowBlock;
        if (world > 1) {
            DPR_HIP(hipMalloc(&b.slice, sizeof(double) * (size_t)(3 * b.slice_len)));
            DPR_HIP(hipMalloc(&b.gath, sizeof(double) * (size_t)(3 * b.slice_len * world)));
            DPR_HIP(hipMemset(b.slice, 0, sizeof(double) * (size_t)(3 * b.slice_len)));
            DPR_HIP(hipMemset(b.gath, 0, sizeof(double) * (size_t)(3 * b.slice_len * world)));
        }
    }
    DPR_HIP(hipMalloc(&b.st, sizeof(NjState)));
    DPR_HIP(hipMalloc(&b.log_x, sizeof(int32_t) * (size_t)(N + 1)));
    DPR_HIP(hipMalloc(&b.log_y, sizeof(int32_t) * (size_t)(N + 1)));
    DPR_HIP(hipMalloc(&b.log_bx, sizeof(double) * (size_t)(N + 1)));
    DPR_HIP(hipMalloc(&b.log_by, sizeof(double) * (size_t)(N + 1)));

    // tile prefix over owned row blocks (full, unclipped blocks)
    const int64_t nrb_glob = (N + kRowBlock - 1) / kRowBlock;
    b.nlrb = nrb_glob > rank ? (int32_t)((nrb_glob - rank + world - 1) / world) : 0;
    std::vector<int32_t> ts((size_t)b.nlrb + 1, 0);
    for (int l = 0; l < b.nlrb; ++l) {
        const int64_t g0 = ((int64_t)l * world + rank) * kRowBlock;
        const int64_t gEnd = g0 + kRowBlock < N ? g0 + kRowBlock : N;
        ts[(size_t)l + 1] = ts[(size_t)l] + (int32_t)((gEnd - 1 + kTileCols - 1) / kTileCols);
    }
    DPR_HIP(hipMalloc(&b.tile_start, sizeof(int32_t) * ts.size()));
    DPR_HIP(hipMemcpy(b.tile_start, ts.data(), sizeof(int32_t) * ts.size(), hipMemcpyHostToDevice));
    return DPR_OK;
}

void nj_free(NjBuffers& b)
{
    void* ptrs[] = { b.D, b.U, b.Ur, b.KA, b.partials, b.recs, b.xpart, b.gath, b.slice, b.st,
                     b.tile_start, b.log_x, b.log_y, b.log_bx, b.log_by };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    b = NjBuffers();
}

int nj_expand_lower(NjBuffers& b, const double* d_packed_lower, hipStream_t s)
{
    if (b.rows_local == 0) return DPR_OK;
    dim3 grid((unsigned)((b.N + kThreads - 1) / kThreads > 64 ? 64 : (b.N + kThreads - 1) / kThreads),
              (unsigned)b.rows_local);
    hipLaunchKernelGGL(nj_expand_lower_kernel, grid, dim3(kThreads), 0, s, d_packed_lower, b.D, b.ld,
                       b.N, b.rows_local, b.rank, b.world);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_init_sums(NjBuffers& b, hipStream_t s)
{
    if (b.rows_local > 0) {
        const unsigned grid = (unsigned)(b.rows_local < 4096 ? b.rows_local : 4096);
        // world > 1: sums of the owned rows go to slice[0..rows_local) and are all-gathered by the caller
        hipLaunchKernelGGL(nj_row_sums_kernel, dim3(grid), dim3(kThreads), 0, s, b.D, b.ld, b.N,
                           b.rows_local, b.rank, b.world, b.world > 1 ? b.slice : b.U, b.world > 1 ? 1 : 0);
        DPR_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(nj_state_init_kernel, dim3(1), dim3(1), 0, s, b.st, b.N);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_prepare(NjBuffers& b, hipStream_t s)
{
    const unsigned grid = (unsigned)((b.N + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(nj_prepare_kernel, dim3(grid), dim3(kThreads), 0, s, b.st, b.U, b.Ur, b.KA);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_launch_scan(NjBuffers& b, bool probe, hipStream_t s)
{
    const size_t lds = sizeof(int32_t) * (size_t)(b.nlrb + 1);
    if (probe)
        hipLaunchKernelGGL(nj_scan_kernel<true>, dim3(kScanBlocks), dim3(kThreads), lds, s, b.D, b.ld,
                           b.st, b.Ur, b.KA, b.tile_start, b.nlrb, b.rank, b.world, b.partials);
    else
        hipLaunchKernelGGL(nj_scan_kernel<false>, dim3(kScanBlocks), dim3(kThreads), lds, s, b.D, b.ld,
                           b.st, b.Ur, b.KA, b.tile_start, b.nlrb, b.rank, b.world, b.partials);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// commit = true: world == 1, reduce + commit.  commit = false: local record only, written to
// b.recs[b.rank] (all-gathered in place by the caller when world > 1).
int nj_launch_select(NjBuffers& b, bool commit, hipStream_t s)
{
    if (commit)
        hipLaunchKernelGGL(nj_select_kernel<true>, dim3(1), dim3(kThreads), 0, s, b.D, b.ld, b.st, b.U,
                           b.partials, kScanBlocks, b.rank, b.world, b.log_x, b.log_y, b.log_bx,
                           b.log_by, b.recs + b.rank);
    else
        hipLaunchKernelGGL(nj_select_kernel<false>, dim3(1), dim3(kThreads), 0, s, b.D, b.ld, b.st, b.U,
                           b.partials, kScanBlocks, b.rank, b.world, b.log_x, b.log_y, b.log_bx,
                           b.log_by, b.recs + b.rank);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_launch_unpack_u(NjBuffers& b, hipStream_t s)
{
    const unsigned grid = (unsigned)((b.N + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(nj_unpack_u_kernel, dim3(grid), dim3(kThreads), 0, s, b.gath, b.slice_len, b.N, b.world, b.U);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_launch_commit(NjBuffers& b, hipStream_t s)
{
    hipLaunchKernelGGL(nj_commit_kernel, dim3(1), dim3(64), 0, s, b.st, b.U, b.recs, b.world, b.log_x, b.log_y,
                       b.log_bx, b.log_by);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_launch_extract(NjBuffers& b, hipStream_t s)
{
    if (b.rows_local == 0) return DPR_OK;
    const unsigned grid = (unsigned)((b.rows_local + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(nj_extract_kernel, dim3(grid), dim3(kThreads), 0, s, b.D, b.ld, b.st, b.slice, b.slice_len,
                       b.rows_local, b.rank, b.world);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_launch_update(NjBuffers& b, hipStream_t s)
{
    const unsigned grid = (unsigned)((b.N + kThreads - 1) / kThreads);
    if (b.world > 1) {
        hipLaunchKernelGGL(nj_update_sharded_kernel, dim3(grid), dim3(kThreads), 0, s, b.D, b.ld, b.st, b.U, b.Ur,
                           b.KA, b.xpart, b.gath, b.slice_len, b.rank, b.world);
        DPR_HIP(hipGetLastError());
        return DPR_OK;
    }
    hipLaunchKernelGGL(nj_update_kernel, dim3(grid), dim3(kThreads), 0, s, b.D, b.ld, b.st, b.U, b.Ur,
                       b.KA, b.xpart);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_launch_finalize(NjBuffers& b, hipStream_t s)
{
    hipLaunchKernelGGL(nj_finalize_kernel, dim3(1), dim3(kThreads), 0, s, b.st, b.U, b.Ur, b.xpart);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

}  // namespace dpr
