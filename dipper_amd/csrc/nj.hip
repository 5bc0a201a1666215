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
#include "nj_dev.hpp"

#include <atomic>

namespace dpr {

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
    st->itb = 0; st->it_limit = 0; st->N = N;
    st->cnt_list[0] = 0; st->cnt_list[1] = 0; st->cnt_list[2] = 0; st->cnt_list[3] = 0; st->units_scanned = 0;
    st->pnew[0] = -1; st->pnew[1] = -1;
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
        KA[i] = nj_key_a_dev(i, n);
    }
}

// ------------------------------------------------------------------------------------------------
// Q-argmin scan (findMinDist, src/neighborJoining.cu:117-148).
// Work unit = RG owned rows x 512 columns of the strict lower triangle, enumerated strip-major
// (column strip cb, then row groups downwards); every block takes a CONTIGUOUS, equal share of the
// units, so it keeps its two columns' Ur/KA/KB in registers while it walks down a strip and all
// blocks finish together.  A lane owns two adjacent columns (one 16-byte non-temporal load per row),
// eight rows in flight; row parameters come through scalar loads.  Each loaded D[a][b] (b<a) yields
// both ordered candidates of the reference:
//   (i=a,j=b): q = (D - Ur[a]) - Ur[b], key = KA[a] | KB[b]
//   (i=b,j=a): q = (D - Ur[b]) - Ur[a], key = KA[b] | KB[a]
// Prologue (it > 0): the row sum of the node created in the previous iteration, U[x] = canonical sum
// of the 256-chunk partials left by the update kernel, is finished here by every block for itself
// (block 0 also stores it), which saves a kernel per iteration.
// ------------------------------------------------------------------------------------------------
template <bool PROBE, int RG, bool NT, bool FILT>
__global__ __launch_bounds__(kThreads) void nj_scan_kernel(
    const double* __restrict__ D, int64_t ld, const NjState* __restrict__ st, double* __restrict__ U_w,
    double* __restrict__ Ur_w, const double* __restrict__ Ur, const uint64_t* __restrict__ KA,
    const uint64_t* __restrict__ KB, const int32_t* __restrict__ pos_of_slot,
    const double* __restrict__ xpart, int64_t n, int64_t it, int rank, int world,
    NjRecord* __restrict__ partials)
{
    // Ur_w aliases Ur: block 0 stores the single element Ur[xprev] through it, and no block ever USES
    // Ur[xprev] read through the const pointer (every use substitutes urx), so the scan can keep
    // scalar loads for the row parameters.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* pref = reinterpret_cast<int32_t*>(smem);  // [nstrips + 1] exclusive prefix of unit counts
    __shared__ double sd[kThreads];
    __shared__ double sq[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64];

    const int tid = threadIdx.x;
    double bq = 10000.0;  // the reference's init value (src/neighborJoining.cu:134)
    uint64_t bk = ~0ull;
    const bool dead = st->status != 0;

    int64_t xprev = -1;
    double urx = 0.0;
    if (it > 0 && !dead) {
        xprev = st->x;
        const double ux = finish_ux_bcast(xpart, n + 1, sd);
        urx = ux / (double)(n - 2);
        if (blockIdx.x == 0 && tid == 0) { U_w[xprev] = ux; Ur_w[xprev] = urx; }
    }

    const int64_t nloc = shard_rows(n, rank, world);
    int nstrips = n > 1 ? (int)((n - 1 + kTileCols - 1) / kTileCols) : 0;
    if (dead) nstrips = 0;

    // unit counts per strip -> exclusive prefix in LDS (one barrier; nj_dev.hpp)
    strip_prefix_lds<RG>(pref, nstrips, n, nloc, rank, world);
    const int64_t utot = nstrips > 0 ? pref[nstrips] : 0;
    const int64_t ub = utot * blockIdx.x / gridDim.x, ue = utot * (blockIdx.x + 1) / gridDim.x;

    if (ub < ue) {
        int lo = 0, hi = nstrips - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pref[mid] <= (int)ub) lo = mid; else hi = mid - 1;
        }
        int cb = __builtin_amdgcn_readfirstlane(lo);
        int g = __builtin_amdgcn_readfirstlane((int)ub - pref[lo]);
        int64_t lstart; int cnt;
        strip_geom<RG>(cb, n, nloc, rank, world, lstart, cnt);
        bool fresh = true;
        double ub0 = 0, ub1 = 0; uint64_t ka0 = 0, ka1 = 0, kb0 = 0, kb1 = 0;
        for (int64_t u = ub; u < ue; ++u) {
            while (g >= cnt) { ++cb; g = 0; strip_geom<RG>(cb, n, nloc, rank, world, lstart, cnt); fresh = true; }
            const int64_t c0 = (int64_t)cb * kTileCols;
            if (fresh) {
                const int64_t b0 = c0 + 2 * tid;
                ub0 = (b0 == xprev) ? urx : Ur[b0];
                ub1 = (b0 + 1 == xprev) ? urx : Ur[b0 + 1];
                ka0 = KA[b0]; ka1 = KA[b0 + 1];
                kb0 = KB ? KB[b0] : nj_key_b(b0);
                kb1 = KB ? KB[b0 + 1] : nj_key_b(b0 + 1);
                fresh = false;
            }
            const int64_t l0 = lstart + (int64_t)g * RG;
            const int nrows = (int)min((int64_t)RG, nloc - l0);
            const int64_t a0 = shard_global_row(l0, rank, world);
            if (a0 < c0 + kTileCols)
                scan_rows<true, NT, FILT, false>(D, ld, Ur, KA, KB, a0, l0, nrows, c0, xprev, urx, ub0, ub1, ka0, ka1, kb0, kb1, bq, bk, RowView());
            else
                scan_rows<false, NT, FILT, false>(D, ld, Ur, KA, KB, a0, l0, nrows, c0, xprev, urx, ub0, ub1, ka0, ka1, kb0, kb1, bq, bk, RowView());
            ++g;
        }
    }

    block_best(bq, bk, sq, sk);
    if (tid == 0) {
        NjRecord rec;
        rec.q = bq; rec.key = bk; rec.d = 0.0; rec.pad = 0;
        if (bk != ~0ull) {
            // d = D[max][min]: the block only visited owned rows a > b, so the row is local
            const int64_t i = (int64_t)(bk & 0xFFFFFFull), j = (int64_t)((bk >> 24) & 0xFFFFFFull);
            int64_t x = i < j ? i : j, y = i < j ? j : i;
            if (pos_of_slot) {   // position space (single GPU): keys hold slots, the matrix is indexed by position
                const int64_t pi = pos_of_slot[i], pj = pos_of_slot[j];
                x = pi < pj ? pi : pj; y = pi < pj ? pj : pi;
                rec.pad = (uint64_t)pi | ((uint64_t)pj << 32);      // the pruned path's record format: position of key slot i | of key slot j << 32
            }
            rec.d = D[shard_local_row(y, world) * ld + x];
        }
        partials[blockIdx.x] = rec;
    }
}

// local winner of this rank -> recs[rank] (all-gathered when world > 1; read by the host for probes)
__global__ __launch_bounds__(kThreads) void nj_select_local_kernel(const NjState* __restrict__ st,
                                                                   const NjRecord* __restrict__ partials,
                                                                   int nparts, NjRecord* __restrict__ out)
{
    __shared__ double sq[kThreads / 64], sdd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64];
    double bq, bd; uint64_t bk;
    reduce_records(partials, nparts, bq, bk, bd, sq, sk, sdd);
    if (threadIdx.x != 0) return;
    if (st->status != 0) { bq = 10000.0; bk = ~0ull; bd = 0.0; }
    out->q = bq; out->key = bk; out->d = bd; out->pad = 0;
}

// ------------------------------------------------------------------------------------------------
// world == 1: select (thrust::min_element, src/neighborJoining.cu:214) + host bookkeeping (:219-239)
// + updateDisMatrix (:161-194) in ONE kernel.  Every block reduces the scan partials for itself;
// thread i handles slot i; the thread of the last slot also plays the reference's thread (0,0) tail
// and is the single writer of the merge log and of the state.  Prepares Ur/KA for n' = n-1 and the
// 256-chunk partial sums of U[x].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void nj_post_kernel(double* __restrict__ D, int64_t ld,
                                                           NjState* __restrict__ st, double* __restrict__ U,
                                                           double* __restrict__ Ur, uint64_t* __restrict__ KA,
                                                           double* __restrict__ xpart,
                                                           const NjRecord* __restrict__ partials, int nparts,
                                                           int64_t n, int64_t it, int32_t* __restrict__ log_x,
                                                           int32_t* __restrict__ log_y, double* __restrict__ log_bx,
                                                           double* __restrict__ log_by)
{
    __shared__ double s[kThreads];
    __shared__ double sq[kThreads / 64], sdd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64];
    if (st->status != 0) return;
    if ((int64_t)blockIdx.x * kThreads >= n) return;  // whole block idle
    double bq, d; uint64_t bk;
    reduce_records(partials, nparts, bq, bk, d, sq, sk, sdd);
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t last = n - 1;
    if (bk == ~0ull || !(bq < 10000.0)) {     // (q == 10000.0 is no candidate: the reference's strict `<` against its init value, src/neighborJoining.cu:134-141)
        if (i == last) st->status = 1;
        return;
    }
    const int64_t ki = (int64_t)(bk & 0xFFFFFFull), kj = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t x = ki < kj ? ki : kj, y = ki < kj ? kj : ki;
    const int64_t n1 = n - 1;
    const double r1 = (double)(n1 - 2);

    double val = 0.0;
    if (i == last) commit_merge(st, U, n, it, x, y, d, bq, log_x, log_y, log_bx, log_by);  // reads U[y] before the tail rewrites it
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
    if (i < n1) KA[i] = nj_key_a_dev(i, n1);
    const double cs = block_tree256(val, s);
    if (threadIdx.x == 0) xpart[blockIdx.x] = cs;
}

// ------------------------------------------------------------------------------------------------
// world > 1.  Per iteration: scan -> local record -> all-gather -> commit (identical on every rank)
// + column slices of x, y, n-1 for the owned rows -> all-gather -> sharded update.
// ------------------------------------------------------------------------------------------------
// slice[v][li] = D[li][c_v] for c = (x, y, n-1), owned rows with global index < n
__global__ __launch_bounds__(kThreads) void nj_commit_extract_kernel(
    const double* __restrict__ D, int64_t ld, NjState* __restrict__ st, const double* __restrict__ U,
    const NjRecord* __restrict__ recs, double* __restrict__ slice, int64_t slice_len, int64_t rows_local,
    int64_t n, int64_t it, int rank, int world, int32_t* __restrict__ log_x, int32_t* __restrict__ log_y,
    double* __restrict__ log_bx, double* __restrict__ log_by)
{
    if (st->status != 0) return;
    double bq = 10000.0, d = 0.0;
    uint64_t bk = ~0ull;
    for (int r = 0; r < world; ++r) {
        const double q = recs[r].q;
        const uint64_t k = recs[r].key;
        if ((q < bq) | ((q == bq) & (k < bk))) { bq = q; bk = k; d = recs[r].d; }
    }
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    if (bk == ~0ull || !(bq < 10000.0)) { if (writer) st->status = 1; return; }
    const int64_t ki = (int64_t)(bk & 0xFFFFFFull), kj = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t x = ki < kj ? ki : kj, y = ki < kj ? kj : ki;
    if (writer) commit_merge(st, U, n, it, x, y, d, bq, log_x, log_y, log_bx, log_by);
    const int64_t li = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (li >= rows_local) return;
    const int64_t i = shard_global_row(li, rank, world);
    if (i >= n) return;
    const double* row = D + li * ld;
    slice[0 * slice_len + li] = row[x];
    slice[1 * slice_len + li] = row[y];
    slice[2 * slice_len + li] = row[n - 1];
}

// gathered layout: gath[(r*3 + v)*slice_len + li]; x, y, d come from the state the commit wrote
__global__ __launch_bounds__(kThreads) void nj_update_sharded_kernel(
    double* __restrict__ D, int64_t ld, const NjState* __restrict__ st, double* __restrict__ U,
    double* __restrict__ Ur, uint64_t* __restrict__ KA, double* __restrict__ xpart,
    const double* __restrict__ gath, int64_t slice_len, int64_t n, int rank, int world)
{
    __shared__ double s[kThreads];
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
    if (i < n1) KA[i] = nj_key_a_dev(i, n1);
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

// after the last enqueued iteration: materialise U[x], Ur[x] (what the next scan's prologue would do),
// so that hooks and a resumed run see a consistent state
__global__ __launch_bounds__(kThreads) void nj_finish_kernel(const NjState* __restrict__ st,
                                                             double* __restrict__ U, double* __restrict__ Ur,
                                                             const double* __restrict__ xpart, int64_t n, int64_t it)
{
    __shared__ double s[kThreads];
    if (st->status != 0 || it <= 0) return;
    const double ux = finish_ux(xpart, n + 1, s);
    if (threadIdx.x == 0) {
        const int64_t x = st->x;
        U[x] = ux;
        Ur[x] = ux / (double)(n - 2);
    }
}

// ------------------------------------------------------------------------------------------------
// calibration: plain streaming read of the matrix allocation (16 B per lane, min-reduce), to know
// what a pure HBM read reaches on the same buffer the scan streams
// ------------------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(kThreads) void bw_read_kernel(const v2d* __restrict__ p, int64_t nvec, double* __restrict__ out)
{
    double m = 1e300;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + 7 * stride < nvec; i += 8 * stride) {
        v2d v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmin(m, fmin(v[u].x, v[u].y));
    }
    for (; i < nvec; i += stride) { const v2d v = p[i]; m = fmin(m, fmin(v.x, v.y)); }
    if (m == -1.2345e300) out[0] = m;  // never true: keeps the loads alive
}

int nj_bw_probe(const double* buf, int64_t cap, double* sink, int64_t bytes, int nt, int grid, int reps, hipStream_t s, hipEvent_t e0,
                hipEvent_t e1, float* ms)
{
    if (!buf || cap <= 0) { set_error("nj_bw_probe: no buffer"); return DPR_ERR_STATE; }
    if (bytes <= 0 || bytes > cap) bytes = cap;
    const int64_t nvec = bytes / 16;
    for (int r = -1; r < reps; ++r) {
        if (r == 0) DPR_HIP(hipEventRecord(e0, s));
        if (nt) hipLaunchKernelGGL(bw_read_kernel<true>, dim3(grid), dim3(kThreads), 0, s, reinterpret_cast<const v2d*>(buf), nvec, sink);
        else hipLaunchKernelGGL(bw_read_kernel<false>, dim3(grid), dim3(kThreads), 0, s, reinterpret_cast<const v2d*>(buf), nvec, sink);
    }
    DPR_HIP(hipEventRecord(e1, s));
    DPR_HIP(hipStreamSynchronize(s));
    DPR_HIP(hipEventElapsedTime(ms, e0, e1));
    *ms /= (float)reps;
    return DPR_OK;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// zero the parts of a [rows_alloc][ld] matrix buffer that the producers of the nrows x ncols block do not write: the
// columns [ncols, ld) of the rows below nrows, the rows [nrows, rows_alloc) and the tail pad behind the last row (the scan tiles may
// read them; what they hold never reaches a result -- the rows / columns are masked by index or by a NaN row sum --
// but a reused buffer then holds exactly what a fresh, zero-filled one does)
__global__ __launch_bounds__(kThreads) void nj_fill_pads_kernel(double* __restrict__ D, int64_t ld, int64_t nrows,
                                                                int64_t ncols, int64_t rows_alloc, int64_t tail, int diag)
{
    for (int64_t r = blockIdx.x; r < rows_alloc + 1; r += gridDim.x) {
        double* row = D + r * ld;
        const int64_t c0 = r < nrows ? ncols : 0, c1 = r < rows_alloc ? ld : tail;
        for (int64_t c = c0 + threadIdx.x; c < c1; c += kThreads) row[c] = 0.0;
        if (diag && r < nrows && r < ncols && threadIdx.x == 0) row[r] = 0.0;   // (the Mash pair kernel writes j < i and its mirror only)
    }
}

int nj_fill_pads(double* D, int64_t ld, int64_t nrows, int64_t ncols, int64_t rows_alloc, int64_t tail, bool diag, hipStream_t s)
{
    const int64_t g = rows_alloc + 1 < 4096 ? rows_alloc + 1 : 4096;
    hipLaunchKernelGGL(nj_fill_pads_kernel, dim3((unsigned)g), dim3(kThreads), 0, s, D, ld, nrows, ncols, rows_alloc, tail, diag ? 1 : 0);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// (all fills are ordered on the caller's stream: a plain hipMemset runs on the null stream, which a non-blocking
// stream does not wait for -- the fill of a 7 GB matrix would still be running when the first distance tiles land)
//
// A context that builds a matrix of the same shape again (bench.py's steps, a second dpr_dist_matrix) keeps every
// buffer: freeing and re-allocating 2 x 7.2 GB per call at 30 000 tips cost more than the distance kernel itself.
// Only the pads are zeroed then -- every element of the n x n block is overwritten by the distance kernels.
static int nj_alloc_inner(NjBuffers& b, int64_t N, int rank, int world, hipStream_t s, int64_t twin_rows);
int nj_alloc(NjBuffers& b, int64_t N, int rank, int world, hipStream_t s, int64_t twin_rows)
{
    // a failure part-way (out of memory after the matrix) must not leave a half-built NjBuffers behind: the reuse test
    // of the next call looks at b.D only (dpr_reserve_nj is called best-effort by the CLI, its return code ignored)
    const int rc = nj_alloc_inner(b, N, rank, world, s, twin_rows);
    if (rc != DPR_OK) nj_free(b);
    return rc;
}
static int nj_alloc_inner(NjBuffers& b, int64_t N, int rank, int world, hipStream_t s, int64_t twin_rows)
{
    const bool reuse = b.D != nullptr && b.N == N && b.rank == rank && b.world == world && b.twin_rows == twin_rows;
    if (!reuse) {
        nj_free(b);
        b.N = N; b.rank = rank; b.world = world;
        b.ld = round_up(N, 16);
        b.rows_local = shard_rows(N, rank, world);
        b.twin_rows = twin_rows;
    }
    // (+32 rows: the pruned path reuses this buffer for its odd epochs, whose row groups end up to 31 rows behind N)
    int64_t rows_alloc = round_up(b.rows_local > 0 ? b.rows_local : 1, kRowBlock) + 32;
    if (twin_rows > rows_alloc) rows_alloc = twin_rows;
    // (twin: two halves of this size in ONE allocation, njr.hip)
    const size_t hbytes = ((size_t)(rows_alloc * b.ld + kTileCols + 16) * sizeof(double) + 4095) / 4096 * 4096;
    const size_t dbytes = twin_rows > 0 ? 2 * hbytes : hbytes;
    b.half_bytes = hbytes;
    const size_t vec = (size_t)(N + kTileCols + 16);
    const int64_t nblk = (N + kRowBlock - 1) / kRowBlock;
    if (!reuse) {
        DPR_HIP(hipMalloc(&b.D, dbytes));
        DPR_HIP(hipMemsetAsync(b.D, 0, dbytes, s));
        DPR_HIP(hipMalloc(&b.U, vec * sizeof(double)));
        DPR_HIP(hipMalloc(&b.Ur, vec * sizeof(double)));
        DPR_HIP(hipMalloc(&b.KA, vec * sizeof(uint64_t)));
        DPR_HIP(hipMalloc(&b.partials, sizeof(NjRecord) * kScanBlocks));
        DPR_HIP(hipMalloc(&b.recs, sizeof(NjRecord) * (size_t)(world > 1 ? world : 1)));
        DPR_HIP(hipMalloc(&b.recs64, sizeof(NjsRec) * (size_t)(world > 1 ? world : 1)));
        DPR_HIP(hipMemsetAsync(b.recs64, 0, sizeof(NjsRec) * (size_t)(world > 1 ? world : 1), s));
        // (at least 512 entries: the pruned scan's new-row blocks load entry threadIdx.x before they know the chunk count)
        const size_t xcnt = (size_t)((N + kThreads - 1) / kThreads + 1);
        DPR_HIP(hipMalloc(&b.xpart, sizeof(double) * (xcnt < 512 ? 512 : xcnt)));
        DPR_HIP(hipMemsetAsync(b.xpart, 0, sizeof(double) * (xcnt < 512 ? 512 : xcnt), s));
        // uniform slice length: local rows of rank 0 at n = N, padded to whole ownership blocks
        b.slice_len = ((nblk + world - 1) / world) * kRowBlock;
        if (world > 1) {
            DPR_HIP(hipMalloc(&b.slice, sizeof(double) * (size_t)(3 * b.slice_len)));
            DPR_HIP(hipMalloc(&b.gath, sizeof(double) * (size_t)(3 * b.slice_len * world)));
        }
        DPR_HIP(hipMalloc(&b.st, sizeof(NjState)));
        DPR_HIP(hipMalloc(&b.log_x, sizeof(int32_t) * (size_t)(N + 1)));
        DPR_HIP(hipMalloc(&b.log_y, sizeof(int32_t) * (size_t)(N + 1)));
        DPR_HIP(hipMalloc(&b.log_bx, sizeof(double) * (size_t)(N + 1)));
        DPR_HIP(hipMalloc(&b.log_by, sizeof(double) * (size_t)(N + 1)));
    } else {
        njp_reset(b.pr);       // the pruned path's arena stays, its epoch state goes
        njr_free(b);           // (the row-sharded pruned path's epoch state; its window region and buffers stay with b.peer / b.D)
        // the rows [0, rows_local) x [0, N) are rewritten by the producers; zero what they leave alone
        if (int rc = nj_fill_pads(b.D, b.ld, b.rows_local, N, rows_alloc, kTileCols + 16, world == 1, s)) return rc;
    }
    DPR_HIP(hipMemsetAsync(b.U, 0, vec * sizeof(double), s));
    DPR_HIP(hipMemsetAsync(b.Ur, 0, vec * sizeof(double), s));
    DPR_HIP(hipMemsetAsync(b.KA, 0, vec * sizeof(uint64_t), s));
    DPR_HIP(hipMemsetAsync(b.partials, 0xff, sizeof(NjRecord) * kScanBlocks, s));  // key = ~0: "no candidate"
    if (world > 1) {
        DPR_HIP(hipMemsetAsync(b.slice, 0, sizeof(double) * (size_t)(3 * b.slice_len), s));
        DPR_HIP(hipMemsetAsync(b.gath, 0, sizeof(double) * (size_t)(3 * b.slice_len * world), s));
    }
    return DPR_OK;
}

void nj_free(NjBuffers& b)
{
    njr_free(b);
    njp_free(b.pr);
    njs_free_window(b);
    void* ptrs[] = { b.D, b.U, b.Ur, b.KA, b.partials, b.recs, b.recs64, b.xpart, b.gath, b.slice, b.st,
                     b.log_x, b.log_y, b.log_bx, b.log_by };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    const NjPeer keep = b.peer;      // (the plan and the poll limit are settings of the context, not of an allocation)
    b = NjBuffers();
    b.peer.plan = keep.plan; b.peer.poll_ticks = keep.poll_ticks;
    b.peer.fault_it = keep.fault_it; b.peer.fault_rank = keep.fault_rank;      // (dpr_ctx_set_debug_fault: a setting of the context too)
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

int nj_init_sums(NjBuffers& b, hipStream_t s, double* local_sums)
{
    if (b.rows_local > 0) {
        const unsigned grid = (unsigned)(b.rows_local < 4096 ? b.rows_local : 4096);
        // local_sums != null (several ranks): the sums of the own rows go to local_sums[0 .. rows_local) and are gathered
        // by the caller
        hipLaunchKernelGGL(nj_row_sums_kernel, dim3(grid), dim3(kThreads), 0, s, b.D, b.ld, b.N,
                           b.rows_local, b.rank, b.world, local_sums ? local_sums : b.U, local_sums ? 1 : 0);
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

// scan tuning knobs (dpr_scan_tune): row-group size, non-temporal loads, grid size
// (process-wide by the ABI's definition of dpr_scan_tune; atomics, so that a tuning call from one host thread and a launch
//  from another context's thread are at least well-defined -- results never depend on them)
static std::atomic<int> g_scan_rg{ 16 }, g_scan_nt{ 1 }, g_scan_grid{ 2048 };
void nj_scan_config(int rg, int nt, int grid) { g_scan_rg.store(rg, std::memory_order_relaxed); g_scan_nt.store(nt, std::memory_order_relaxed); g_scan_grid.store(grid, std::memory_order_relaxed); }
int nj_scan_grid() { const int g = g_scan_grid.load(std::memory_order_relaxed); return g > 0 ? g : 2048; }

template <bool PROBE>
static void scan_dispatch(NjBuffers& b, int64_t n, int64_t it, hipStream_t s)
{
    const int grid = nj_scan_grid();
    const size_t lds = sizeof(int32_t) * (size_t)((b.N + kTileCols - 1) / kTileCols + 2);
    // pruned mode keeps the matrix in position space with explicit slot keys (probe only)
    const NjPruned& q = b.pr;
    const bool posn = q.in_positions();
    const double* D = posn ? q.D : b.D;
    const int64_t ld = posn ? q.ld : b.ld;
    double *U = posn ? q.U : b.U, *Ur = posn ? q.Ur : b.Ur;
    const uint64_t *KA = posn ? q.KA : b.KA, *KB = posn ? q.KB : nullptr;
    const int32_t* pos = posn ? q.pos_of_slot : nullptr;
#define DPR_SCAN(RG, NT, FILT)                                                                                     \
    hipLaunchKernelGGL((nj_scan_kernel<PROBE, RG, NT, FILT>), dim3(grid), dim3(kThreads), lds, s, D, ld, b.st,     \
                       U, Ur, Ur, KA, KB, pos, b.xpart, n, it, b.rank, b.world, b.partials)
    const int rg_knob = g_scan_rg.load(std::memory_order_relaxed), nt_knob = g_scan_nt.load(std::memory_order_relaxed);
    const int rg = rg_knob & 127;
    const bool filt = (rg_knob & 128) != 0;   // bit 7 of the row-group knob selects the filtered update
    if (filt) {
        if (nt_knob) { if (rg == 64) DPR_SCAN(64, true, true); else DPR_SCAN(16, true, true); }
        else { if (rg == 64) DPR_SCAN(64, false, true); else DPR_SCAN(16, false, true); }
    } else {
        if (nt_knob) { if (rg == 64) DPR_SCAN(64, true, false); else DPR_SCAN(16, true, false); }
        else { if (rg == 64) DPR_SCAN(64, false, false); else DPR_SCAN(16, false, false); }
    }
#undef DPR_SCAN
}

int nj_launch_scan(NjBuffers& b, bool probe, int64_t n, int64_t it, hipStream_t s)
{
    if (probe) scan_dispatch<true>(b, n, it, s); else scan_dispatch<false>(b, n, it, s);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// world == 1: select + commit + update in one kernel
int nj_launch_post(NjBuffers& b, int64_t n, int64_t it, hipStream_t s)
{
    const unsigned grid = (unsigned)((n + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(nj_post_kernel, dim3(grid), dim3(kThreads), 0, s, b.D, b.ld, b.st, b.U, b.Ur, b.KA, b.xpart,
                       b.partials, nj_scan_grid(), n, it, b.log_x, b.log_y, b.log_bx, b.log_by);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// local record of this rank -> b.recs[b.rank]
int nj_launch_select_local(NjBuffers& b, int nparts, hipStream_t s)
{
    hipLaunchKernelGGL(nj_select_local_kernel, dim3(1), dim3(kThreads), 0, s, b.st, b.partials, nparts,
                       b.recs + b.rank);
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

// world > 1: reduce b.recs[world], commit, extract the column slices of x, y, n-1
int nj_launch_commit_extract(NjBuffers& b, int64_t n, int64_t it, hipStream_t s)
{
    const int64_t rows = b.rows_local > 0 ? b.rows_local : 1;
    const unsigned grid = (unsigned)((rows + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(nj_commit_extract_kernel, dim3(grid), dim3(kThreads), 0, s, b.D, b.ld, b.st, b.U, b.recs,
                       b.slice, b.slice_len, b.rows_local, n, it, b.rank, b.world, b.log_x, b.log_y, b.log_bx, b.log_by);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_launch_update_sharded(NjBuffers& b, int64_t n, hipStream_t s)
{
    const unsigned grid = (unsigned)((n + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(nj_update_sharded_kernel, dim3(grid), dim3(kThreads), 0, s, b.D, b.ld, b.st, b.U, b.Ur, b.KA,
                       b.xpart, b.gath, b.slice_len, n, b.rank, b.world);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int nj_launch_finish(NjBuffers& b, int64_t n, int64_t it, hipStream_t s)
{
    hipLaunchKernelGGL(nj_finish_kernel, dim3(1), dim3(kThreads), 0, s, b.st, b.U, b.Ur, b.xpart, n, it);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

}  // namespace dpr
