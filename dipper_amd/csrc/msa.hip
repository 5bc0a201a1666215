// Aligned-sequence (MSA) pairwise distances on gfx950.  Replaces MSADeviceArrays
// (src/MSA.cu:14-72), calculateParamsParallel (:103-156) and MSADistConstruction (:214-268), which
// compute ONE row per launch with one block per pair, by an all-pairs tiled kernel.
//
// Input layout (host ABI): 4-bit codes, 16 bases per uint64, as fourBitCompressor produces.
// Device layout: three bit planes per sequence, 32 bases per uint32 word:
//   V (code < 4), LO (code & 1), HI (code >> 1 & 1), positions >= L cleared.
// For a pair (a,b) per 32 bases:  useful += popc(Va | Vb)
//                                 match  += popc(Va & Vb & ~((LOa^LOb) | (HIa^HIb)))
// which equals the reference's  (a<4 || b<4)  and  (a<4 && a==b)  counts exactly (integers).
// The kernel is integer-VALU/LDS bound; HBM only sees the N^2 fp64 output.
#include "dpr_internal.hpp"

namespace dpr {

constexpr int kPT = 64;   // pairs tile edge (64 rows x 64 cols per block)
constexpr int kKC = 16;   // plane words (32 bases each) staged per step

__global__ __launch_bounds__(kThreads) void msa_planes_kernel(const uint64_t* __restrict__ packed4,
                                                              int64_t n, int64_t L, int64_t W64,
                                                              int64_t W32, uint32_t* __restrict__ planes)
{
    const int64_t total = n * W32;
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * kThreads) {
        const int64_t s = idx / W32, w = idx % W32;
        uint32_t V = 0, LO = 0, HI = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t w64 = 2 * w + h;
            const uint64_t word = w64 < W64 ? packed4[s * W64 + w64] : 0ull;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int64_t pos = w64 * 16 + j;
                const uint32_t c = (uint32_t)((word >> (4 * j)) & 15u);
                const uint32_t ok = (c < 4u && pos < L) ? 1u : 0u;
                const int bit = h * 16 + j;
                V |= ok << bit;
                LO |= (ok & c & 1u) << bit;
                HI |= (ok & (c >> 1) & 1u) << bit;
            }
        }
        planes[(0 * n + s) * W32 + w] = V;
        planes[(1 * n + s) * W32 + w] = LO;
        planes[(2 * n + s) * W32 + w] = HI;
    }
}

__device__ __forceinline__ double msa_epilogue(int useful, int match, int dist_type)
{
    // src/MSA.cu:233-235
    const double uncor = 1 - double(match) / useful;
    if (dist_type == DPR_DIST_UNCORRECTED) return uncor;
    return -0.75 * log(1.0 - uncor / 0.75);
}

// Block = 64 owned rows x 64 columns; thread (ty,tx) of 16x16 owns rows ty*4.., cols tx*4..
// LDS: [side][plane][k][64 sequences] so that four consecutive sequences are one 16-byte read.
__global__ __launch_bounds__(kThreads) void msa_dist_kernel(const uint32_t* __restrict__ planes,
                                                            int64_t n, int64_t W32, int dist_type,
                                                            double* __restrict__ D, int64_t ld,
                                                            int64_t rows_local, int rank, int world,
                                                            int64_t row0)
{
    // rows padded to 68 words: the staging writes (consecutive lanes = consecutive k) then hit 8 banks
    // two ways instead of one bank sixteen ways; 68*4 B keeps the 16-byte reads aligned
    __shared__ __attribute__((aligned(16))) uint32_t sA[3][kKC][kPT + 4];
    __shared__ __attribute__((aligned(16))) uint32_t sB[3][kKC][kPT + 4];

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t l0 = (int64_t)blockIdx.y * kPT;   // local row block (== ownership block)
    const int64_t c0 = (int64_t)blockIdx.x * kPT;   // global column block
    // world > 0: rows are the owned rows of (rank, world); world == 0: plain tip ids row0 + l
    const int64_t g0 = world > 0 ? shard_global_row(l0, rank, world) : row0 + l0;
    // single GPU, whole matrix: tiles strictly above the diagonal are produced by their mirror tile
    const bool mirror = (world == 1);
    if (mirror && c0 > g0 + kPT - 1) return;

    int useful[4][4], match[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { useful[r][c] = 0; match[r][c] = 0; }

    for (int64_t k0 = 0; k0 < W32; k0 += kKC) {
        // stage: 2 sides x 3 planes x 64 seqs x 16 words = 6144 words, 24 per thread.
        // consecutive lanes read consecutive words of one sequence (64-byte runs).
        for (int e = tid; e < 3 * kPT * kKC; e += kThreads) {
            const int p = e / (kPT * kKC), rem = e % (kPT * kKC);
            const int sq = rem / kKC, kk = rem % kKC;
            const int64_t k = k0 + kk;
            const int64_t ga = g0 + sq, gb = c0 + sq;
            uint32_t va = 0, vb = 0;
            if (k < W32) {
                if (ga < n && l0 + sq < rows_local) va = planes[((int64_t)p * n + ga) * W32 + k];
                if (gb < n) vb = planes[((int64_t)p * n + gb) * W32 + k];
            }
            sA[p][kk][sq] = va;
            sB[p][kk][sq] = vb;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < kKC; ++kk) {
            const uint4 aV = *reinterpret_cast<const uint4*>(&sA[0][kk][ty * 4]);
            const uint4 aL = *reinterpret_cast<const uint4*>(&sA[1][kk][ty * 4]);
            const uint4 aH = *reinterpret_cast<const uint4*>(&sA[2][kk][ty * 4]);
            const uint4 bV = *reinterpret_cast<const uint4*>(&sB[0][kk][tx * 4]);
            const uint4 bL = *reinterpret_cast<const uint4*>(&sB[1][kk][tx * 4]);
            const uint4 bH = *reinterpret_cast<const uint4*>(&sB[2][kk][tx * 4]);
            const uint32_t av[4] = { aV.x, aV.y, aV.z, aV.w }, al[4] = { aL.x, aL.y, aL.z, aL.w },
                           ah[4] = { aH.x, aH.y, aH.z, aH.w };
            const uint32_t bv[4] = { bV.x, bV.y, bV.z, bV.w }, bl[4] = { bL.x, bL.y, bL.z, bL.w },
                           bh[4] = { bH.x, bH.y, bH.z, bH.w };
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    useful[r][c] += __popc(av[r] | bv[c]);
                    const uint32_t diff = (al[r] ^ bl[c]) | (ah[r] ^ bh[c]);
                    match[r][c] += __popc(av[r] & bv[c] & ~diff);
                }
        }
        __syncthreads();
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t li = l0 + ty * 4 + r;
        const int64_t gi = g0 + ty * 4 + r;
        if (li >= rows_local || gi >= n) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t gj = c0 + tx * 4 + c;
            if (gj >= n) continue;
            const double d = (gi == gj) ? 0.0 : msa_epilogue(useful[r][c], match[r][c], dist_type);
            D[li * ld + gj] = d;
            if (mirror && c0 + kPT - 1 < g0) D[gj * ld + gi] = d;   // counts are symmetric (src/MSA.cu:121-122)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Distance types 3-6 (Tajima-Nei, K2P, Tamura, Jin-Nei): formulas and counters of
// src/divide_and_conquer/msa.cu:107-217 (counts), :238-265 (epilogues); the copies in src/MSA.cu
// index with the wrong variable.  Only sites valid in BOTH sequences count.
// With the planes (code = 2*HI + LO; A,C,G,T = 0..3), per 32 bases of a (row r, column c) pair:
//   both = Vr & Vc                      tot   += popc(both)
//   eq   = both & ~((LOr^LOc)|(HIr^HIc))          (match)
//   transitions  p: mismatch & ~(LOr^LOc)   (same parity), transversions q: mismatch & (LOr^LOc)
//   C or G       : HI ^ LO               (Tamura's gc1 = row, gc2 = column, mismatching sites only)
//   Tajima-Nei pair classes {A,G},{A,T},{C,G},{C,T} and base counts over both sequences.
// Block = 32 x 32 pairs, thread (ty,tx) of 16 x 16 owns a 2 x 2 sub-tile.
// ------------------------------------------------------------------------------------------------
constexpr int kET = 32;

template <int TYPE>
struct ExtCounts {
    int tot = 0, eq = 0, p = 0, q = 0, gc1 = 0, gc2 = 0;
    int fA = 0, fC = 0, fG = 0;   // base counts over both sequences (T = 2*tot - others)
    int pr0 = 0, pr1 = 0, pr2 = 0, pr3 = 0;
};

template <int TYPE>
__device__ __forceinline__ void ext_accum(ExtCounts<TYPE>& k, uint32_t vr, uint32_t lr, uint32_t hr, uint32_t vc,
                                          uint32_t lc, uint32_t hc)
{
    const uint32_t both = vr & vc;
    const uint32_t dl = lr ^ lc, dh = hr ^ hc;
    const uint32_t eq = both & ~(dl | dh);
    const uint32_t mis = both & ~eq;
    k.tot += __popc(both);
    if (TYPE == DPR_DIST_TAJIMANEI) {
        k.eq += __popc(eq);
        k.fA += __popc(both & ~hr & ~lr) + __popc(both & ~hc & ~lc);
        k.fC += __popc(both & ~hr & lr) + __popc(both & ~hc & lc);
        k.fG += __popc(both & hr & ~lr) + __popc(both & hc & ~lc);
        k.pr0 += __popc(both & ~lr & ~lc & dh);          // {A,G}
        k.pr1 += __popc(both & dh & dl & ~(hr ^ lr));    // {A,T}
        k.pr2 += __popc(both & dh & dl & (hr ^ lr));     // {C,G}
        k.pr3 += __popc(both & lr & lc & dh);            // {C,T}
    } else {
        k.p += __popc(mis & ~dl);
        k.q += __popc(mis & dl);
        if (TYPE == DPR_DIST_TAMURA) {
            k.gc1 += __popc(mis & (hr ^ lr));
            k.gc2 += __popc(mis & (hc ^ lc));
        }
    }
}

template <int TYPE>
__device__ __forceinline__ double ext_epilogue(const ExtCounts<TYPE>& k)
{
    const int tot = k.tot;
    if (TYPE == DPR_DIST_TAJIMANEI) {
        const int frac[4] = { k.fA, k.fC, k.fG, 2 * tot - k.fA - k.fC - k.fG };
        double fr[4];
        for (int i = 0; i < 4; ++i) fr[i] = double(frac[i]) / tot / 2.0;
        double h = 0;
        h += 0.5 * k.pr0 * fr[0] * fr[2];
        h += 0.5 * k.pr1 * fr[0] * fr[3];
        h += 0.5 * k.pr2 * fr[1] * fr[2];
        h += 0.5 * k.pr3 * fr[1] * fr[3];
        const double D = double(tot - k.eq) / tot;
        const double b = 0.5 * (1.0 - fr[0] * fr[0] - fr[2] * fr[2] + D * D / h);
        return -b * log(1.0 - D / b);
    }
    const double pp = double(k.p) / tot, qq = double(k.q) / tot;
    if (TYPE == DPR_DIST_K2P) return -0.5 * log((1 - 2 * pp - qq) * sqrt(1 - 2 * qq));
    if (TYPE == DPR_DIST_JINNEI) return 0.5 * (1.0 / (1 - 2 * pp - qq) + 0.5 / (1 - qq * 2) - 1.5);
    const double c = double(k.gc1) / tot + double(k.gc2) / tot - 2 * double(k.gc1) * double(k.gc2) / tot / tot;
    return -c * log(1 - pp / c - qq) - 0.5 * (1 - c) * log(1 - 2 * qq);
}

template <int TYPE>
__global__ __launch_bounds__(kThreads) void msa_dist_ext_kernel(const uint32_t* __restrict__ planes, int64_t n,
                                                                int64_t W32, double* __restrict__ D, int64_t ld,
                                                                int64_t rows_local, int rank, int world, int64_t row0)
{
    __shared__ uint32_t sA[3][kKC][kET + 1];
    __shared__ uint32_t sB[3][kKC][kET + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t l0 = (int64_t)blockIdx.y * kET;
    const int64_t c0 = (int64_t)blockIdx.x * kET;
    // kET divides the ownership block, so a row tile never straddles two owners
    const int64_t g0 = world > 0 ? shard_global_row(l0, rank, world) : row0 + l0;
    const bool mirror = (world == 1);
    if (mirror && c0 > g0 + kET - 1) return;
    ExtCounts<TYPE> k[2][2];
    for (int64_t k0 = 0; k0 < W32; k0 += kKC) {
        for (int e = tid; e < 3 * kET * kKC; e += kThreads) {
            const int p = e / (kET * kKC), rem = e % (kET * kKC);
            const int sq = rem / kKC, kk = rem % kKC;
            const int64_t w = k0 + kk;
            const int64_t ga = g0 + sq, gb = c0 + sq;
            uint32_t va = 0, vb = 0;
            if (w < W32) {
                if (ga < n && l0 + sq < rows_local) va = planes[((int64_t)p * n + ga) * W32 + w];
                if (gb < n) vb = planes[((int64_t)p * n + gb) * W32 + w];
            }
            sA[p][kk][sq] = va;
            sB[p][kk][sq] = vb;
        }
        __syncthreads();
        for (int kk = 0; kk < kKC; ++kk)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    ext_accum<TYPE>(k[r][c], sA[0][kk][ty * 2 + r], sA[1][kk][ty * 2 + r], sA[2][kk][ty * 2 + r],
                                    sB[0][kk][tx * 2 + c], sB[1][kk][tx * 2 + c], sB[2][kk][tx * 2 + c]);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int64_t li = l0 + ty * 2 + r, gi = g0 + ty * 2 + r;
        if (li >= rows_local || gi >= n) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int64_t gj = c0 + tx * 2 + c;
            if (gj >= n) continue;
            const double d = (gi == gj) ? 0.0 : ext_epilogue<TYPE>(k[r][c]);
            D[li * ld + gj] = d;
            if (mirror && c0 + kET - 1 < g0) D[gj * ld + gi] = d;
        }
    }
}

static int launch_ext(int dist_type, dim3 grid, hipStream_t s, const uint32_t* planes, int64_t n, int64_t W32,
                      double* D, int64_t ld, int64_t rows, int rank, int world, int64_t row0)
{
    switch (dist_type) {
    case DPR_DIST_TAJIMANEI: hipLaunchKernelGGL(msa_dist_ext_kernel<DPR_DIST_TAJIMANEI>, grid, dim3(kThreads), 0, s, planes, n, W32, D, ld, rows, rank, world, row0); break;
    case DPR_DIST_K2P:       hipLaunchKernelGGL(msa_dist_ext_kernel<DPR_DIST_K2P>, grid, dim3(kThreads), 0, s, planes, n, W32, D, ld, rows, rank, world, row0); break;
    case DPR_DIST_TAMURA:    hipLaunchKernelGGL(msa_dist_ext_kernel<DPR_DIST_TAMURA>, grid, dim3(kThreads), 0, s, planes, n, W32, D, ld, rows, rank, world, row0); break;
    case DPR_DIST_JINNEI:    hipLaunchKernelGGL(msa_dist_ext_kernel<DPR_DIST_JINNEI>, grid, dim3(kThreads), 0, s, planes, n, W32, D, ld, rows, rank, world, row0); break;
    default: set_error("unknown distance type (valid: 1-6)"); return DPR_ERR_ARG;
    }
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// test hook: integer counts of one row against columns [0,row)
__global__ __launch_bounds__(kThreads) void msa_counts_row_kernel(const uint32_t* __restrict__ planes,
                                                                  int64_t n, int64_t W32, int64_t row,
                                                                  int32_t* __restrict__ useful,
                                                                  int32_t* __restrict__ match)
{
    const int64_t c = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (c >= row) return;
    int u = 0, m = 0;
    for (int64_t k = 0; k < W32; ++k) {
        const uint32_t av = planes[(0 * n + row) * W32 + k], al = planes[(1 * n + row) * W32 + k],
                       ah = planes[(2 * n + row) * W32 + k];
        const uint32_t bv = planes[(0 * n + c) * W32 + k], bl = planes[(1 * n + c) * W32 + k],
                       bh = planes[(2 * n + c) * W32 + k];
        u += __popc(av | bv);
        m += __popc(av & bv & ~((al ^ bl) | (ah ^ bh)));
    }
    useful[c] = u;
    match[c] = m;
}

int msa_upload(MsaBuffers& m, const uint64_t* packed4, int64_t n, int64_t L, hipStream_t s)
{
    msa_free(m);
    m.n = n; m.L = L; m.W32 = (L + 31) / 32;
    const int64_t W64 = (L + 15) / 16;
    uint64_t* d_in = nullptr;
    DPR_HIP(hipMalloc(&d_in, sizeof(uint64_t) * (size_t)(n * W64)));
    DPR_HIP(hipMemcpyAsync(d_in, packed4, sizeof(uint64_t) * (size_t)(n * W64), hipMemcpyHostToDevice, s));
    DPR_HIP(hipMalloc(&m.planes, sizeof(uint32_t) * (size_t)(3 * n * m.W32)));
    const int64_t total = n * m.W32;
    const unsigned grid = (unsigned)((total + kThreads - 1) / kThreads > 8192 ? 8192 : (total + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(msa_planes_kernel, dim3(grid ? grid : 1), dim3(kThreads), 0, s, d_in, n, L, W64,
                       m.W32, m.planes);
    DPR_HIP(hipGetLastError());
    DPR_HIP(hipStreamSynchronize(s));
    DPR_HIP(hipFree(d_in));
    return DPR_OK;
}

void msa_free(MsaBuffers& m)
{
    if (m.planes) (void)hipFree(m.planes);
    m = MsaBuffers();
}

int msa_dist_rows(const MsaBuffers& m, NjBuffers& b, int dist_type, hipStream_t s)
{
    if (b.rows_local == 0) return DPR_OK;
    if (dist_type != DPR_DIST_UNCORRECTED && dist_type != DPR_DIST_JC) {
        dim3 g((unsigned)((m.n + kET - 1) / kET), (unsigned)((b.rows_local + kET - 1) / kET));
        return launch_ext(dist_type, g, s, m.planes, m.n, m.W32, b.D, b.ld, b.rows_local, b.rank, b.world, 0);
    }
    dim3 grid((unsigned)((m.n + kPT - 1) / kPT), (unsigned)((b.rows_local + kPT - 1) / kPT));
    hipLaunchKernelGGL(msa_dist_kernel, grid, dim3(kThreads), 0, s, m.planes, m.n, m.W32, dist_type, b.D,
                       b.ld, b.rows_local, b.rank, b.world, (int64_t)0);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int msa_dist_block_rows(const MsaBuffers& m, int64_t r0, int64_t nr, int rank, int world, int64_t ncols,
                        int dist_type, double* out, int64_t ld, hipStream_t s)
{
    if (nr <= 0 || ncols <= 0) return DPR_OK;
    (void)rank;
    if (dist_type != DPR_DIST_UNCORRECTED && dist_type != DPR_DIST_JC) {
        dim3 g((unsigned)((ncols + kET - 1) / kET), (unsigned)((nr + kET - 1) / kET));
        return launch_ext(dist_type, g, s, m.planes, m.n, m.W32, out, ld, nr, 0, world, r0);
    }
    dim3 grid((unsigned)((ncols + kPT - 1) / kPT), (unsigned)((nr + kPT - 1) / kPT));
    hipLaunchKernelGGL(msa_dist_kernel, grid, dim3(kThreads), 0, s, m.planes, m.n, m.W32, dist_type, out, ld, nr,
                       0, world, r0);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int msa_counts_row(const MsaBuffers& m, int64_t row, int32_t* d_useful, int32_t* d_match, hipStream_t s)
{
    if (row <= 0) return DPR_OK;
    const unsigned grid = (unsigned)((row + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(msa_counts_row_kernel, dim3(grid), dim3(kThreads), 0, s, m.planes, m.n, m.W32, row,
                       d_useful, d_match);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

}  // namespace dpr
