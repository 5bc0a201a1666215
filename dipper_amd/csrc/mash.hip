// Mash sketches and Mash distances on gfx950.  Replaces sketchConstruction (src/mash.cu:260-369,
// MurmurHash3_x64_128_MASH :159-236, decompress/memcmp_device :239-258), rearrangeHashList
// (:371-384) and mashDistConstruction (:426-455) of the reference.
//
// Sketch: one workgroup per sequence.  All k-mer hashes of a chunk go to LDS next to the current
// bottom-S list and a bitonic sort keeps the S smallest (duplicates kept); the
// reference instead radix-sorts 512 new hashes against the kept 1000 per round.
// Canonical k-mer without materialising strings: with the window w (base i at bits 2i) the
// big-endian value of the forward string is rev2(w) and that of the reverse complement is
// ~w & mask, so "forward <= reverse (ASCII, A<C<G<T)" is rev2(w) <= (~w & mask).
//
// Layout: sketches row-major [n][S] u64 in HBM (the reference transposes to [S][n] for its
// one-row-per-launch kernel).
#include "dpr_internal.hpp"

#include <cstdlib>

namespace dpr {

constexpr int kSketchThreads = 1024;
constexpr int kSortCap = 16384;  // u64 slots of LDS used by the sketch kernel (128 KiB)

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

// reverse the order of the k 2-bit groups of w (k <= 32)
__device__ __forceinline__ uint64_t rev2(uint64_t w, int k)
{
    w = ((w >> 2) & 0x3333333333333333ull) | ((w & 0x3333333333333333ull) << 2);
    w = ((w >> 4) & 0x0f0f0f0f0f0f0f0full) | ((w & 0x0f0f0f0f0f0f0f0full) << 4);
    w = __builtin_bswap64(w);
    return w >> (64 - 2 * k);
}

// MurmurHash3_x64_128(seed 42).h1 of the canonical k-mer whose first base sits at bit 0 of cw
// (2 bits per base), k in [1,15]: only the tail path of the hash runs (len < 16).
__device__ __forceinline__ uint64_t murmur_kmer(uint64_t cw, int k)
{
    const uint32_t lut = 0x54474341u;  // 'A','C','G','T'
    uint64_t k1 = 0, k2 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < k) k1 |= (uint64_t)((lut >> (8 * ((cw >> (2 * i)) & 3))) & 0xFF) << (8 * i);
#pragma unroll
    for (int i = 8; i < 15; ++i)
        if (i < k) k2 |= (uint64_t)((lut >> (8 * ((cw >> (2 * i)) & 3))) & 0xFF) << (8 * (i - 8));
    const uint64_t c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    uint64_t h1 = 42, h2 = 42;
    if (k > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 ^= (uint64_t)k; h2 ^= (uint64_t)k;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

__device__ __forceinline__ uint64_t kmer_hash_at(const uint64_t* __restrict__ seq, uint64_t nwords,
                                                 uint64_t p, int k)
{
    const uint64_t idx = p >> 5;
    const int sh = (int)(2 * (p & 31));
    uint64_t w = seq[idx] >> sh;
    if (sh > 0 && idx + 1 < nwords) w |= seq[idx + 1] << (64 - sh);
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    w &= mask;
    const uint64_t fwd_be = rev2(w, k), rc_be = ~w & mask;
    const uint64_t cw = (fwd_be <= rc_be) ? w : rev2(rc_be, k);
    return murmur_kmer(cw, k);
}

// test hook: hashes of every k-mer position of one sequence
__global__ void mash_hash_positions_kernel(const uint64_t* __restrict__ packed2, uint64_t nwords,
                                           uint64_t len, int k, uint64_t* __restrict__ out)
{
    if (len < (uint64_t)k) return;
    const uint64_t nk = len - (uint64_t)k + 1;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nk; p += (uint64_t)gridDim.x * blockDim.x)
        out[p] = kmer_hash_at(packed2, nwords, p, k);
}

// bitonic sort of m (power of two) u64 keys in LDS, ascending
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* s, int m)
{
    for (int size = 2; size <= m; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (m >> 1); t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool asc = ((lo & size) == 0);
                const uint64_t a = s[lo], b = s[hi];
                if ((a > b) == asc) { s[lo] = b; s[hi] = a; }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kSketchThreads) void mash_sketch_kernel(const uint64_t* __restrict__ packed2,
                                                                     const uint64_t* __restrict__ word_off,
                                                                     const uint64_t* __restrict__ lens,
                                                                     int64_t n, int k, int S,
                                                                     uint64_t* __restrict__ sketches)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem);
    const int cap = kSortCap - S;  // hashes per chunk
    for (int64_t q = blockIdx.x; q < n; q += gridDim.x) {
        const uint64_t len = lens[q];
        const uint64_t* seq = packed2 + word_off[q];
        const uint64_t nwords = (len + 31) / 32;
        const uint64_t nk = len >= (uint64_t)k ? len - (uint64_t)k + 1 : 0;
        for (int i = threadIdx.x; i < S; i += blockDim.x) s[i] = ~0ull;
        for (uint64_t base = 0; base < nk || base == 0; base += (uint64_t)cap) {
            const uint64_t cnt = nk > base ? (nk - base < (uint64_t)cap ? nk - base : (uint64_t)cap) : 0;
            int m = 1024;
            while (m < S + (int)cnt) m <<= 1;
            for (int i = threadIdx.x; i < m - S; i += blockDim.x)
                s[S + i] = (uint64_t)i < cnt ? kmer_hash_at(seq, nwords, base + (uint64_t)i, k) : ~0ull;
            bitonic_sort_lds(s, m);
            if (nk == 0) break;
        }
        for (int i = threadIdx.x; i < S; i += blockDim.x) sketches[q * S + i] = s[i];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Mash distance of one pair: outer list X (the LOWER index tip, "A"), inner list Y (the higher
// index tip, "B"); exact restatement of the loop of src/mash.cu:437-454 flattened to one step per
// consumed element.
// ------------------------------------------------------------------------------------------------
template <typename PX, typename PY>
__device__ __forceinline__ double mash_pair(PX X, PY Y, int S, int k)
{
    int uni = 0, inter = 0, ai = 0, bp = 0;
    uint64_t a = X[0], b = Y[0];
    while (true) {
        if (bp < S && b <= a) {
            if (b < a) uni++; else inter++;
            bp++;
            if (bp < S) b = Y[bp];
            if (uni >= S) break;
        } else {
            uni++; ai++;
            if (uni >= S) break;
            a = X[ai];
        }
    }
    const double j = fmax((double)inter, 1.0) / uni;
    return fmin(1.0, fabs(log(2.0 * j / (1.0 + j)) / (double)k));
}

// rows [r0, r0+nr) (global tip ids given by row_ids or r0+t), columns [0, ncols): out[t*ld + j].
// One block = 256 columns x 1 row; the row's sketch is staged in LDS.
// FULL: also j > i (roles swapped) and the diagonal (0); otherwise only j < i is written.
template <bool FULL>
__global__ __launch_bounds__(kThreads) void mash_dist_rows_kernel(const uint64_t* __restrict__ sk, int S, int k,
                                                                  int64_t n, int64_t r0, int64_t nr,
                                                                  int rank, int world, int64_t ncols,
                                                                  double* __restrict__ out, int64_t ld)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* srow = reinterpret_cast<uint64_t*>(smem);
    const int64_t t = blockIdx.y;
    if (t >= nr) return;
    // world > 0: rows are owned rows (local index r0+t -> global); world == 0: plain tip ids
    const int64_t i = world > 0 ? shard_global_row(r0 + t, rank, world) : r0 + t;
    if (i >= n) return;
    const int64_t lim = FULL ? ncols : (i < ncols ? i : ncols);
    const int64_t j0 = (int64_t)blockIdx.x * kThreads;
    if (j0 >= lim) return;
    for (int e = threadIdx.x; e < S; e += kThreads) srow[e] = sk[i * S + e];
    __syncthreads();
    const int64_t j = j0 + threadIdx.x;
    if (j >= lim) return;
    double d;
    if (j == i) d = 0.0;
    else if (j < i) d = mash_pair(sk + j * S, srow, S, k);   // A = column (lower index), B = row
    else d = mash_pair(srow, sk + j * S, S, k);              // A = row (lower index),    B = column
    out[t * ld + j] = d;
}

// ------------------------------------------------------------------------------------------------
// Lookup formulation of the same distance (S <= 1024, pairs with column index < row index).
// The sequential merge is equivalent to: events in value order, B before A on ties; every A element
// and every B element whose value is not in A advances `uni`; B elements equal to an A value advance
// `inter`; stop at the S-th advance.  Hence, iterating over the DISTINCT values v of A (outer list =
// the lower-index tip) that also occur in B (inner list):
//     inter = sum of mult_B(v) over those v with   first_A(v) + #{b < v} - #{matching b < v}  <  S
// and uni = S.  So B is never walked: per row tip the block keeps its sketch in LDS with a bucket
// index (bucket[k] = first t with b_t >> shift >= k, 1024 buckets), a wave holds one column sketch in
// registers (16 values per lane) and resolves each value with one bucket read and ~1 sketch read;
// a wave prefix sum supplies the matches of the lanes before it.  A block keeps 12 row tips
// resident (146 KiB of LDS) and streams the columns, 8 at a time (one per wave).
// ------------------------------------------------------------------------------------------------
constexpr int kLS = 1024;            // largest sketch this kernel handles
constexpr int kLRows = 12;           // row tips resident per block
constexpr int kLThreads = 512;       // 8 waves
constexpr int kLColsPerBlock = 512;
constexpr int kLBuckets = 1024;

// JOBS = false: rows r0 + [0, nr) (tip ids) x columns [0, ncols), pairs with column id < row id;
//   out[t * ld + j], transposed: out[j * ld + t] (t = row - r0); mirror: also out[j * ld + row].
// JOBS = true: divide-and-conquer cluster blocks (PairJobs, dpr_internal.hpp): job = (cluster, first
//   member t0, first leaf-list position u0); pair (t, u) with u < kDcLeaves + t.
template <bool JOBS>
__global__ __launch_bounds__(kLThreads) void mash_dist_lookup_kernel(const uint64_t* __restrict__ sk, int S, int k,
                                                                     int64_t n, int64_t r0, int64_t nr,
                                                                     int64_t ncols, double* __restrict__ out,
                                                                     int64_t ld, int mirror, int transposed,
                                                                     PairJobs J)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* vals = reinterpret_cast<uint64_t*>(smem);                              // [kLRows][kLS]
    uint32_t* bucket = reinterpret_cast<uint32_t*>(vals + kLRows * kLS);             // [kLRows][kLBuckets + 1]
    __shared__ int s_shift[kLRows];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int64_t t0, cbeg, cend;
    int64_t irow[kLRows];     // tip id of the resident rows (-1: none)
    int64_t rlim[kLRows];     // row r takes column positions < rlim[r]
    int64_t imax = -1;        // largest rlim
    const int32_t* colids = nullptr;
    if (JOBS) {
        const int4 job = J.jobs[blockIdx.x];
        const int ci = job.x, m = J.cl_m[ci];
        t0 = job.y; cbeg = job.z;
        colids = J.cols + J.cl_coff[ci];
        out = J.out + J.cl_out[ci];
        ld = J.cl_ld[ci];
#pragma unroll
        for (int r = 0; r < kLRows; ++r) {
            const int64_t tt = t0 + r;
            irow[r] = tt < m ? J.members[J.cl_moff[ci] + tt] : -1;
            rlim[r] = tt < m ? kDcLeaves + tt : -1;
            imax = max(imax, rlim[r]);
        }
        cend = min(imax, cbeg + kLColsPerBlock);
    } else {
        t0 = (int64_t)blockIdx.y * kLRows;
        cbeg = (int64_t)blockIdx.x * kLColsPerBlock;
        cend = min(ncols, cbeg + kLColsPerBlock);
#pragma unroll
        for (int r = 0; r < kLRows; ++r) {
            const int64_t tt = t0 + r;
            int64_t i = -1;
            if (tt < nr) { i = r0 + tt; if (i >= n) i = -1; }
            irow[r] = i; rlim[r] = i;
            imax = max(imax, i);
        }
    }
    if (imax < 0 || cbeg >= imax) return;   // nothing to do in this column range

    // ---- resident row structures
    for (int e = tid; e < kLRows * kLS; e += kLThreads) {
        const int r = e / kLS, sidx = e % kLS;
        vals[e] = (irow[r] >= 0 && sidx < S) ? sk[irow[r] * S + sidx] : ~0ull;
    }
    __syncthreads();
    if (tid < kLRows) {
        const uint64_t mx = vals[tid * kLS + (S - 1)];
        const int bits = mx ? 64 - __clzll((long long)mx) : 1;
        s_shift[tid] = bits > 10 ? bits - 10 : 0;
    }
    __syncthreads();
    for (int e = tid; e < kLRows * (kLBuckets + 1); e += kLThreads) {
        const int r = e / (kLBuckets + 1), kb = e % (kLBuckets + 1);
        const uint64_t* v = vals + r * kLS;
        const int sh = s_shift[r];
        int lo = 0, hi = S;                      // first t with (v[t] >> sh) >= kb
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((v[mid] >> sh) < (uint64_t)kb) lo = mid + 1; else hi = mid;
        }
        bucket[e] = (uint32_t)lo;
    }
    __syncthreads();

    // ---- stream the columns: wave w takes column position c0 + w
    for (int64_t c0 = cbeg; c0 < cend && c0 < imax; c0 += kLThreads / 64) {
        const int64_t jpos = c0 + w;
        if (jpos >= cend || jpos >= imax) continue;
        const int64_t j = JOBS ? (int64_t)colids[jpos] : jpos;
        if (j < 0) continue;                                     // empty leaf-list entry (wave-uniform)
        uint64_t a[16];
        const uint64_t* col = sk + j * S + 16 * lane;
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = (16 * lane + u < S) ? col[u] : ~0ull;
        uint64_t aprev = __shfl_up((unsigned long long)a[15], 1, 64);
        const bool has_prev = lane > 0;
#pragma unroll 1
        for (int r = 0; r < kLRows; ++r) {
            if (irow[r] < 0 || jpos >= rlim[r]) continue;        // wave-uniform
            const uint64_t* v = vals + r * kLS;
            const uint32_t* bk = bucket + r * (kLBuckets + 1);
            const int sh = s_shift[r];
            // phase 1-2: bucket and first probe of all 16 values (independent LDS reads overlap)
            int t[16];
            uint64_t bv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                uint64_t kb = a[u] >> sh;
                if (kb > (uint64_t)kLBuckets) kb = kLBuckets;
                t[u] = (int)bk[kb];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) bv[u] = v[t[u] < S ? t[u] : S - 1];
            // phase 3: advance to #{b < a} (rarely more than one or two rounds)
            for (;;) {
                bool any = false;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const bool adv = t[u] < S && bv[u] < a[u];
                    t[u] += adv ? 1 : 0;
                    any |= adv;
                }
                if (!__any(any)) break;
#pragma unroll
                for (int u = 0; u < 16; ++u) bv[u] = v[t[u] < S ? t[u] : S - 1];
            }
            // phase 4: matches, multiplicities (the next slot is read for every value: no branch)
            uint64_t nx[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) nx[u] = v[t[u] + 1 < S ? t[u] + 1 : S - 1];
            int cu[16], mu[16];
            int msum = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                cu[u] = 1 << 30; mu[u] = 0;
                const int sidx = 16 * lane + u;
                const uint64_t av = a[u];
                const bool first = (u == 0) ? (!has_prev || aprev != av) : (a[u > 0 ? u - 1 : 0] != av);
                const bool match = sidx < S && first && t[u] < S && bv[u] == av;
                if (match) {
                    int mult = 1;
                    if (t[u] + 1 < S && nx[u] == av) {           // duplicates in B: rare, walk them
                        mult = 2;
                        while (t[u] + mult < S && v[t[u] + mult] == av) ++mult;
                    }
                    cu[u] = sidx + t[u] - msum;                  // rank before the lanes' prefix is subtracted
                    mu[u] = mult;
                    msum += mult;
                }
            }
            int incl = msum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int x = __shfl_up(incl, off, 64);
                if (lane >= off) incl += x;
            }
            const int thr = S + (incl - msum);                   // rank - prefix < S
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) cnt += (cu[u] < thr) ? mu[u] : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
            if (lane == 0) {
                const double jac = fmax((double)cnt, 1.0) / S;
                const double d = fmin(1.0, fabs(log(2.0 * jac / (1.0 + jac)) / (double)k));
                if (JOBS) out[(t0 + r) * ld + jpos] = d;
                else {
                    if (transposed) out[j * ld + (t0 + r)] = d; else out[(t0 + r) * ld + j] = d;
                    if (mirror) out[j * ld + (r0 + t0 + r)] = d;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int mash_upload(MashBuffers& m, const uint64_t* packed2, const uint64_t* word_off, const uint64_t* len,
                int64_t n, hipStream_t s)
{
    mash_free(m);
    m.n = n;
    uint64_t total = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t end = word_off[i] + (len[i] + 31) / 32;
        if (end > total) total = end;
    }
    m.total_words = total;
    DPR_HIP(hipMalloc(&m.packed2, sizeof(uint64_t) * (size_t)(total + 2)));
    DPR_HIP(hipMemsetAsync(m.packed2, 0, sizeof(uint64_t) * (size_t)(total + 2), s));
    DPR_HIP(hipMalloc(&m.word_off, sizeof(uint64_t) * (size_t)n));
    DPR_HIP(hipMalloc(&m.len, sizeof(uint64_t) * (size_t)n));
    DPR_HIP(hipMemcpyAsync(m.packed2, packed2, sizeof(uint64_t) * (size_t)total, hipMemcpyHostToDevice, s));
    DPR_HIP(hipMemcpyAsync(m.word_off, word_off, sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, s));
    DPR_HIP(hipMemcpyAsync(m.len, len, sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, s));
    DPR_HIP(hipStreamSynchronize(s));
    return DPR_OK;
}

void mash_free(MashBuffers& m)
{
    void* ptrs[] = { m.packed2, m.word_off, m.len, m.sketches };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    m = MashBuffers();
}

int mash_sketch(MashBuffers& m, int k, int S, hipStream_t s)
{
    if (k < 2 || k > 15) { set_error("kmer size must be in [2,15]"); return DPR_ERR_ARG; }
    if (S < 1 || S > 4096) { set_error("sketch size must be in [1,4096]"); return DPR_ERR_ARG; }
    if (m.sketches) { (void)hipFree(m.sketches); m.sketches = nullptr; }
    DPR_HIP(hipMalloc(&m.sketches, sizeof(uint64_t) * (size_t)(m.n * S)));
    m.S = S; m.k = k;
    static bool attr_set = false;
    if (!attr_set) {
        DPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mash_sketch_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kSortCap * (int)sizeof(uint64_t)));
        attr_set = true;
    }
    const unsigned grid = (unsigned)(m.n < 1024 ? m.n : 1024);
    hipLaunchKernelGGL(mash_sketch_kernel, dim3(grid), dim3(kSketchThreads), kSortCap * sizeof(uint64_t), s,
                       m.packed2, m.word_off, m.len, m.n, k, S, m.sketches);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

static int lookup_attr()
{
    static bool attr_set = false;
    const size_t tlds = sizeof(uint64_t) * kLRows * kLS + sizeof(uint32_t) * kLRows * (kLBuckets + 1);
    if (!attr_set) {
        DPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mash_dist_lookup_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds));
        DPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mash_dist_lookup_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds));
        attr_set = true;
    }
    return DPR_OK;
}

int mash_jobs_rows() { return kLRows; }
int mash_jobs_cols() { return kLColsPerBlock; }

int mash_dist_jobs(const MashBuffers& m, const PairJobs& J, int njobs, hipStream_t s)
{
    if (njobs <= 0) return DPR_OK;
    if (m.S > kLS) { set_error("divide-and-conquer mode needs a sketch size <= 1024"); return DPR_ERR_ARG; }
    if (int rc = lookup_attr()) return rc;
    const size_t tlds = sizeof(uint64_t) * kLRows * kLS + sizeof(uint32_t) * kLRows * (kLBuckets + 1);
    hipLaunchKernelGGL(mash_dist_lookup_kernel<true>, dim3((unsigned)njobs), dim3(kLThreads), tlds, s, m.sketches, m.S,
                       m.k, m.n, (int64_t)0, (int64_t)0, (int64_t)0, (double*)nullptr, (int64_t)0, 0, 0, J);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int mash_dist_rows(const MashBuffers& m, int64_t r0, int64_t nr, int rank, int world, bool full,
                   int64_t ncols, double* out, int64_t ld, hipStream_t s, bool transposed)
{
    if (nr <= 0 || ncols <= 0) return DPR_OK;
    if (transposed && (full || world > 1 || m.S > kLS)) { set_error("mash_dist_rows: transposed output needs the lookup kernel"); return DPR_ERR_ARG; }
    // lower-triangle pairs (placement batches; single-GPU NJ with a mirror write): lookup kernel
    const bool mirror = full && world == 1 && r0 == 0;
    if (m.S <= kLS && (!full || mirror) && world <= 1 && (transposed || !std::getenv("DPR_MASH_SIMPLE"))) {
        if (int rc = lookup_attr()) return rc;
        const size_t tlds = sizeof(uint64_t) * kLRows * kLS + sizeof(uint32_t) * kLRows * (kLBuckets + 1);
        dim3 tgrid((unsigned)((ncols + kLColsPerBlock - 1) / kLColsPerBlock), (unsigned)((nr + kLRows - 1) / kLRows));
        hipLaunchKernelGGL(mash_dist_lookup_kernel<false>, tgrid, dim3(kLThreads), tlds, s, m.sketches, m.S, m.k, m.n, r0,
                           nr, ncols, out, ld, mirror ? 1 : 0, transposed ? 1 : 0, PairJobs());
        DPR_HIP(hipGetLastError());
        return DPR_OK;
    }
    dim3 grid((unsigned)((ncols + kThreads - 1) / kThreads), (unsigned)nr);
    const size_t lds = sizeof(uint64_t) * (size_t)m.S;
    if (full)
        hipLaunchKernelGGL(mash_dist_rows_kernel<true>, grid, dim3(kThreads), lds, s, m.sketches, m.S, m.k, m.n,
                           r0, nr, rank, world, ncols, out, ld);
    else
        hipLaunchKernelGGL(mash_dist_rows_kernel<false>, grid, dim3(kThreads), lds, s, m.sketches, m.S, m.k, m.n,
                           r0, nr, rank, world, ncols, out, ld);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int mash_hash_positions(const MashBuffers& m, int64_t seq, int k, uint64_t* d_out, uint64_t len, uint64_t word_off,
                        hipStream_t s)
{
    if (len < (uint64_t)k) return DPR_OK;
    hipLaunchKernelGGL(mash_hash_positions_kernel, dim3(64), dim3(256), 0, s, m.packed2 + word_off, (len + 31) / 32,
                       len, k, d_out);
    (void)seq;
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

}  // namespace dpr
