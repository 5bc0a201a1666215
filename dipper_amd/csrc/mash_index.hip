// Mash distances through an inverted index (round 2).  Replaces mashDistConstruction (src/mash.cu:426-455) for the
// row-against-many-columns shapes (NJ matrix, placement batches, divide-and-conquer query x backbone blocks): the default
// pair kernel wherever the index can be built (mash.hip: kernel choice).
//
// The reference's merge of column sketch A (outer list) and row sketch B (inner list) counts, for the DISTINCT values
// v of A that also occur in B,    inter += mult_B(v)   as long as   first_A(v) + #{b < v} - #{matched b < v} < S
// (and uni = S at the end; mash.hip).  So a pair needs nothing but its SHARED values, in
// ascending order, with three small integers each: the position of v in A, the position of v in B, the multiplicity
// of v in B -- and one counter.  That is a join, not a merge:
//   * index (once per sketch set): the tips are cut into chunks of kIC = 1 024; a chunk's (value, tip, position) triples --
//     first occurrences only -- are sorted by value (round 4: every tip's sketch is already ascending, so a chunk is 512
//     sorted runs and the sort is nine rounds of pairwise merge-path merges, mi_merge_kernel; until round 3 a rocPRIM
//     segmented radix sort -- the one piece of vendor device code near the path, and 3.7 of the library's 5 MB);
//     per chunk the distinct values, the start of each value's posting list, and a 65 536-bucket directory on the
//     leading bits;
//   * query: ONE WAVEFRONT per (row, chunk).  It walks the row's distinct values in ascending order (64 directory
//     look-ups at a time, one per lane), and for every value found it streams the posting list: lane = posting =
//     one column tip; counter c[tip] (16 bits, LDS, 1 KiB per wavefront) is read, the reference's condition
//     pos_A + pos_B - c < S tested, c += mult_B stored.  Tips of one posting list are distinct, lists are applied in
//     value order, so every counter sees its shared values in the reference's order.  At the end c[tip] IS inter.
//     A value held by most tips of the chunk (every value of a clonal data set) also has a block of 512 positions indexed
//     by tip: lanes then read positions and counters in tip order (no address arithmetic, conflict-free LDS).
// Work per pair = its shared values (550 of 1 000 at 4 % divergence, ~20 for unrelated reads) instead of the ~1 500
// wave instructions of the table kernel's rank look-ups (68 VALU + 32 LDS wave instructions per pair on clonal reads by
// rocprofv3), and the chunk's postings (2 MB) stay in L2 while all wavefronts of the moment work on the same chunk (tasks
// are chunk-major): 4.4-10 G pairs/s at 20 000 reads x 3 kb, against 0.4-0.7 G of the table kernel.
#include "dpr_internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace dpr {

constexpr int kIC = 1024;            // tips per chunk.  Round 5, with the packed dense path: 1024 beats 512 at every divergence (distance rows of a
                                     // 50 000-tip placement run, mean branch 2e-5 / 1e-3 / 1e-2 / 0.1 / 1: 201 / 199 / 123 / 99 / 25 -> 183 / 173 / 91 / 73 / 22 ms;
                                     // --add 50 000 onto 500 000: 2 410 -> 2 164 ms); 2 048 is better from scratch (159 / 158 / 77 / 66 / 26) but
                                     // worse for --add (2 337 ms: sketches of 1 000-base reads); beside the tree kernels of a run from scratch 512
                                     // is 5 % ahead of 1024 (100 000 tips 1.71 / 1.80 s), which --add's 2 410 -> 2 164 ms outweighs;
                                     // profiles/mash_rows_sweep.sh, mash_kic_eval.sh
// a posting: byte offset of the tip's 16-bit counter (2 * (tip mod kIC)) in the high half, sketch position in the low half
constexpr int kIBktLog = 16;
constexpr int kINB = 1 << kIBktLog;  // directory buckets per chunk
constexpr int kIThreads = 256;       // (index build kernels)
// not a first copy: counter of tip 511, position 65535 -- the reference's condition never holds for it (S <= 4096), so the
// kernel needs no special case
constexpr uint32_t kINone = ((uint32_t)(2 * (kIC - 1)) << 16) | 0xFFFFu;

// payload of every sketch entry: 2 * (tip mod 512) << 16 | position for the first occurrence of a value in its sketch,
// none for further copies; mult = copies of the value at first occurrences, 0 elsewhere
__global__ __launch_bounds__(kIThreads) void mi_payload_kernel(const uint64_t* __restrict__ sk, int S, int64_t total,
                                                               uint32_t* __restrict__ pay, uint16_t* __restrict__ mult)
{
    const int64_t idx = (int64_t)blockIdx.x * kIThreads + threadIdx.x;
    if (idx >= total) return;
    const int64_t t = idx / S;
    const int p = (int)(idx - t * S);
    const uint64_t v = sk[idx];
    const bool first = p == 0 || sk[idx - 1] != v;
    int m = 0;
    if (first) {
        m = 1;
        while (p + m < S && sk[idx + m] == v) ++m;
    }
    mult[idx] = (uint16_t)m;
    pay[idx] = first ? (((uint32_t)(2 * (t & (kIC - 1))) << 16) | (uint32_t)p) : kINone;
}


// ------------------------------------------------------------------------------------------------
// Index build without vendor device code (round 4).
// (1) Sort of a chunk's entries by value = MERGE of its 512 runs: a tip's sketch is ascending already.  Round r merges
//     neighbouring runs of R = S 2^r entries; nine rounds.  One workgroup produces 1 024 consecutive outputs of one pair of
//     runs: two merge-path searches on the pair's diagonals find the inputs it needs (<= 1 024 entries, staged in LDS), every
//     thread then finds its own four outputs by the same search in LDS and merges them serially.  Ties take the LEFT run
//     first and runs keep their order, i.e. the result is the stable sort of the tip-major input -- what the radix sort gave.
// (2) Inclusive scan of 32-bit flags: block sums, one block over the block sums, block-local scan + offset.
// ------------------------------------------------------------------------------------------------
constexpr int kMT = 1024;            // outputs per workgroup of the merge kernel (4 per thread)

__global__ __launch_bounds__(kIThreads) void mi_merge_kernel(const uint64_t* __restrict__ srcK, const uint32_t* __restrict__ srcP,
                                                             uint64_t* __restrict__ dstK, uint32_t* __restrict__ dstP, int64_t total,
                                                             int64_t seg, int64_t R, int64_t pairs_per_chunk, int64_t tiles_per_pair)
{
    __shared__ uint64_t sk[kMT];
    __shared__ uint32_t sp[kMT];
    __shared__ int64_t s_part[4];
    const int tid = threadIdx.x;
    const int64_t blk = blockIdx.x;
    const int64_t tile = blk % tiles_per_pair;
    const int64_t pj = (blk / tiles_per_pair) % pairs_per_chunk;
    const int64_t c = blk / (tiles_per_pair * pairs_per_chunk);
    const int64_t cbeg = c * seg, cend = cbeg + seg < total ? cbeg + seg : total;
    const int64_t abeg = cbeg + 2 * pj * R;
    if (abeg >= cend) return;
    const int64_t aend = abeg + R < cend ? abeg + R : cend;
    const int64_t bbeg = aend, bend = bbeg + R < cend ? bbeg + R : cend;
    const int64_t la = aend - abeg, lb = bend - bbeg;
    const int64_t d0 = tile * kMT;
    if (d0 >= la + lb) return;
    const int64_t d1 = d0 + kMT < la + lb ? d0 + kMT : la + lb;
    const uint64_t* __restrict__ A = srcK + abeg;
    const uint64_t* __restrict__ B = srcK + bbeg;
    if (tid < 2) {      // merge path: the number of A entries among the first d outputs (ties: A first)
        const int64_t d = tid ? d1 : d0;
        int64_t lo = d > lb ? d - lb : 0, hi = d < la ? d : la;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (A[mid] <= B[d - 1 - mid]) lo = mid + 1; else hi = mid;
        }
        s_part[tid] = lo;
    }
    __syncthreads();
    const int64_t a0 = s_part[0], a1 = s_part[1];
    const int64_t b0 = d0 - a0, b1 = d1 - a1;
    const int na = (int)(a1 - a0), nb = (int)(b1 - b0);
    for (int k = tid; k < na; k += kIThreads) { sk[k] = A[a0 + k]; sp[k] = srcP[abeg + a0 + k]; }
    for (int k = tid; k < nb; k += kIThreads) { sk[na + k] = B[b0 + k]; sp[na + k] = srcP[bbeg + b0 + k]; }
    __syncthreads();
    const int nout = na + nb;
    const int o0 = 4 * tid;
    if (o0 >= nout) return;
    int lo = o0 > nb ? o0 - nb : 0, hi = o0 < na ? o0 : na;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sk[mid] <= sk[na + o0 - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    int ai = lo, bi = o0 - lo;
    const int64_t obase = abeg + d0 + o0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (o0 + k >= nout) break;
        const bool takeA = ai < na && (bi >= nb || sk[ai] <= sk[na + bi]);
        const int src = takeA ? ai : na + bi;
        dstK[obase + k] = sk[src];
        dstP[obase + k] = sp[src];
        ai += takeA ? 1 : 0;
        bi += takeA ? 0 : 1;
    }
}

constexpr int kScanItems = 16;       // per thread: a workgroup scans 4 096 flags
__device__ __forceinline__ uint32_t mi_block_scan_incl(uint32_t v, uint32_t* swarp)      // inclusive scan over the 256 threads
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off, 64);
        if (lane >= off) v += u;
    }
    if (lane == 63) swarp[w] = v;
    __syncthreads();
    uint32_t add = 0;
    for (int k = 0; k < w; ++k) add += swarp[k];
    __syncthreads();
    return v + add;
}
__global__ __launch_bounds__(kIThreads) void mi_scan_sums_kernel(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ bsum)
{
    __shared__ uint32_t swarp[kIThreads / 64];
    const int64_t base = ((int64_t)blockIdx.x * kIThreads + threadIdx.x) * kScanItems;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) acc += base + k < n ? in[base + k] : 0u;
    const uint32_t incl = mi_block_scan_incl(acc, swarp);
    if (threadIdx.x == kIThreads - 1) bsum[blockIdx.x] = incl;
}
// exclusive scan of the block sums in place, one workgroup
__global__ __launch_bounds__(kIThreads) void mi_scan_top_kernel(uint32_t* __restrict__ bsum, int64_t nb)
{
    __shared__ uint32_t swarp[kIThreads / 64];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += kIThreads) {
        const int64_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? bsum[i] : 0u;
        const uint32_t incl = mi_block_scan_incl(v, swarp);
        const uint32_t carry = s_carry;
        if (i < nb) bsum[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == kIThreads - 1) s_carry = carry + incl;
        __syncthreads();
    }
}
__global__ __launch_bounds__(kIThreads) void mi_scan_apply_kernel(const uint32_t* __restrict__ in, int64_t n, const uint32_t* __restrict__ bsum,
                                                                  uint32_t* __restrict__ out)
{
    __shared__ uint32_t swarp[kIThreads / 64];
    const int64_t base = ((int64_t)blockIdx.x * kIThreads + threadIdx.x) * kScanItems;
    uint32_t v[kScanItems];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { v[k] = base + k < n ? in[base + k] : 0u; acc += v[k]; }
    uint32_t run = mi_block_scan_incl(acc, swarp) - acc + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        run += v[k];
        if (base + k < n) out[base + k] = run;
    }
}
// out[i] = in[0] + ... + in[i]; in and out may be the same buffer only if ... they must not alias
static int mi_inclusive_scan(const uint32_t* in, uint32_t* out, int64_t n, hipStream_t s)
{
    if (n <= 0) return DPR_OK;
    const int64_t per = (int64_t)kIThreads * kScanItems;
    const int64_t nb = (n + per - 1) / per;
    uint32_t* bsum = nullptr;
    DPR_HIP(hipMalloc(&bsum, sizeof(uint32_t) * (size_t)nb));
    hipLaunchKernelGGL(mi_scan_sums_kernel, dim3((unsigned)nb), dim3(kIThreads), 0, s, in, n, bsum);
    hipLaunchKernelGGL(mi_scan_top_kernel, dim3(1), dim3(kIThreads), 0, s, bsum, nb);
    hipLaunchKernelGGL(mi_scan_apply_kernel, dim3((unsigned)nb), dim3(kIThreads), 0, s, in, n, (const uint32_t*)bsum, out);
    const hipError_t e = hipGetLastError();
    const hipError_t e2 = hipStreamSynchronize(s);       // (the block sums are released here)
    (void)hipFree(bsum);
    if (e != hipSuccess) return hip_fail(e, "mi_inclusive_scan");
    if (e2 != hipSuccess) return hip_fail(e2, "mi_inclusive_scan");
    return DPR_OK;
}

// dst (keys, payload) = every chunk's entries sorted by key (stable); src: the sketches (every tip's S values ascending) and
// their payloads.  k2 / p2: scratch of `total` entries each.
static int mi_sort_chunks(const uint64_t* sketches, const uint32_t* pay, uint64_t* ks, uint32_t* post, uint64_t* k2, uint32_t* p2,
                          int64_t total, int64_t seg, int S, hipStream_t s)
{
    int rounds = 0;
    for (int v = kIC; v > 1; v >>= 1) ++rounds;          // log2(kIC)
    for (int r = 0; r < rounds; ++r) {
        // round r writes buffer (rounds - 1 - r) & 1: the last round writes buffer 0 = (ks, post)
        const int dsti = (rounds - 1 - r) & 1;
        const uint64_t* sK = r == 0 ? sketches : (dsti ? ks : k2);
        const uint32_t* sP = r == 0 ? pay : (dsti ? post : p2);
        uint64_t* dK = dsti ? k2 : ks;
        uint32_t* dP = dsti ? p2 : post;
        const int64_t R = (int64_t)S << r;
        const int64_t ppc = (int64_t)(kIC >> (r + 1));
        const int64_t tpp = (2 * R + kMT - 1) / kMT;
        const int64_t chunks = (total + seg - 1) / seg;
        const int64_t blocks = chunks * ppc * tpp;
        if (blocks > (int64_t)0x7FFFFFFF) { set_error("mash_index_build: too many merge tiles"); return DPR_ERR_ARG; }
        hipLaunchKernelGGL(mi_merge_kernel, dim3((unsigned)blocks), dim3(kIThreads), 0, s, sK, sP, dK, dP, total, seg, R, ppc, tpp);
        DPR_HIP(hipGetLastError());
    }
    return DPR_OK;
}

// head flags of the sorted keys (chunks are fixed segments of 512 S entries)
__global__ __launch_bounds__(kIThreads) void mi_heads_kernel(const uint64_t* __restrict__ ks, int64_t total, int64_t seg,
                                                             uint32_t* __restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * kIThreads + threadIdx.x;
    if (i >= total) return;
    flag[i] = (i % seg == 0 || ks[i] != ks[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(kIThreads) void mi_compact_kernel(const uint64_t* __restrict__ ks, const uint32_t* __restrict__ flag,
                                                               const uint32_t* __restrict__ g, int64_t total,
                                                               uint64_t* __restrict__ uniq, uint32_t* __restrict__ off)
{
    const int64_t i = (int64_t)blockIdx.x * kIThreads + threadIdx.x;
    if (i >= total) return;
    if (flag[i]) { uniq[g[i] - 1] = ks[i]; off[g[i] - 1] = (uint32_t)i; }
    if (i == total - 1) off[g[i]] = (uint32_t)total;
}

// per chunk: first / one-past-last distinct value (global indices), directory shift
__global__ __launch_bounds__(kIThreads) void mi_chunks_kernel(const uint64_t* __restrict__ ks, const uint32_t* __restrict__ g,
                                                              int64_t total, int64_t seg, int64_t chunks,
                                                              uint32_t* __restrict__ ubase, int32_t* __restrict__ shift,
                                                              uint64_t* __restrict__ vmax)
{
    const int64_t c = (int64_t)blockIdx.x * kIThreads + threadIdx.x;
    if (c > chunks) return;
    if (c == chunks) { ubase[c] = g[total - 1]; return; }
    const int64_t start = c * seg, end = start + seg < total ? start + seg : total;
    ubase[c] = start ? g[start - 1] : 0u;
    const uint64_t vm = ks[end - 1];
    vmax[c] = vm;
    const int bits = 64 - __builtin_clzll(vm | 1ull);
    shift[c] = bits > kIBktLog ? bits - kIBktLog : 0;
}

// directory: bkt[c][b] = first distinct value (global index) of chunk c whose bucket is >= b; bkt[c][kINB] = end
__global__ __launch_bounds__(kIThreads) void mi_buckets_kernel(const uint64_t* __restrict__ uniq, const uint32_t* __restrict__ off,
                                                               const uint32_t* __restrict__ ubase, const int32_t* __restrict__ shift,
                                                               int64_t nu, int64_t seg, uint32_t* __restrict__ bkt)
{
    const int64_t u = (int64_t)blockIdx.x * kIThreads + threadIdx.x;
    if (u >= nu) return;
    const int64_t c = (int64_t)off[u] / seg;
    const int sh = shift[c];
    uint32_t* row = bkt + c * (int64_t)(kINB + 1);
    const int b = (int)(uniq[u] >> sh);
    const int bprev = (uint32_t)u == ubase[c] ? -1 : (int)(uniq[u - 1] >> sh);
    for (int bb = bprev + 1; bb <= b; ++bb) row[bb] = (uint32_t)u;
    if ((uint32_t)(u + 1) == ubase[c + 1])
        for (int bb = b + 1; bb <= kINB; ++bb) row[bb] = (uint32_t)(u + 1);
}

// Dense values: a value held by at least kIDenseMin tips of a chunk gets, besides its posting list, a block of kIC positions
// indexed by tip (65535 = absent) -- no larger than the list it replaces in the kernel's inner loop, read in tip order, and the
// counters are then touched in tip order too (conflict-free, no address arithmetic).
constexpr uint32_t kIDenseMin = 3 * kIC / 8;      // (a dense step costs ~7 wave instructions per 128 tips, a posting group ~10 per 64 entries)
__global__ __launch_bounds__(kIThreads) void mi_dense_flag_kernel(const uint32_t* __restrict__ off, int64_t nu, uint32_t* __restrict__ dflag)
{
    const int64_t u = (int64_t)blockIdx.x * kIThreads + threadIdx.x;
    if (u >= nu) return;
    dflag[u] = off[u + 1] - off[u] >= kIDenseMin ? 1u : 0u;
}
// dblk[u] = index of u's dense block or -1; posting i of a dense value stores its position into the block
__global__ __launch_bounds__(kIThreads) void mi_dense_index_kernel(const uint32_t* __restrict__ dflag, const uint32_t* __restrict__ dscan,
                                                                   int64_t nu, int32_t* __restrict__ dblk)
{
    const int64_t u = (int64_t)blockIdx.x * kIThreads + threadIdx.x;
    if (u >= nu) return;
    dblk[u] = dflag[u] ? (int32_t)(dscan[u] - 1u) : -1;
}
__global__ __launch_bounds__(kIThreads) void mi_dense_fill_kernel(const uint32_t* __restrict__ post, const uint32_t* __restrict__ g,
                                                                  const int32_t* __restrict__ dblk, int64_t total,
                                                                  uint16_t* __restrict__ dense)
{
    const int64_t i = (int64_t)blockIdx.x * kIThreads + threadIdx.x;
    if (i >= total) return;
    const int32_t b = dblk[g[i] - 1u];
    if (b < 0) return;
    const uint32_t ent = post[i];
    if (ent == kINone) return;
    dense[(int64_t)b * kIC + (ent >> 17)] = (uint16_t)(ent & 0xFFFFu);
}

// the distance of every possible count: same expression as the pair kernels of mash.hip (uni = S at the end of the merge)
__global__ void mi_dtab_kernel(int S, int k, double* __restrict__ dtab)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x > S) return;
    const double jac = fmax((double)x, 1.0) / (double)S;
    dtab[x] = fmin(1.0, fabs(log(2.0 * jac / (1.0 + jac)) / (double)k));
}

// ------------------------------------------------------------------------------------------------
// one wavefront (= one workgroup: the counters sit at a fixed LDS address) per (chunk, row): tasks chunk-major
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mash_dist_index_kernel(MashIndex ix, const uint64_t* __restrict__ sk, int S,
                                                             int64_t n, int64_t r0, int64_t nr, int64_t ncols,
                                                             int64_t cchunks, double* __restrict__ out, int64_t ld,
                                                             double* __restrict__ mir, int transposed, int rank, int world)
{
    __shared__ __attribute__((aligned(16))) uint16_t cnt[kIC];
    const int lane = threadIdx.x;
    for (int64_t task = (int64_t)blockIdx.x; task < cchunks * nr; task += (int64_t)gridDim.x) {
    const int64_t c = task / nr, t = task - c * nr;
    const int64_t i = r0 + t;                       // row tip
    if (i >= n) continue;
    const int64_t lim = ncols < i ? ncols : i;      // columns j < lim
    const int64_t j0 = c * kIC;
    if (j0 >= lim) continue;
#pragma unroll
    for (int m = 0; m < kIC / 64; ++m) cnt[lane + 64 * m] = 0;
    const uint64_t vm = ix.vmax[c];
    const int sh = ix.shift[c];
    const uint32_t* __restrict__ bkt = ix.bkt + c * (int64_t)(kINB + 1);
    const uint64_t* __restrict__ row = sk + i * S;
    const uint16_t* __restrict__ rmult = ix.mult + i * S;
    for (int b0 = 0; b0 < S; b0 += 64) {
        const int p = b0 + lane;
        const uint64_t v = p < S ? row[p] : 0ull;
        const int mu = p < S ? (int)rmult[p] : 0;
        uint32_t start = 0, len = 0;
        int dense = -1;
        if (mu > 0 && v <= vm) {                    // directory look-up of this lane's value
            const uint32_t bb = (uint32_t)(v >> sh);
            uint32_t lo = bkt[bb];
            const uint32_t hi0 = bkt[bb + 1];
            uint32_t hi = hi0;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ix.uniq[mid] < v) lo = mid + 1; else hi = mid;
            }
            if (lo < hi0 && ix.uniq[lo] == v) { start = ix.off[lo]; len = ix.off[lo + 1] - start; dense = ix.dblk[lo]; }
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(len > 0);
        // the row's values in ascending order
        // the loads of the NEXT value are issued while this one is applied: the first 64 postings, or -- dense value -- the whole
        // block of positions (4 words per lane: tips 2 lane, 2 lane + 1 of each quarter of the chunk)
        constexpr int kPQ = kIC / 128;
        auto first_load = [&](int l, uint32_t* w) {
            const int dn = __builtin_amdgcn_readlane(dense, l);
            if (dn >= 0) {
                const uint32_t* __restrict__ db = reinterpret_cast<const uint32_t*>(ix.dense + (int64_t)dn * kIC);
#pragma unroll
                for (int q = 0; q < kPQ; ++q) w[q] = db[64 * q + lane];
                return;
            }
            const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)start, l), ln = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
            w[0] = (uint32_t)lane < ln ? ix.post[st + lane] : kINone;
        };
        // Two values per step: their loads (issued a step ahead) are in flight together, and when both are dense the counters make
        // ONE round trip through LDS for the two updates (value A, then value B on A's result: the row's values in ascending order).
        typedef short pk16 __attribute__((ext_vector_type(2)));
        uint32_t* c32 = reinterpret_cast<uint32_t*>(cnt);
        auto one_value = [&](int l, const uint32_t* cur) {
            const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)start, l);
            const uint32_t ln = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
            const int m = __builtin_amdgcn_readlane(mu, l);
            const int dn = __builtin_amdgcn_readlane(dense, l);
            const int nb = b0 + l;                  // position of the value's first copy in the row sketch
            // reference's condition first_A(v) + nb - c < S  <=>  c - first_A(v) > nb - S
            const int K = nb - S;
            if (dn >= 0) {
                // Dense value: positions by tip, counters in tip order -- TWO tips per lane and step in packed 16-bit arithmetic
                // (round 5: this path is ~all of the work on clonal data; halves its LDS and vector instructions).  Counters and
                // positions stay below 4 096, K = nb - S lies in [-S, -1], an absent tip carries position 0x7F7F: every
                // difference fits 16 signed bits, and  K - (c - pos) < 0  <=>  c - pos > K  (absent: never).
                const pk16 Kp = { (short)K, (short)K }, mp = { (short)m, (short)m };
#pragma unroll
                for (int q = 0; q < kPQ; ++q) {
                    const uint32_t cw = c32[64 * q + lane];
                    const pk16 cv = __builtin_bit_cast(pk16, cw), pv = __builtin_bit_cast(pk16, cur[q]);
                    const pk16 t = Kp - (cv - pv);
                    const pk16 inc = (t >> 15) & mp;                     // all ones where the reference's condition holds
                    c32[64 * q + lane] = __builtin_bit_cast(uint32_t, (pk16)(cv + inc));
                }
                return;
            }
            auto apply = [&](uint32_t ent) {
                uint16_t* pc = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(cnt) + (ent >> 16));
                const int c0 = (int)*pc;
                if (c0 - (int)(ent & 0xFFFFu) > K) *pc = (uint16_t)(c0 + m);      // (tips of one posting list are distinct)
            };
            apply(cur[0]);
            const uint32_t* __restrict__ pl = ix.post + st;
            uint32_t e0 = 64;
            for (; e0 + 192 <= ln; e0 += 192) {                 // whole groups, three in flight: no bounds to check
                const uint32_t a0 = pl[e0 + (uint32_t)lane], a1 = pl[e0 + 64u + (uint32_t)lane], a2 = pl[e0 + 128u + (uint32_t)lane];
                apply(a0); apply(a1); apply(a2);
            }
            if (e0 < ln) {                                      // the rest: up to three groups, the last one partial
                uint32_t ent[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const uint32_t e = e0 + 64u * (uint32_t)q + (uint32_t)lane;
                    ent[q] = e < ln ? pl[e] : kINone;
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (e0 + 64u * (uint32_t)q < ln) apply(ent[q]);
            }
        };
        auto peek2 = [](unsigned long long t, int& a, int& b2) {
            a = t ? (int)__builtin_ctzll(t) : -1;
            const unsigned long long t2 = t & (t - 1);
            b2 = t2 ? (int)__builtin_ctzll(t2) : -1;
        };
        uint32_t preA[kPQ], preB[kPQ];
#pragma unroll
        for (int q = 0; q < kPQ; ++q) { preA[q] = kINone; preB[q] = kINone; }
        int la, lb;
        peek2(todo, la, lb);
        if (la >= 0) first_load(la, preA);
        if (lb >= 0) first_load(lb, preB);
        while (la >= 0) {
            const int ca = la, cb = lb;
            todo &= todo - 1;
            if (cb >= 0) todo &= todo - 1;
            uint32_t curA[kPQ], curB[kPQ];
#pragma unroll
            for (int q = 0; q < kPQ; ++q) { curA[q] = preA[q]; curB[q] = preB[q]; }
            peek2(todo, la, lb);
            if (la >= 0) first_load(la, preA);
            if (lb >= 0) first_load(lb, preB);
            const int dnA = __builtin_amdgcn_readlane(dense, ca);
            const int dnB = cb >= 0 ? __builtin_amdgcn_readlane(dense, cb) : -1;
            if (dnA >= 0 && dnB >= 0) {
                const int KA = b0 + ca - S, KB = b0 + cb - S;
                const int mA = __builtin_amdgcn_readlane(mu, ca), mB = __builtin_amdgcn_readlane(mu, cb);
                const pk16 KpA = { (short)KA, (short)KA }, mpA = { (short)mA, (short)mA };
                const pk16 KpB = { (short)KB, (short)KB }, mpB = { (short)mB, (short)mB };
#pragma unroll
                for (int q = 0; q < kPQ; ++q) {
                    const uint32_t cw = c32[64 * q + lane];
                    const pk16 cv = __builtin_bit_cast(pk16, cw);
                    const pk16 tA = KpA - (cv - __builtin_bit_cast(pk16, curA[q]));
                    const pk16 c1 = cv + ((tA >> 15) & mpA);
                    const pk16 tB = KpB - (c1 - __builtin_bit_cast(pk16, curB[q]));
                    const pk16 c2 = c1 + ((tB >> 15) & mpB);
                    c32[64 * q + lane] = __builtin_bit_cast(uint32_t, c2);
                }
            } else {
                one_value(ca, curA);
                if (cb >= 0) one_value(cb, curB);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < kIC / 64; ++m) {
        const int64_t j = j0 + lane + 64 * m;
        if (j < lim) {
            const int x = (int)cnt[lane + 64 * m];
            const double d = ix.dtab[x > 1 ? x : 1];
            if (world > 1) {
                // rows of a matrix sharded by row blocks: the pair (i, j), j < i, goes to row i where this rank owns it and, as
                // (j, i), to row j where it owns that one (every rank walks every row: the index yields a row against a CHUNK of
                // columns, so the pairs above the diagonal of an own row only come as mirrors of rows it does not own)
                if (shard_owner(i, world) == rank) mir[shard_local_row(i, world) * ld + j] = d;
                if (shard_owner(j, world) == rank) mir[shard_local_row(j, world) * ld + i] = d;
            } else {
                if (transposed) out[j * ld + t] = d; else out[t * ld + j] = d;
                if (mir) mir[j * ld + i] = d;
            }
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
void mash_index_free(MashIndex& ix)
{
    void* ptrs[] = { ix.post, ix.uniq, ix.off, ix.bkt, ix.ubase, ix.shift, ix.vmax, ix.mult, ix.dtab, ix.dblk, ix.dense };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    ix = MashIndex();
}

int mash_index_build(MashBuffers& m, hipStream_t s)
{
    mash_index_free(m.index);
    const int S = m.S;
    const int64_t n = m.n, total = n * S;
    if (S > 4096 || total >= (int64_t)0xFFFF0000ll || n < 2) return DPR_OK;      // not indexable: the other kernels take over
    MashIndex& ix = m.index;
    const int64_t seg = (int64_t)kIC * S, chunks = (n + kIC - 1) / kIC;
    const unsigned gt = (unsigned)((total + kIThreads - 1) / kIThreads);
    uint64_t* ks = nullptr;
    uint32_t *pay = nullptr, *flag = nullptr, *g = nullptr;
    auto cleanup = [&]() {
        void* ptrs[] = { ks, pay, flag, g };
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    };
    auto fail = [&](int rc) { cleanup(); mash_index_free(ix); return rc; };
#define MI_HIP(x)                                                       \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) return fail(hip_fail(e_, #x));            \
    } while (0)
    MI_HIP(hipMalloc(&ix.mult, sizeof(uint16_t) * (size_t)total));
    MI_HIP(hipMalloc(&ix.post, sizeof(uint32_t) * (size_t)total));
    MI_HIP(hipMalloc(&pay, sizeof(uint32_t) * (size_t)total));
    MI_HIP(hipMalloc(&ks, sizeof(uint64_t) * (size_t)total));
    hipLaunchKernelGGL(mi_payload_kernel, dim3(gt), dim3(kIThreads), 0, s, m.sketches, S, total, pay, ix.mult);
    MI_HIP(hipGetLastError());
    // sort every chunk's entries by value: nine rounds of pairwise merges of the (already ascending) sketches
    {
        uint64_t* k2 = nullptr;
        uint32_t* p2 = nullptr;
        MI_HIP(hipMalloc(&k2, sizeof(uint64_t) * (size_t)total));
        const hipError_t e2 = hipMalloc(&p2, sizeof(uint32_t) * (size_t)total);
        if (e2 != hipSuccess) { (void)hipFree(k2); return fail(hip_fail(e2, "hipMalloc(merge scratch)")); }
        const int rcs = mi_sort_chunks((const uint64_t*)m.sketches, pay, ks, ix.post, k2, p2, total, seg, S, s);
        const hipError_t es = hipStreamSynchronize(s);
        (void)hipFree(k2); (void)hipFree(p2);
        if (rcs != DPR_OK) return fail(rcs);
        if (es != hipSuccess) return fail(hip_fail(es, "mash_index_build: chunk sort"));
    }
    (void)hipFree(pay); pay = nullptr;
    // distinct values of every chunk and the start of their posting lists
    MI_HIP(hipMalloc(&flag, sizeof(uint32_t) * (size_t)total));
    MI_HIP(hipMalloc(&g, sizeof(uint32_t) * (size_t)total));
    hipLaunchKernelGGL(mi_heads_kernel, dim3(gt), dim3(kIThreads), 0, s, ks, total, seg, flag);
    MI_HIP(hipGetLastError());
    if (int rcq = mi_inclusive_scan((const uint32_t*)flag, g, total, s)) return fail(rcq);
    uint32_t nu = 0;
    MI_HIP(hipMemcpyAsync(&nu, g + (total - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    ix.nu = nu;
    MI_HIP(hipMalloc(&ix.uniq, sizeof(uint64_t) * (size_t)nu));
    MI_HIP(hipMalloc(&ix.off, sizeof(uint32_t) * ((size_t)nu + 1)));
    hipLaunchKernelGGL(mi_compact_kernel, dim3(gt), dim3(kIThreads), 0, s, ks, flag, g, total, ix.uniq, ix.off);
    MI_HIP(hipGetLastError());
    MI_HIP(hipMalloc(&ix.ubase, sizeof(uint32_t) * (size_t)(chunks + 1)));
    MI_HIP(hipMalloc(&ix.shift, sizeof(int32_t) * (size_t)chunks));
    MI_HIP(hipMalloc(&ix.vmax, sizeof(uint64_t) * (size_t)chunks));
    hipLaunchKernelGGL(mi_chunks_kernel, dim3((unsigned)((chunks + 1 + kIThreads - 1) / kIThreads)), dim3(kIThreads), 0, s, ks, g, total, seg,
                       chunks, ix.ubase, ix.shift, ix.vmax);
    MI_HIP(hipGetLastError());
    MI_HIP(hipMalloc(&ix.bkt, sizeof(uint32_t) * (size_t)(chunks * (kINB + 1))));
    hipLaunchKernelGGL(mi_buckets_kernel, dim3((unsigned)((nu + kIThreads - 1) / kIThreads)), dim3(kIThreads), 0, s, ix.uniq, ix.off, ix.ubase,
                       ix.shift, (int64_t)nu, seg, ix.bkt);
    MI_HIP(hipGetLastError());
    // dense blocks (flag and scan reuse the head-flag buffers: nu <= total)
    {
        const unsigned gu = (unsigned)((nu + kIThreads - 1) / kIThreads);
        hipLaunchKernelGGL(mi_dense_flag_kernel, dim3(gu), dim3(kIThreads), 0, s, ix.off, (int64_t)nu, flag);
        MI_HIP(hipGetLastError());
        uint32_t* dscan = nullptr;
        MI_HIP(hipMalloc(&dscan, sizeof(uint32_t) * (size_t)nu));
        hipError_t e1 = hipSuccess;
        if (int rcq = mi_inclusive_scan((const uint32_t*)flag, dscan, (int64_t)nu, s)) { (void)hipFree(dscan); return fail(rcq); }
        uint32_t nd = 0;
        if (e1 == hipSuccess) e1 = hipMemcpyAsync(&nd, dscan + (nu - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e1 == hipSuccess) e1 = hipStreamSynchronize(s);
        if (e1 == hipSuccess) e1 = hipMalloc(&ix.dblk, sizeof(int32_t) * (size_t)nu);
        if (e1 == hipSuccess) e1 = hipMalloc(&ix.dense, sizeof(uint16_t) * ((size_t)nd * kIC + 64));
        if (e1 == hipSuccess) e1 = hipMemsetAsync(ix.dense, 0x7f, sizeof(uint16_t) * ((size_t)nd * kIC + 64), s);      // absent: position 0x7F7F
        if (e1 == hipSuccess) {
            hipLaunchKernelGGL(mi_dense_index_kernel, dim3(gu), dim3(kIThreads), 0, s, flag, dscan, (int64_t)nu, ix.dblk);
            hipLaunchKernelGGL(mi_dense_fill_kernel, dim3(gt), dim3(kIThreads), 0, s, ix.post, g, ix.dblk, total, ix.dense);
            e1 = hipGetLastError();
        }
        if (e1 == hipSuccess) e1 = hipStreamSynchronize(s);
        (void)hipFree(dscan);
        if (e1 != hipSuccess) return fail(hip_fail(e1, "mash_index_build: dense blocks"));
        ix.ndense = nd;
    }
    MI_HIP(hipMalloc(&ix.dtab, sizeof(double) * (size_t)(S + 1)));
    hipLaunchKernelGGL(mi_dtab_kernel, dim3((unsigned)((S + 1 + 255) / 256)), dim3(256), 0, s, S, m.k, ix.dtab);
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(s));
#undef MI_HIP
    cleanup();
    ix.chunks = chunks;
    if (log_level("mash") > 0) {
        // (how much of the pair kernel's work takes the dense path: the postings of dense values)
        std::vector<uint32_t> hoff((size_t)nu + 1);
        std::vector<int32_t> hblk((size_t)nu);
        long long dense_post = 0;
        if (hipMemcpy(hoff.data(), ix.off, sizeof(uint32_t) * hoff.size(), hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(hblk.data(), ix.dblk, sizeof(int32_t) * hblk.size(), hipMemcpyDeviceToHost) == hipSuccess)
            for (size_t u = 0; u < hblk.size(); ++u)
                if (hblk[u] >= 0) dense_post += (long long)(hoff[u + 1] - hoff[u]);
        std::fprintf(stderr, "[mash] inverted index: %lld chunks of %d tips, %u distinct (chunk, value) pairs of %lld entries, %u of them dense "
                     "(%.1f %% of the entries)\n", (long long)chunks, kIC, nu, (long long)total, ix.ndense, 100.0 * (double)dense_post / (double)total);
    }
    return DPR_OK;
}

// rows r0 .. r0+nr x columns j < min(ncols, row): out[t*ld + j] (transposed: out[j*ld + t]); mir != nullptr: also mir[j*ld + row]
// (the matrix base: the mirror of a row block that does not start at row 0 lands outside the block).  world > 1: mir = this
// rank's row-sharded matrix; rows r0 .. r0+nr are GLOBAL rows, every rank walks all of them and keeps what it owns.
int mash_dist_index(const MashBuffers& m, int64_t r0, int64_t nr, int64_t ncols, double* out, int64_t ld, double* mir,
                    bool transposed, hipStream_t s, int rank, int world)
{
    const MashIndex& ix = m.index;
    if (!ix.post) { set_error("mash_dist_index: no index"); return DPR_ERR_STATE; }
    int64_t top = r0 + nr - 1 < ncols ? r0 + nr - 1 : ncols;       // columns any row of the batch can need: j < top
    if (top <= 0) return DPR_OK;
    const int64_t cchunks = (top + kIC - 1) / kIC;
    int64_t blocks = cchunks * nr;                  // one task per wavefront (a bounded grid walking the tasks: 25 % slower, uneven tasks)
    if (blocks > (int64_t)0x3FFFFFFF) blocks = 0x3FFFFFFF;
    // When tree kernels of a placement batch run beside this launch on another stream (share_chip), the grid is bounded instead:
    // 12 wavefronts per CU walk the tasks.  A grid of 100 000 small workgroups keeps the dispatcher busy and every wave slot
    // taken, and the tree kernels -- 2 launches per tip, up to 1 500 workgroups each -- then make NO progress beside it
    // (100 000 tips: distance 1.2 s + tree 2.0 s = 3.2 s, nothing hidden).
    constexpr int share_waves = 12;      // (round 5, every batch beside + tree kernels at wave priority 3: 100 000 tips, mean branch 2e-5 / 1e-3:
                                         //  8: 1.78 / 2.10 s, 12: 1.73 / 2.14, 16: 1.75 / 2.21, 20: 1.76 / 2.28, 24: 2.07 / 2.63)
    if (m.share_chip && share_waves > 0 && blocks > 256ll * share_waves) blocks = 256ll * share_waves;
    const size_t pad = 0;
    hipLaunchKernelGGL(mash_dist_index_kernel, dim3((unsigned)blocks), dim3(64), pad, s, ix, m.sketches, m.S, m.n, r0, nr, ncols, cchunks,
                       out, ld, mir, transposed ? 1 : 0, rank, world);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

}  // namespace dpr
