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
#include <cstdio>
#include <vector>

#include <cstdlib>
#include <cstring>

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
// Table formulation of the same distance (S <= 1024).
// The sequential merge is equivalent to: events in value order, B before A on ties; every A element
// and every B element whose value is not in A advances `uni`; B elements equal to an A value advance
// `inter`; stop at the S-th advance.  Hence, iterating over the DISTINCT values v of A (outer list)
// that also occur in B (inner list):
//     inter = sum of mult_B(v) over those v with   first_A(v) + #{b < v} - #{matched b < v}  <  S
// and uni = S: B is never walked, every A value only needs  rank_B(v) = #{b < v}  and  mult_B(v).
// Per resident row tip (the B list) the block keeps in LDS the sorted sketch and an ORDERED bucket
// table on the leading 9 significant bits: base[bucket] = number of values in lower buckets, and
// 8 x 15-bit tags per bucket (the next 15 bits of each value, ascending, in 16-bit fields).  One
// 16-byte read of the bucket of an A value gives rank_B(v) = base + #(tags below v's tag); the
// sketch entry at that rank decides the match (64-bit compare; equal tags of different values and
// duplicates are walked, buckets with more than 8 values are flagged and binary-searched -- all rare).
// A wave holds one column sketch (the A list) in registers, strided: step u covers the 64 consecutive
// values a[64u .. 64u+63], so (i) the lanes of one LDS instruction read ~33 consecutive buckets and
// ~64 consecutive sketch entries (near conflict-free, unlike random probing, which is what bounds a
// hash-probe formulation), (ii) the prefix of matches inside a step is a ballot/popcount and the
// running totals are wave-uniform.  Union ranks grow with v, so the wave stops at the first step whose
// smallest rank is already >= S (about half of the 16 steps).  16 waves share 8 resident rows.
// ------------------------------------------------------------------------------------------------
constexpr int kLS = 1024;             // largest sketch this kernel handles
constexpr int kTB = 512;              // ordered buckets per row (+1 catch-all above the row's maximum)
constexpr int kLRows = 8;             // row tips resident per block
constexpr int kLThreads = 1024;       // 16 waves
constexpr int kLColsPerBlock = 512;
constexpr int kTValsStride = kLS + 2;                              // u64 per row (2 pad entries = ~0)
constexpr int kTBaseStride = (kTB + 1 + 7) / 8 * 8;                // u16 per row
constexpr size_t kTLds = (size_t)kLRows * (sizeof(uint64_t) * kTValsStride + sizeof(uint4) * (kTB + 1) +
                                            sizeof(uint16_t) * kTBaseStride);

// 15 tag bits below the 9 bucket bits.  An empty entry is stored as tag 0x7FFF: like a real tag of that
// value it is never "below" anything, and equal tags are always told apart by the values themselves.
__device__ __forceinline__ uint32_t mash_tag15(uint64_t v, int tsh) { return (uint32_t)(v >> tsh) & 0x7FFFu; }

__device__ __forceinline__ int lds_lower_bound(const uint64_t* v, int S, uint64_t key)
{
    int lo = 0, hi = S;                      // first t with v[t] >= key
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Tags are 15 bits, stored with bit 15 set (0xFFFF = empty).  For m2 = the 15-bit tag of the A value in
// both halves, (stored - m2) never borrows across the halves and keeps bit 15 of a half exactly when
// that tag is >= m: eight tags are compared with 4 x (sub, and, bcnt).
__device__ __forceinline__ int tags_below(uint4 tg, uint32_t m2)
{
    const uint32_t H = 0x80008000u;
    int ge = __popc((tg.x - m2) & H);
    ge += __popc((tg.y - m2) & H);
    ge += __popc((tg.z - m2) & H);
    ge += __popc((tg.w - m2) & H);
    return 8 - ge;
}

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
    uint64_t* vals = reinterpret_cast<uint64_t*>(smem);                              // [kLRows][kTValsStride]
    uint4* tags = reinterpret_cast<uint4*>(vals + kLRows * kTValsStride);            // [kLRows][kTB + 1]
    uint16_t* base = reinterpret_cast<uint16_t*>(tags + kLRows * (kTB + 1));         // [kLRows][kTBaseStride]
    __shared__ int s_shift[kLRows], s_dups[kLRows];
    __shared__ int64_t s_irow[kLRows], s_rlim[kLRows];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int64_t t0, cbeg, cend;
    int64_t imax = -1;        // largest column bound of the resident rows
    const int32_t* colids = nullptr;
    // ---- resident rows: tip id (-1: none) and the bound on the column position (row r takes positions < rlim)
    if (JOBS) {
        const int4 job = J.jobs[blockIdx.x];
        const int ci = job.x, m = J.cl_m[ci];
        t0 = job.y; cbeg = job.z;
        colids = J.cols + J.cl_coff[ci];
        out = J.out + J.cl_out[ci];
        ld = J.cl_ld[ci];
        for (int r = 0; r < kLRows; ++r) {
            const int64_t tt = t0 + r;
            const int64_t rl = tt < m ? kDcLeaves + tt : -1;
            if (tid == 0) { s_irow[r] = tt < m ? J.members[J.cl_moff[ci] + tt] : -1; s_rlim[r] = rl; }
            imax = max(imax, rl);
        }
        cend = min(imax, cbeg + kLColsPerBlock);
    } else {
        t0 = (int64_t)blockIdx.y * kLRows;
        cbeg = (int64_t)blockIdx.x * kLColsPerBlock;
        cend = min(ncols, cbeg + kLColsPerBlock);
        for (int r = 0; r < kLRows; ++r) {
            const int64_t tt = t0 + r;
            int64_t i = -1;
            if (tt < nr) { i = r0 + tt; if (i >= n) i = -1; }
            if (tid == 0) { s_irow[r] = i; s_rlim[r] = i; }
            imax = max(imax, i);
        }
    }
    if (imax < 0 || cbeg >= imax) return;   // nothing to do in this column range (block-uniform)
    __syncthreads();

    // ---- resident row structures
    for (int e = tid; e < kLRows * kTValsStride; e += kLThreads) {
        const int r = e / kTValsStride, sidx = e % kTValsStride;
        const int64_t i = s_irow[r];
        vals[e] = (i >= 0 && sidx < S) ? sk[i * S + sidx] : ~0ull;
    }
    __syncthreads();
    if (tid < kLRows) {
        const uint64_t mx = vals[tid * kTValsStride + (S - 1)];
        const int bits = mx ? 64 - __clzll((long long)mx) : 1;
        s_shift[tid] = bits > 24 ? bits - 9 : 15;     // >= 15 so that the tag shift is never negative
        s_dups[tid] = 0;
    }
    __syncthreads();
    for (int e = tid; e < kLRows * kLS; e += kLThreads) {        // does the row hold a value twice?
        const int r = e / kLS, sidx = e % kLS;
        if (sidx + 1 < S && vals[r * kTValsStride + sidx] == vals[r * kTValsStride + sidx + 1]) s_dups[r] = 1;
    }
    for (int e = tid; e < kLRows * (kTB + 1); e += kLThreads) {
        const int r = e / (kTB + 1), kb = e % (kTB + 1);
        const uint64_t* v = vals + r * kTValsStride;
        const int sh = s_shift[r], tsh = sh - 15;
        int lo = S, hi = S;
        if (kb < kTB) {
            lo = lds_lower_bound(v, S, (uint64_t)kb << sh);
            hi = kb == kTB - 1 ? S : lds_lower_bound(v, S, (uint64_t)(kb + 1) << sh);
        }
        uint32_t t8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t8[q] = 0x8000u | (lo + q < hi ? mash_tag15(v[lo + q], tsh) : 0x7FFFu);
        tags[r * (kTB + 1) + kb] = make_uint4(t8[0] | (t8[1] << 16), t8[2] | (t8[3] << 16), t8[4] | (t8[5] << 16), t8[6] | (t8[7] << 16));
        base[r * kTBaseStride + kb] = (uint16_t)(lo | (hi - lo > 8 ? 0x8000 : 0));
    }
    __syncthreads();

    // ---- stream the columns: wave w takes column position c0 + w
    for (int64_t c0 = cbeg; c0 < cend && c0 < imax; c0 += kLThreads / 64) {
        const int64_t jpos = c0 + w;
        if (jpos >= cend || jpos >= imax) continue;
        const int64_t j = JOBS ? (int64_t)colids[jpos] : jpos;
        if (j < 0) continue;                                     // empty leaf-list entry (wave-uniform)
        uint64_t a[16];
        const uint64_t* col = sk + j * S + lane;
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = (64 * u + lane < S) ? col[64 * u] : ~0ull;
        // bit u of firstmask: a[64u + lane] exists and differs from its predecessor -- only the first
        // occurrence of a value in A can match (later copies find B exhausted)
        uint32_t firstmask = 0;
        {
            uint64_t carry = ~0ull;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                uint64_t prev = __shfl_up((unsigned long long)a[u], 1, 64);
                if (lane == 0) prev = carry;
                const bool first = (64 * u + lane < S) && ((64 * u + lane == 0) || prev != a[u]);
                firstmask |= (first ? 1u : 0u) << u;
                carry = __shfl((unsigned long long)a[u], 63, 64);
            }
        }
        int mycnt = 0;                                           // lane r keeps the count of row r
#pragma unroll 1
        for (int r = 0; r < kLRows; ++r) {
            if (s_irow[r] < 0 || jpos >= s_rlim[r]) continue;    // wave-uniform
            const uint64_t* v = vals + r * kTValsStride;
            const uint4* tg_r = tags + r * (kTB + 1);
            const uint16_t* base_r = base + r * kTBaseStride;
            const int sh = s_shift[r], tsh = sh - 15;
            const bool row_dups = s_dups[r] != 0;                // wave-uniform
            int cnt = 0, mtotal = 0;                             // wave-uniform running totals
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int pos = 64 * u + lane;
                const uint64_t av = a[u];
                // bucket = leading 9 significant bits (catch-all kTB above the row's range), tag = next 15 bits
                const uint64_t x = av >> tsh;
                const uint32_t xlo = (uint32_t)x;
                const int kb = ((uint32_t)(x >> 32) != 0u || xlo >= ((uint32_t)kTB << 15)) ? kTB : (int)(xlo >> 15);
                const uint4 tg = tg_r[kb];
                const uint32_t bs = base_r[kb];
                const uint32_t mt = xlo & 0x7FFFu;
                int rankb = (int)(bs & 0x7FFFu) + tags_below(tg, mt | (mt << 16));
                if (__any((bs & 0x8000u) != 0u)) {               // a crowded bucket in this step (rare)
                    if (bs & 0x8000u) rankb = lds_lower_bound(v, S, av);
                }
                uint64_t b0 = v[rankb];
                if (__any(b0 < av)) {                            // equal tags of different values (rare)
                    while (b0 < av) b0 = v[++rankb];
                }
                const bool match = ((firstmask >> u) & 1u) && b0 == av;
                const unsigned long long mb = __ballot(match);
                int madd;
                if (row_dups) {                                  // multiplicities in B: full prefix sum
                    int mult = 0;
                    if (match) { mult = 1; while (rankb + mult < S && v[rankb + mult] == av) ++mult; }
                    int incl = mult;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int x = __shfl_up(incl, off, 64);
                        if (lane >= off) incl += x;
                    }
                    madd = __shfl(incl, 63, 64);
                    const bool ok = match && (pos + rankb - (mtotal + incl - mult) < S);
                    int c = ok ? mult : 0;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
                    cnt += c;
                } else {
                    madd = __popcll(mb);
                    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
                    const bool ok = match && (pos + rankb - (mtotal + below) < S);
                    cnt += __popcll(__ballot(ok));
                }
                mtotal += madd;
                // ranks grow with the value: the next step starts at rank >= 64(u+1) + #{b < a[64u+63]} - mtotal
                const int rb63 = __builtin_amdgcn_readlane(rankb, 63);
                if (64 * (u + 1) >= S || 64 * (u + 1) + rb63 - mtotal >= S) break;
            }
            if (lane == r) mycnt = cnt;
        }
        if (lane < kLRows && s_irow[lane] >= 0 && jpos < s_rlim[lane]) {
            const double jac = fmax((double)mycnt, 1.0) / S;
            const double d = fmin(1.0, fabs(log(2.0 * jac / (1.0 + jac)) / (double)k));
            const int64_t t = t0 + lane;
            if (JOBS) out[t * ld + jpos] = d;
            else {
                if (transposed) out[j * ld + t] = d; else out[t * ld + j] = d;
                if (mirror) out[j * ld + (r0 + t)] = d;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Run-encoded pair kernel (round 2; default while sketches resemble each other).
// On the inputs the reference is run on (near-clonal alignments, scripts/alisim.sh) two sketches share
// almost all of their 1000 values, and the reference's loop spends its 2000 steps finding that out again
// for every pair.  Here every sketch is encoded ONCE against a common reference list R (the distinct values
// of one of the sketches, picked for short encodings) as a sequence of tokens:
//     RUN(p, len)   the next len values are R[p], R[p+1], ... (first copies only)
//     LIT(v, rk, eq) one value v that does not continue a run; rk = #{r in R : r < v}, eq = (R[rk] == v,
//                   i.e. an extra copy of a reference value)
// and the pair loop -- the reference's own event loop (src/mash.cu:437-454: consume b while b <= a, count
// uni / inter, stop at uni == S) -- runs over tokens: two runs at the same reference index are L matching
// pairs at once (inter += L, uni += L), runs at different indices skip to the later index with uni += gap,
// a literal against a run compares rk with the run's index.  Every comparison of two run elements or of a
// run element with a literal is an integer comparison in reference-index space (only two literals compare
// their 64-bit values), and the cut-off is exact because every bulk step is capped at S - uni.  Same
// counts as the literal loop bit for bit -- duplicates, padding values and S of any size included -- in
// ~(tokens of A + tokens of B) steps instead of ~2 S.
// A lane owns one pair: 64 consecutive row tips against one column tip at a time, the token arrays of the
// rows (a few hundred bytes each on clonal data) stay cache-resident over the column loop.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kTokLit = 1u, kTokEq = 2u;
constexpr int kTWaves = 4;               // waves (64-row tiles) per block

// one thread per sketch
// (output slot q holds sketch q * stride: stride > 1 encodes a sample)
__global__ __launch_bounds__(kThreads) void mash_encode_kernel(const uint64_t* __restrict__ sk, int S, int64_t n, int64_t stride,
                                                               const uint64_t* __restrict__ R, int nR,
                                                               uint4* __restrict__ tokens, int32_t* __restrict__ tok_cnt)
{
    const int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (q >= n) return;
    const uint64_t* X = sk + q * stride * S;
    uint4* out = tokens + q * S;
    int nt = 0, i = 0;
    while (i < S) {
        const uint64_t x = X[i];
        int lo = 0, hi = nR;                          // rk = #{r < x}
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (R[mid] < x) lo = mid + 1; else hi = mid; }
        const int rk = lo;
        const bool inR = rk < nR && R[rk] == x;
        if (inR && (i == 0 || X[i - 1] != x)) {       // first copy of a reference value: a run starts
            int len = 1;
            while (i + len < S && rk + len < nR && X[i + len] == R[rk + len]) ++len;
            out[nt++] = make_uint4((uint32_t)rk, 0u, (uint32_t)len, 0u);
            i += len;
        } else {
            out[nt++] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)rk, kTokLit | (inR ? kTokEq : 0u));
            ++i;
        }
    }
    tok_cnt[q] = nt;
}

// distinct values of a sorted list (one block; nR via counter)
__global__ void mash_dedup_kernel(const uint64_t* __restrict__ x, int S, uint64_t* __restrict__ R, int* __restrict__ nR)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int k = 0;
    for (int i = 0; i < S; ++i)
        if (i == 0 || x[i] != x[i - 1]) R[k++] = x[i];
    *nR = k;
}

// Cursor over a token sequence.  The head element is described in REFERENCE-INDEX space by an integer key:
// run element R[p] -> 2p + 1; literal with rank rk -> 2 rk + 1 if it equals R[rk] (an extra copy), else 2 rk (strictly
// between R[rk-1] and R[rk]).  Keys order the values; equal odd keys are equal values; only two literals with the same
// even key have to compare their 64-bit values.
struct TokCursor {
    const uint4* p;        // next token to fetch
    int left;              // tokens not fetched yet
    uint4 nxt;             // prefetched token
    int key, rem;          // key of the head element, elements left in the segment
    bool run, done;
    uint64_t val;          // literal value
    __device__ __forceinline__ void fetch()
    {
        if (left > 0) { nxt = *p; ++p; --left; } else { nxt = make_uint4(0u, 0u, 0u, 0xffffffffu); }
    }
    __device__ __forceinline__ void advance_segment()      // make the prefetched token current
    {
        const uint4 t = nxt;
        done = t.w == 0xffffffffu;
        const bool lit = (t.w & kTokLit) != 0u;
        run = !lit;
        val = ((uint64_t)t.y << 32) | t.x;
        key = lit ? 2 * (int)t.z + ((t.w & kTokEq) ? 1 : 0) : 2 * (int)t.x + 1;
        rem = lit ? 1 : (int)t.z;
        fetch();
    }
    __device__ __forceinline__ void init(const uint4* tok, int cnt)
    {
        p = tok; left = cnt;
        fetch();
        advance_segment();
    }
    __device__ __forceinline__ void consume(int L)          // 0 <= L <= rem
    {
        rem -= L;
        key += run ? 2 * L : 0;
        if (L > 0 && rem == 0) advance_segment();
    }
};

__global__ __launch_bounds__(kTWaves * 64) void mash_dist_tokens_kernel(const uint4* __restrict__ tokens, const int32_t* __restrict__ tok_cnt,
                                                                        int S, int k, int64_t n, int64_t r0, int64_t nr, int64_t ncols,
                                                                        double* __restrict__ out, int64_t ld, int mirror, int transposed,
                                                                        int cols_per_block)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t = ((int64_t)blockIdx.y * kTWaves + w) * 64 + lane;      // row index inside the batch
    const int64_t i = r0 + t;                                                // row tip id
    const bool row_ok = t < nr && i < n;
    const int64_t tile_last = r0 + ((int64_t)blockIdx.y * kTWaves + w) * 64 + 63;   // wave-uniform
    const int64_t cbeg = (int64_t)blockIdx.x * cols_per_block;
    int64_t cend = cbeg + cols_per_block;
    if (cend > ncols) cend = ncols;
    if (cend > tile_last) cend = tile_last;                                  // columns below the tile's last row only
    const uint4* tokB = tokens + (row_ok ? i : 0) * S;
    const int cntB = row_ok ? tok_cnt[i] : 0;
    TokCursor B0;                                                            // this lane's row at its first element (same for every column)
    B0.init(tokB, cntB);
    for (int64_t j = cbeg; j < cend; ++j) {
        const bool pair_ok = row_ok && j < i;
        TokCursor A, B = B0;                                                 // A: column j (the outer list), B: this lane's row
        A.init(tokens + j * S, tok_cnt[j]);
        int uni = 0, inter = 0;
        bool alive = pair_ok;
        int guard = 4 * S + 64;                                              // every step consumes an element: never reached
        while (__builtin_amdgcn_ballot_w64(alive) && --guard > 0) {
            // one step of the reference's loop per lane, written without branches on the token kinds (the lanes of a wave
            // are in different states; a branch per kind made every step cost the sum of all of them)
            const int cap = S - uni;                                         // > 0 for a live lane
            const int kA = A.key, kB = B.key;
            const bool bdone = B.done;
            const bool eqk = !bdone & (kB == kA);
            const bool odd = (kA & 1) != 0;
            const bool vlt = B.val < A.val, veq = B.val == A.val;           // decide only between two literals of one even key
            const bool b_less = (!bdone & (kB < kA)) | (eqk & !odd & vlt);
            const bool equal = eqk & (odd | veq);
            const bool a_less = !(b_less | equal);                           // b > a, or no b left
            const bool pairs = equal & A.run & B.run;
            const int Lb = B.run ? (kA - kB + 1) >> 1 : 1;                   // B elements below a
            const int La = bdone ? A.rem : (A.run ? (kB - kA + 1) >> 1 : 1); // A elements below b (all of them if B is exhausted)
            int L = pairs ? (A.rem < B.rem ? A.rem : B.rem) : b_less ? (Lb < B.rem ? Lb : B.rem) : a_less ? (La < A.rem ? La : A.rem) : 1;
            L = L < cap ? L : cap;
            // (b == a, then a) pairs: the a of a pair may only go once the NEXT b is known to be larger -- inside B's run it
            // is (the next reference value), behind the run's last element it is not (an extra copy of that value may follow)
            const bool bend = pairs & (L == B.rem);
            const int Lp = L - (bend ? 1 : 0);
            int cA = 0, cB = 0, du = 0, di = 0;
            if (pairs) {
                di = Lp; du = Lp; cA = Lp; cB = Lp;
                if (bend & (uni + Lp < S)) { di += 1; cB += 1; }
            } else if (b_less) { du = L; cB = L; }
            else if (a_less) { du = L; cA = L; }
            else { di = 1; cB = 1; }                                         // b == a with a literal involved: the b goes, a stays
            if (!alive) { cA = 0; cB = 0; du = 0; di = 0; }                  // a finished lane keeps its counts
            uni += du; inter += di;
            A.consume(cA);
            B.consume(cB);
            if (alive) alive = uni < S && !A.done;
        }
        if (pair_ok) {
            const double jac = fmax((double)inter, 1.0) / (double)uni;
            const double d = fmin(1.0, fabs(log(2.0 * jac / (1.0 + jac)) / (double)k));
            if (transposed) out[j * ld + t] = d; else out[t * ld + j] = d;
            if (mirror) out[j * ld + i] = d;
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
    void* ptrs[] = { m.packed2, m.word_off, m.len, m.sketches, m.tokens, m.tok_cnt, m.ref };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    mash_index_free(m.index);
    m = MashBuffers();
}

// Kernel choice for the row-against-columns shapes (read at every call: the tests switch kernels inside one process).
// The inverted index (mash_index.hip) wherever it can be built: 4.4-10 G pairs/s at any divergence (20 000 reads x 3 kb);
// without it (no memory, more than 2^32 sketch values, DPR_MASH_KERNEL=noindex) the run-encoded tokens while the sketches resemble the
// reference list (3.4 G pairs/s at 13 tokens per sketch, 0.8 G at 97; up to 150 tokens), else the bucket tables (0.4-0.7 G),
// which also serve the cluster jobs.  ONE switch for tests and A/B runs, DPR_MASH_KERNEL = auto (default) | index (always) |
// noindex (the automatic choice without the index) | tokens | table | literal (the reference's merge, one thread per pair).
enum MashKernel { kMkAuto = 0, kMkIndex, kMkNoIndex, kMkTokens, kMkTable, kMkLiteral };
static int mash_kernel_choice()
{
    const char* e = std::getenv("DPR_MASH_KERNEL");
    if (!e) return kMkAuto;
    const char* names[] = { "auto", "index", "noindex", "tokens", "table", "literal" };
    for (int k = 0; k < 6; ++k)
        if (std::strcmp(e, names[k]) == 0) return k;
    return kMkAuto;
}
static bool mash_tok_forced() { return mash_kernel_choice() == kMkTokens; }
static double mash_tok_max()        // tokens per sketch up to which the token kernel is used
{
    const int c = mash_kernel_choice();
    return c == kMkTokens ? 1e300 : (c == kMkTable || c == kMkLiteral) ? -1.0 : 150.0;
}
static int mash_index_policy()      // 1 always, 0 never, -1 automatic
{
    const int c = mash_kernel_choice();
    return c == kMkIndex ? 1 : c == kMkAuto ? -1 : 0;
}
static bool mash_indexable(const MashBuffers& m) { return m.S <= 4096 && m.n >= 2 && m.n * (int64_t)m.S < (int64_t)0xFFFF0000ll; }

// run encoding of all sketches against the distinct values of sketch 0 (see mash_dist_tokens_kernel)
static int mash_encode(MashBuffers& m, hipStream_t s)
{
    if (mash_kernel_choice() == kMkTable || mash_kernel_choice() == kMkLiteral) return DPR_OK;      // (A/B runs: no token kernel at all)
    const int S = m.S;
    DPR_HIP(hipMalloc(&m.tokens, sizeof(uint4) * (size_t)(m.n * S)));
    DPR_HIP(hipMalloc(&m.tok_cnt, sizeof(int32_t) * (size_t)m.n));
    DPR_HIP(hipMalloc(&m.ref, sizeof(uint64_t) * (size_t)(S + 1)));
    int* d_nr = nullptr;
    DPR_HIP(hipMalloc(&d_nr, sizeof(int)));
    // The reference list: any sketch gives exact results, one in the middle of the data gives short encodings.  A few
    // candidates (evenly spaced tips) are tried on a sample of the sketches; the one with the fewest tokens is kept.
    int64_t best_tip = 0;
    {
        const int64_t neval = m.n < 256 ? m.n : 256, ncand = m.n < 16 ? m.n : 16;
        const int64_t estride = m.n / neval;
        std::vector<int32_t> cnt((size_t)neval);
        double best = 1e300;
        for (int64_t c = 0; c < ncand; ++c) {
            const int64_t tip = c * (m.n / ncand) + (c ? (m.n / ncand) / 2 : 0);          // tip 0 first (the only candidate of tiny inputs)
            hipLaunchKernelGGL(mash_dedup_kernel, dim3(1), dim3(64), 0, s, m.sketches + tip * S, S, m.ref, d_nr);
            int nRc = 0;
            DPR_HIP(hipMemcpyAsync(&nRc, d_nr, sizeof(int), hipMemcpyDeviceToHost, s));
            DPR_HIP(hipStreamSynchronize(s));
            hipLaunchKernelGGL(mash_encode_kernel, dim3((unsigned)((neval + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, m.sketches, S, neval,
                               estride, m.ref, nRc, m.tokens, m.tok_cnt);
            DPR_HIP(hipMemcpyAsync(cnt.data(), m.tok_cnt, sizeof(int32_t) * (size_t)neval, hipMemcpyDeviceToHost, s));
            DPR_HIP(hipStreamSynchronize(s));
            double sum = 0;
            for (int32_t v : cnt) sum += v;
            if (sum < best) { best = sum; best_tip = tip; }
        }
    }
    hipLaunchKernelGGL(mash_dedup_kernel, dim3(1), dim3(64), 0, s, m.sketches + best_tip * S, S, m.ref, d_nr);
    int nR = 0;
    DPR_HIP(hipMemcpyAsync(&nR, d_nr, sizeof(int), hipMemcpyDeviceToHost, s));
    DPR_HIP(hipStreamSynchronize(s));
    (void)hipFree(d_nr);
    m.ref_n = nR;
    hipLaunchKernelGGL(mash_encode_kernel, dim3((unsigned)((m.n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, m.sketches, S, m.n,
                       (int64_t)1, m.ref, nR, m.tokens, m.tok_cnt);
    DPR_HIP(hipGetLastError());
    // tokens per sketch over a sample: the token kernel pays while the sketches resemble the reference
    const int64_t ns = m.n < 4096 ? m.n : 4096;
    std::vector<int32_t> cnt((size_t)ns);
    const int64_t step = m.n / ns;
    if (step <= 1) {
        DPR_HIP(hipMemcpyAsync(cnt.data(), m.tok_cnt, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost, s));
    } else {
        DPR_HIP(hipMemcpy2DAsync(cnt.data(), sizeof(int32_t), m.tok_cnt, sizeof(int32_t) * (size_t)step, sizeof(int32_t), (size_t)ns,
                                 hipMemcpyDeviceToHost, s));
    }
    DPR_HIP(hipStreamSynchronize(s));
    double sum = 0;
    for (int32_t c : cnt) sum += c;
    m.tok_mean = sum / (double)ns;
    if (log_level("mash") > 0) {
        int32_t mx = 0;
        for (int32_t c : cnt) mx = c > mx ? c : mx;
        std::fprintf(stderr, "[mash] run encoding against the sketch of tip %lld (%d distinct values): %.1f tokens per sketch (sample of %lld, most %d)\n",
                     (long long)best_tip, nR, m.tok_mean, (long long)ns, mx);
    }
    return DPR_OK;
}

int mash_sketch(MashBuffers& m, int k, int S, hipStream_t s)
{
    if (k < 2 || k > 15) { set_error("kmer size must be in [2,15]"); return DPR_ERR_ARG; }
    if (S < 1 || S > 4096) { set_error("sketch size must be in [1,4096]"); return DPR_ERR_ARG; }
    void* olds[] = { m.sketches, m.tokens, m.tok_cnt, m.ref };
    for (void* q : olds) if (q) (void)hipFree(q);
    m.sketches = nullptr; m.tokens = nullptr; m.tok_cnt = nullptr; m.ref = nullptr; m.tok_mean = 0.0;
    mash_index_free(m.index);
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
    const int want = mash_index_policy();
    if (want == 1 || (want < 0 && mash_indexable(m))) {
        const int rc = mash_index_build(m, s);
        if (rc != DPR_OK && want == 1) return rc;
        if (rc != DPR_OK) (void)hipGetLastError();      // (no memory for the index: tokens / tables take over)
    }
    // token encoding only where it can be used: no index, or a threshold given explicitly
    if (!m.index.post || mash_tok_forced())
        if (int rc = mash_encode(m, s)) return rc;
    return DPR_OK;
}

static int lookup_attr()
{
    static bool attr_set = false;
    const size_t tlds = kTLds;
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
    const size_t tlds = kTLds;
    hipLaunchKernelGGL(mash_dist_lookup_kernel<true>, dim3((unsigned)njobs), dim3(kLThreads), tlds, s, m.sketches, m.S,
                       m.k, m.n, (int64_t)0, (int64_t)0, (int64_t)0, (double*)nullptr, (int64_t)0, 0, 0, J);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int mash_dist_rows(const MashBuffers& m, int64_t r0, int64_t nr, int rank, int world, bool full,
                   int64_t ncols, double* out, int64_t ld, hipStream_t s, bool transposed)
{
    if (nr <= 0 || ncols <= 0) return DPR_OK;
    // lower-triangle pairs (placement batches; single-GPU NJ with a mirror write): token kernel while the sketches
    // resemble each other (at most 150 tokens per sketch on average, of up to S: the measured break-even with the table kernel), else the table kernel
    // (round 6: a row block that does not start at row 0 -- the whole matrix above 32 768 tips is built in such blocks -- mirrors
    //  through the matrix base, out - r0 * ld; before, only the first block took the index / token / lookup kernels and the rest
    //  the literal one, ~30 x slower)
    const bool mirror = full && world == 1;
    double* const mir = mirror ? out - r0 * ld : nullptr;
    const bool tokens_ok = m.tokens && m.tok_mean <= mash_tok_max();
    const bool use_index = m.index.post && mash_index_policy() != 0 && (mash_index_policy() == 1 || !(mash_tok_forced() && tokens_ok));
    // transposed output exists in the index, token and lookup kernels only: reject the call unless one of them WILL be
    // selected below (the literal row kernel at the end ignores `transposed`; a sketch size above the lookup tables with
    // the index switched off and divergent reads used to fall through to it and lay the block out row-major)
    if (transposed && (full || world > 1 || !(use_index || tokens_ok || m.S <= kLS))) {
        set_error("mash_dist_rows: transposed output needs the index, the token or the lookup kernel (sketch size above 1024 with "
                  "the index unavailable and dissimilar sketches)");
        return DPR_ERR_ARG;
    }
    if (use_index && (!full || mirror) && world <= 1)
        return mash_dist_index(m, r0, nr, ncols, out, ld, mir, transposed, s, 0, 1);
    if (tokens_ok && (!full || (mirror && r0 == 0)) && world <= 1) {
        // columns a wave walks through: 128 for a whole matrix, fewer when the launch has few row tiles (placement batches
        // of 256 rows), so that it still fills the chip (>= ~4096 waves) and no wave runs long after the others
        const int64_t tiles = (nr + 63) / 64;
        int64_t cpb = ncols * tiles / 4096;
        cpb = cpb < 8 ? 8 : (cpb > 128 ? 128 : cpb);
        dim3 tgrid((unsigned)((ncols + cpb - 1) / cpb), (unsigned)((nr + kTWaves * 64 - 1) / (kTWaves * 64)));
        hipLaunchKernelGGL(mash_dist_tokens_kernel, tgrid, dim3(kTWaves * 64), 0, s, m.tokens, m.tok_cnt, m.S, m.k, m.n, r0, nr, ncols,
                           out, ld, mirror ? 1 : 0, transposed ? 1 : 0, (int)cpb);
        DPR_HIP(hipGetLastError());
        return DPR_OK;
    }
    if (m.S <= kLS && (!full || (mirror && r0 == 0)) && world <= 1 && (transposed || mash_kernel_choice() != kMkLiteral)) {
        if (int rc = lookup_attr()) return rc;
        const size_t tlds = kTLds;
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

__global__ __launch_bounds__(kThreads) void mash_zero_diag_kernel(double* __restrict__ D, int64_t ld, int64_t rows_local, int rank, int world, int64_t n)
{
    const int64_t l = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (l >= rows_local) return;
    const int64_t g = shard_global_row(l, rank, world);
    if (g < n) D[l * ld + g] = 0.0;
}

// Whole matrix, rows sharded over `world` ranks by row blocks.  With the inverted index (the default) every rank walks ALL rows
// against all column chunks and keeps the pairs of the rows it owns, direct and mirrored (mash_dist_index_kernel): the index
// yields (row, chunk of columns below it), so the part of an own row above the diagonal only exists as the mirror of rows owned
// by others -- redundant over the ranks, and still several times faster than the literal pair kernel on 1 / world of the pairs
// (13 G against 0.4 G pairs per second).  Without the index: the literal kernel on the own rows, as before.
int mash_dist_matrix_sharded(const MashBuffers& m, int rank, int world, int64_t rows_local, double* D_local, int64_t ld, hipStream_t s)
{
    const bool use_index = m.index.post && mash_index_policy() != 0;
    if (!use_index) {
        for (int64_t r0 = 0; r0 < rows_local; r0 += 32768) {
            const int64_t nr = rows_local - r0 < 32768 ? rows_local - r0 : 32768;
            if (int rc = mash_dist_rows(m, r0, nr, rank, world, true, m.n, D_local + r0 * ld, ld, s)) return rc;
        }
        return DPR_OK;
    }
    if (rows_local > 0) {
        hipLaunchKernelGGL(mash_zero_diag_kernel, dim3((unsigned)((rows_local + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, D_local, ld, rows_local, rank, world, m.n);
        DPR_HIP(hipGetLastError());
    }
    for (int64_t g0 = 0; g0 < m.n; g0 += 32768) {
        const int64_t nr = m.n - g0 < 32768 ? m.n - g0 : 32768;
        if (int rc = mash_dist_index(m, g0, nr, m.n, nullptr, ld, D_local, false, s, rank, world)) return rc;
    }
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
