// Aligned-sequence (MSA) pairwise distances on gfx950.  Replaces MSADeviceArrays
// (src/MSA.cu:14-72), calculateParamsParallel (:103-156) and MSADistConstruction (:214-268), which
// compute ONE row per launch with one block per pair, by an all-pairs tiled kernel.
//
// Input layout (host ABI): 4-bit codes, 16 bases per uint64, as fourBitCompressor produces.
// Device layout: four bit planes per sequence, 32 bases per uint32 word:
//   X (1 = not a base: code >= 4 or position >= L), LO (code & 1), HI (code >> 1 & 1), both 0 where X,
//   and LX = LO | X (the column-side variant of LO).
// For a pair (row a, column b) per 32 bases, with the row using (X, LO, HI) and the column (X, LX, HI):
//   both_invalid += popc(Xa & Xb)                           -> useful = sites - both_invalid
//   mismatch     += popc((LOa^LXb) | (HIa^HIb) | (Xa^Xb))   -> match  = sites - mismatch
// Any invalid side forces a mismatch (one invalid: Xa^Xb; both: LOa = 0 against LXb = 1), so these equal
// the reference's  (a<4 || b<4)  and  (a<4 && a==b)  counts exactly, in 7 instead of 8 integer
// operations per word pair (xor, xor, xor, or3, popcount+add; and, popcount+add).
// The kernel is integer-VALU/LDS bound; HBM only sees the N^2 fp64 output.
//
// Round 6 (tools/valu_probe.hip measured the instruction costs on gfx950: v_xor / v_and / v_or 2 cycles per wave instruction with
// two or more waves per SIMD, v_bcnt_u32_b32 and v_or3_b32 -- VOP3 encodings -- 4.3): the 7-operation body costs 20.9 cycles per
// 32 sites of 64 pairs, of which the not-a-base bookkeeping (Xa ^ Xb, Xa & Xb and its popcount) is 8.3.  It is skipped wherever no
// sequence involved has a not-a-base position: mismatch = popc((LOa ^ LOb) | (HIa ^ HIb)), both_invalid += 0 -- the same counts,
// 10.3 cycles.  Two levels: a 16-word STAGE in which no sequence of the tile has one (MsaBuffers::xstage, a bit per sequence and
// stage) stages two planes per side instead of three; in the other stages each wavefront takes the short body for the WORDS in
// which none of its 16 rows and none of the 64 columns has one (bit masks filled while staging).  A per-pair correction of listed
// words was built and measured first: at two such words per sequence its scattered loads cost more than the bookkeeping it
// replaced (25.5 against 16.1 ms per 5 120 x 50 000 block, profiles/r6/msa_block_variants.txt); not kept.
// Staging moves four words per load (a thread owns one sequence and one word quad of every plane).  Distances
// of pairs whose useful count is within kMsaBand of L come from a (L - useful, match) table built with the epilogue's own function
// (the band of the (useful, match) table that short alignments use in full).
#include "dpr_internal.hpp"

namespace dpr {

constexpr int kKC = 16;   // plane words (32 bases each) staged per step
#ifndef MSA_UNROLL
#define MSA_UNROLL 2
#endif
#ifndef MSA_WAVES
#define MSA_WAVES 4      // waves per SIMD the pair kernels are compiled for (128 registers): a wave alone on its SIMD issues a vector
#endif                   // instruction every 4 cycles, two ready ones every 2 (tools/valu_probe.hip); 182 registers -> 2 waves cost 18 %

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
        planes[(0 * n + s) * W32 + w] = ~V;
        planes[(1 * n + s) * W32 + w] = LO;
        planes[(2 * n + s) * W32 + w] = HI;
        planes[(3 * n + s) * W32 + w] = LO | ~V;
    }
}

__device__ __forceinline__ double msa_epilogue(int useful, int match, int dist_type)
{
    // src/MSA.cu:233-235
    const double uncor = 1 - double(match) / useful;
    if (dist_type == DPR_DIST_UNCORRECTED) return uncor;
    return -0.75 * log(1.0 - uncor / 0.75);
}

// ------------------------------------------------------------------------------------------------
// Pair counters.  Types 1-2: useful / match (above).  Types 3-6 (Tajima-Nei, K2P, Tamura,
// Jin-Nei): formulas and counters of src/divide_and_conquer/msa.cu:107-217 (counts), :238-265
// (epilogues); the copies in src/MSA.cu index with the wrong variable.  Only sites valid
// in BOTH sequences count.  With the planes (code = 2*HI + LO; A,C,G,T = 0..3), per 32 bases of a
// (row r, column c) pair:
//   both = Vr & Vc                      tot   += popc(both)
//   eq   = both & ~((LOr^LOc)|(HIr^HIc))          (match)
//   transitions  p: mismatch & ~(LOr^LOc)   (same parity), transversions q: mismatch & (LOr^LOc)
//   C or G       : HI ^ LO               (Tamura's gc1 = row, gc2 = column, mismatching sites only)
//   Tajima-Nei pair classes {A,G},{A,T},{C,G},{C,T} and base counts over both sequences.
// ------------------------------------------------------------------------------------------------
template <int TYPE>
struct PairCounts {
    int tot = 0, eq = 0, p = 0, q = 0, gc1 = 0, gc2 = 0;
    int fA = 0, fC = 0, fG = 0;   // base counts over both sequences (T = 2*tot - others)
    int pr0 = 0, pr1 = 0, pr2 = 0, pr3 = 0;
    __device__ __forceinline__ void add(uint32_t vr, uint32_t lr, uint32_t hr, uint32_t vc, uint32_t lc, uint32_t hc)
    {
        const uint32_t both = vr & vc;
        const uint32_t dl = lr ^ lc, dh = hr ^ hc;
        const uint32_t e = both & ~(dl | dh);
        const uint32_t mis = both & ~e;
        tot += __popc(both);
        if (TYPE == DPR_DIST_TAJIMANEI) {
            eq += __popc(e);
            fA += __popc(both & ~hr & ~lr) + __popc(both & ~hc & ~lc);
            fC += __popc(both & ~hr & lr) + __popc(both & ~hc & lc);
            fG += __popc(both & hr & ~lr) + __popc(both & hc & ~lc);
            pr0 += __popc(both & ~lr & ~lc & dh);          // {A,G}
            pr1 += __popc(both & dh & dl & ~(hr ^ lr));    // {A,T}
            pr2 += __popc(both & dh & dl & (hr ^ lr));     // {C,G}
            pr3 += __popc(both & lr & lc & dh);            // {C,T}
        } else {
            p += __popc(mis & ~dl);
            q += __popc(mis & dl);
            if (TYPE == DPR_DIST_TAMURA) {
                gc1 += __popc(mis & (hr ^ lr));
                gc2 += __popc(mis & (hc ^ lc));
            }
        }
    }
    __device__ __forceinline__ double value(int) const
    {
        if (TYPE == DPR_DIST_TAJIMANEI) {
            const int frac[4] = { fA, fC, fG, 2 * tot - fA - fC - fG };
            double fr[4];
            for (int i = 0; i < 4; ++i) fr[i] = double(frac[i]) / tot / 2.0;
            double h = 0;
            h += 0.5 * pr0 * fr[0] * fr[2];
            h += 0.5 * pr1 * fr[0] * fr[3];
            h += 0.5 * pr2 * fr[1] * fr[2];
            h += 0.5 * pr3 * fr[1] * fr[3];
            const double D = double(tot - eq) / tot;
            const double b = 0.5 * (1.0 - fr[0] * fr[0] - fr[2] * fr[2] + D * D / h);
            return -b * log(1.0 - D / b);
        }
        const double pp = double(p) / tot, qq = double(q) / tot;
        if (TYPE == DPR_DIST_K2P) return -0.5 * log((1 - 2 * pp - qq) * sqrt(1 - 2 * qq));
        if (TYPE == DPR_DIST_JINNEI) return 0.5 * (1.0 / (1 - 2 * pp - qq) + 0.5 / (1 - qq * 2) - 1.5);
        const double c = double(gc1) / tot + double(gc2) / tot - 2 * double(gc1) * double(gc2) / tot / tot;
        return -c * log(1 - pp / c - qq) - 0.5 * (1 - c) * log(1 - 2 * qq);
    }
    __device__ __forceinline__ double value(int dist_type, const double*, int) const { return value(dist_type); }
};
// types 1 and 2 share one instantiation (TYPE = DPR_DIST_JC), the formula is picked at run time
template <>
struct PairCounts<DPR_DIST_JC> {
    int binv = 0, mism = 0, sites = 0;   // sites = 32 x words fed (padding words count as invalid on both sides)
    // row side (X, LO, HI), column side (X, LX, HI)
    __device__ __forceinline__ void add(uint32_t xr, uint32_t lr, uint32_t hr, uint32_t xc, uint32_t lxc, uint32_t hc)
    {
        binv += __popc(xr & xc);
        uint32_t m;   // v_or3_b32: the compiler emits two v_or_b32 here
        asm("v_or3_b32 %0, %1, %2, %3" : "=v"(m) : "v"(lr ^ lxc), "v"(hr ^ hc), "v"(xr ^ xc));
        mism += __popc(m);
    }
    // a word in which neither sequence has a not-a-base position (X = 0 on both sides, so LX = LO)
    __device__ __forceinline__ void add_fast(uint32_t lr, uint32_t hr, uint32_t lc, uint32_t hc) { mism += __popc((lr ^ lc) | (hr ^ hc)); }
    // a word in which only ONE side has such positions (x: that side's X word; the other side's is 0, so X_r & X_c = 0 and X_r ^ X_c = x):
    // the same four operations as the clean word, with the three-input OR
    __device__ __forceinline__ void add_one_sided(uint32_t x, uint32_t lr, uint32_t hr, uint32_t lxc, uint32_t hc)
    {
        uint32_t m;
        asm("v_or3_b32 %0, %1, %2, %3" : "=v"(m) : "v"(lr ^ lxc), "v"(hr ^ hc), "v"(x));
        mism += __popc(m);
    }
    __device__ __forceinline__ double value(int dist_type) const { return msa_epilogue(sites - binv, sites - mism, dist_type); }
    // tab_ld > 0: the full (useful, match) table of a short alignment; tab_ld < 0: the band useful >= L - kMsaBand of a long one
    // (L = -tab_ld - 1, rows of L + 1 entries indexed by L - useful)
    __device__ __forceinline__ double value(int dist_type, const double* tab, int tab_ld) const
    {
        if (!tab) return value(dist_type);
        if (tab_ld > 0) return tab[(int64_t)(sites - binv) * tab_ld + (sites - mism)];
        const int L = -tab_ld - 1, g = L - (sites - binv);
        return (g >= 0 && g <= kMsaBand) ? tab[(int64_t)g * (L + 1) + (sites - mism)] : value(dist_type);
    }
};
// long alignments: the band useful = L - g, g = 0 .. kMsaBand, of that table: tab[type][g][match]
__global__ __launch_bounds__(kThreads) void msa_jc_band_kernel(int L, double* __restrict__ tab)
{
    const int64_t ld = (int64_t)L + 1, i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= ld * (kMsaBand + 1)) return;
    const int g = (int)(i / ld), match = (int)(i % ld);
    tab[i] = msa_epilogue(L - g, match, DPR_DIST_UNCORRECTED);
    tab[ld * (kMsaBand + 1) + i] = msa_epilogue(L - g, match, DPR_DIST_JC);
}
// per sequence: bit j = its 16-word stage j holds a not-a-base position (a gap, an N, the positions >= L of the last word, padding
// words); stages from 63 on share bit 63.  One thread per sequence, once per upload.
__global__ __launch_bounds__(kThreads) void msa_xstage_kernel(const uint32_t* __restrict__ planes, int64_t n, int64_t W32, int64_t L,
                                                              unsigned long long* __restrict__ xstage)
{
    const int64_t s = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (s >= n) return;
    const int64_t nst = (W32 + kKC - 1) / kKC;
    unsigned long long bits = ((W32 % kKC) != 0 || (L % 32) != 0) ? (1ull << (nst - 1 < 63 ? nst - 1 : 63)) : 0ull;
    for (int64_t k = 0; k < W32; ++k)
        if (planes[(0 * n + s) * W32 + k] != 0u) { const int64_t st = k / kKC; bits |= 1ull << (st < 63 ? st : 63); }
    xstage[s] = bits;
}
// (useful, match) -> distance for both types 1 and 2
__global__ __launch_bounds__(kThreads) void msa_jc_table_kernel(int L, double* __restrict__ tab)
{
    const int64_t ld = (int64_t)L + 1, i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= ld * ld) return;
    const int useful = (int)(i / ld), match = (int)(i % ld);
    tab[i] = msa_epilogue(useful, match, DPR_DIST_UNCORRECTED);
    tab[ld * ld + i] = msa_epilogue(useful, match, DPR_DIST_JC);
}

template <int TYPE> struct TileOf { static constexpr int SUB = 2; };              // 32 x 32 pairs, 2 x 2 per thread
template <> struct TileOf<DPR_DIST_JC> { static constexpr int SUB = 4; };         // 64 x 64 pairs, 4 x 4 per thread

// What a block computes: up to PT row sequences x PT column sequences (ids in LDS, -1 = none) and
// where the PT x PT tile of distances goes.
constexpr int64_t kNoDiag = (int64_t)1 << 40;
struct MsaSparseX { const unsigned long long* stage; };
struct TileOut {
    double* out;        // element (r, c) of the tile -> out[r * ld + c]            (row-major target)
    int64_t ld;
    int nr, nc;         // valid rows / columns of the tile
    int lower_base;     // >= 0: keep only c_pos < lower_base + r_pos (tile-local positions + tile origins below)
    int r_org, c_org;   // positions of the tile origin inside its job (for lower_base)
    double* mir;        // != nullptr: also element (r, c) -> mir[c * mir_ld + r]  (mirror / transposed target)
    int64_t mir_ld;
    bool skip_main;     // only the transposed target is written
    int64_t diag;       // element (r, c) with r + diag == c is a tip against itself -> 0 (kNoDiag: none)
    const double* tab;  // types 1-2: distance by (useful, match): tab_ld > 0 the full table of a short alignment (row stride tab_ld),
    int tab_ld;         // tab_ld < 0 the band useful >= L - kMsaBand of a long one (L = -tab_ld - 1); nullptr: computed
    const unsigned long long* xstage;   // types 1-2: per sequence, the stages that hold a not-a-base position (nullptr: every stage of every sequence)
};

// Block of 256 threads = 16 x 16; thread (ty,tx) owns rows ty*SUB.., cols tx*SUB..
// LDS: [side][plane][k][PT (+4) sequences] so that SUB consecutive sequences are one 16/8-byte read;
// rows padded by 4 words: the staging writes (consecutive lanes = consecutive k) then hit 8 banks two
// ways instead of one bank sixteen ways.  The same memory is reused for the PT x PT tile of
// distances, which is then written with full-row coalescing in both orientations.
template <int TYPE>
__device__ __forceinline__ void msa_tile(const uint32_t* __restrict__ planes, int64_t n, int64_t W32, int dist_type,
                                         const int32_t* s_rid, const int32_t* s_cid, const TileOut& o, char* smem)
{
    constexpr int SUB = TileOf<TYPE>::SUB, PT = 16 * SUB, LDP = PT + 4;
    typedef uint32_t (*Stage)[kKC][LDP];
    Stage sA = reinterpret_cast<Stage>(smem);
    Stage sB = reinterpret_cast<Stage>(smem + sizeof(uint32_t) * 3 * kKC * LDP);
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    PairCounts<TYPE> acc[SUB][SUB];
    if constexpr (TYPE == DPR_DIST_JC) {
        // ---- types 1-2 (round 6): the not-a-base bookkeeping only where there is such a position.
        // Tile level: a 16-word stage in which NO sequence of the tile has one stages two planes per side and runs the
        // four-operation body throughout.  Word level: in the other stages every wavefront runs the seven-operation body only for
        // the words in which one of ITS 16 rows or one of the 64 columns has one (s_xm: bit masks the staging threads fill) --
        // with a few such positions per sequence hardly any stage is clean for all 128 sequences, but most words are for 80.
        __shared__ unsigned long long s_slow;
        __shared__ unsigned int s_xm[2][8];      // [stage parity][wavefront 0..3 = its rows | 4 = the columns]
        const bool track = o.xstage != nullptr;  // (nullptr: alignments of a single stage -- its last word is not-a-base for everybody -- and the A/B switch)
        unsigned long long slow = ~0ull;
        if (track) {
            if (tid == 0) s_slow = 0ull;
            if (tid < 16) s_xm[tid >> 3][tid & 7] = 0u;
            __syncthreads();
            if (tid < 2 * PT) {
                const int id = tid < PT ? s_rid[tid] : s_cid[tid - PT];
                const unsigned long long f = id >= 0 ? o.xstage[id] : 0ull;      // (a missing sequence's pairs are never written)
                if (f) atomicOr(&s_slow, f);
            }
            __syncthreads();
            slow = s_slow;
        }
        // staging: thread -> (sequence sq, word quad kq) of every plane of both sides; one 16-byte load per plane
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
        const int sq = tid >> 2, kq = tid & 3, wave = tid >> 6;
        const int64_t ga = s_rid[sq], gb = s_cid[sq];
        int64_t st = 0;
        for (int64_t k0 = 0; k0 < W32; k0 += kKC, ++st) {
            const bool fast = !((slow >> (st < 63 ? st : 63)) & 1ull);
            const int set = (int)(st & 1);
            const int64_t k = k0 + 4 * kq;
            const bool whole = k + 3 < W32;
#pragma unroll
            for (int p = fast ? 1 : 0; p < 3; ++p) {
                const int pb = p == 1 ? 3 : p;                 // rows (X, LO, HI), columns (X, LX, HI); LX = LO where X = 0
                u32x4 va, vb;
                const uint32_t pad_a = p == 0 ? ~0u : 0u, pad_b = (p == 0 || pb == 3) ? ~0u : 0u;      // padding = not a base
                va = (u32x4)(fast ? 0u : pad_a); vb = (u32x4)(fast ? 0u : pad_b);
                if (whole) {
                    if (ga >= 0) va = *reinterpret_cast<const u32x4*>(planes + ((int64_t)p * n + ga) * W32 + k);
                    if (gb >= 0) vb = *reinterpret_cast<const u32x4*>(planes + ((int64_t)pb * n + gb) * W32 + k);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (k + j < W32 && ga >= 0) va[j] = planes[((int64_t)p * n + ga) * W32 + k + j];
                        if (k + j < W32 && gb >= 0) vb[j] = planes[((int64_t)pb * n + gb) * W32 + k + j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { sA[p][4 * kq + j][sq] = va[j]; sB[p][4 * kq + j][sq] = vb[j]; }
                if (p == 0 && track) {      // which words of this stage hold a not-a-base position: per wavefront's rows, and for the columns
                    unsigned int ma = 0, mb = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ma |= (va[j] ? 1u : 0u) << (4 * kq + j); mb |= (vb[j] ? 1u : 0u) << (4 * kq + j); }
                    if (ma) atomicOr(&s_xm[set][sq >> 4], ma);
                    if (mb) atomicOr(&s_xm[set][4], mb);
                }
            }
            __syncthreads();
            if (track && tid < 8) s_xm[set ^ 1][tid] = 0u;     // (the other set: last read before the barrier that ended the previous stage)
            // words of this stage in which a row of this wavefront / a column of the tile has a not-a-base position
            const unsigned int xm_r = fast ? 0u : !track ? 0xffffu : (unsigned int)__builtin_amdgcn_readfirstlane((int)s_xm[set][wave]);
            const unsigned int xm_c = fast ? 0u : !track ? 0xffffu : (unsigned int)__builtin_amdgcn_readfirstlane((int)s_xm[set][4]);
            const unsigned int xm = xm_r & xm_c;            // both sides: the seven-operation body
            auto body_fast = [&](int kk) {
                const uint4 aL = *reinterpret_cast<const uint4*>(&sA[1][kk][ty * 4]);
                const uint4 aH = *reinterpret_cast<const uint4*>(&sA[2][kk][ty * 4]);
                const uint4 bL = *reinterpret_cast<const uint4*>(&sB[1][kk][tx * 4]);
                const uint4 bH = *reinterpret_cast<const uint4*>(&sB[2][kk][tx * 4]);
                const uint32_t al[4] = { aL.x, aL.y, aL.z, aL.w }, ah[4] = { aH.x, aH.y, aH.z, aH.w };
                const uint32_t bl[4] = { bL.x, bL.y, bL.z, bL.w }, bh[4] = { bH.x, bH.y, bH.z, bH.w };
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c].add_fast(al[r], ah[r], bl[c], bh[c]);
            };
            auto body_slow = [&](int kk) {
                const uint4 aV = *reinterpret_cast<const uint4*>(&sA[0][kk][ty * 4]);
                const uint4 aL = *reinterpret_cast<const uint4*>(&sA[1][kk][ty * 4]);
                const uint4 aH = *reinterpret_cast<const uint4*>(&sA[2][kk][ty * 4]);
                const uint4 bV = *reinterpret_cast<const uint4*>(&sB[0][kk][tx * 4]);
                const uint4 bL = *reinterpret_cast<const uint4*>(&sB[1][kk][tx * 4]);
                const uint4 bH = *reinterpret_cast<const uint4*>(&sB[2][kk][tx * 4]);
                const uint32_t av[4] = { aV.x, aV.y, aV.z, aV.w }, al[4] = { aL.x, aL.y, aL.z, aL.w }, ah[4] = { aH.x, aH.y, aH.z, aH.w };
                const uint32_t bv[4] = { bV.x, bV.y, bV.z, bV.w }, bl[4] = { bL.x, bL.y, bL.z, bL.w }, bh[4] = { bH.x, bH.y, bH.z, bH.w };
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c].add(av[r], al[r], ah[r], bv[c], bl[c], bh[c]);
            };
            // words with such positions on ONE side only: X_r & X_c = 0 (nothing to count as invalid on both sides) and X_r ^ X_c is
            // that side's word -- four operations, like the clean word
            auto body_rows = [&](int kk) {
                const uint4 aV = *reinterpret_cast<const uint4*>(&sA[0][kk][ty * 4]);
                const uint4 aL = *reinterpret_cast<const uint4*>(&sA[1][kk][ty * 4]);
                const uint4 aH = *reinterpret_cast<const uint4*>(&sA[2][kk][ty * 4]);
                const uint4 bL = *reinterpret_cast<const uint4*>(&sB[1][kk][tx * 4]);
                const uint4 bH = *reinterpret_cast<const uint4*>(&sB[2][kk][tx * 4]);
                const uint32_t av[4] = { aV.x, aV.y, aV.z, aV.w }, al[4] = { aL.x, aL.y, aL.z, aL.w }, ah[4] = { aH.x, aH.y, aH.z, aH.w };
                const uint32_t bl[4] = { bL.x, bL.y, bL.z, bL.w }, bh[4] = { bH.x, bH.y, bH.z, bH.w };
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c].add_one_sided(av[r], al[r], ah[r], bl[c], bh[c]);
            };
            auto body_cols = [&](int kk) {
                const uint4 aL = *reinterpret_cast<const uint4*>(&sA[1][kk][ty * 4]);
                const uint4 aH = *reinterpret_cast<const uint4*>(&sA[2][kk][ty * 4]);
                const uint4 bV = *reinterpret_cast<const uint4*>(&sB[0][kk][tx * 4]);
                const uint4 bL = *reinterpret_cast<const uint4*>(&sB[1][kk][tx * 4]);
                const uint4 bH = *reinterpret_cast<const uint4*>(&sB[2][kk][tx * 4]);
                const uint32_t al[4] = { aL.x, aL.y, aL.z, aL.w }, ah[4] = { aH.x, aH.y, aH.z, aH.w };
                const uint32_t bv[4] = { bV.x, bV.y, bV.z, bV.w }, bl[4] = { bL.x, bL.y, bL.z, bL.w }, bh[4] = { bH.x, bH.y, bH.z, bH.w };
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c].add_one_sided(bv[c], al[r], ah[r], bl[c], bh[c]);
            };
            // (one loop over the words of each kind, not a branch per word: a branch makes the sixteen pairs' counters meet
            //  at a join after every word -- register copies that cost what the short body saves, measured)
            if ((xm_r | xm_c) == 0u) {
#pragma unroll MSA_UNROLL
                for (int kk = 0; kk < kKC; ++kk) body_fast(kk);
            } else if (xm == 0xffffu) {
#pragma unroll MSA_UNROLL
                for (int kk = 0; kk < kKC; ++kk) body_slow(kk);
            } else {
                unsigned int mf = ~(xm_r | xm_c) & 0xffffu, mr = xm_r & ~xm_c & 0xffffu, mc = xm_c & ~xm_r & 0xffffu, ms = xm & 0xffffu;
                while (mf) { const int kk = __builtin_ctz(mf); mf &= mf - 1u; body_fast(kk); }
                while (mc) { const int kk = __builtin_ctz(mc); mc &= mc - 1u; body_cols(kk); }
                while (mr) { const int kk = __builtin_ctz(mr); mr &= mr - 1u; body_rows(kk); }
                while (ms) { const int kk = __builtin_ctz(ms); ms &= ms - 1u; body_slow(kk); }
            }
            __syncthreads();
        }
    } else
    for (int64_t k0 = 0; k0 < W32; k0 += kKC) {
        // stage: 2 sides x 3 planes x PT seqs x 16 words; consecutive lanes read consecutive words of
        // one sequence (64-byte runs)
        for (int e = tid; e < 3 * PT * kKC; e += kThreads) {
            const int p = e / (PT * kKC), rem = e % (PT * kKC);
            const int sq = rem / kKC, kk = rem % kKC;
            const int64_t k = k0 + kk;
            const int64_t ga = s_rid[sq], gb = s_cid[sq];
            // types 1-2: rows (X, LO, HI), columns (X, LX, HI); padding = not a base.  Types 3-6 work on the
            // valid plane V = ~X
            constexpr bool JC = TYPE == DPR_DIST_JC;
            const int pb = (JC && p == 1) ? 3 : p;
            uint32_t va = p == 0 ? ~0u : 0u, vb = (p == 0 || pb == 3) ? ~0u : 0u;
            if (k < W32) {
                if (ga >= 0) va = planes[((int64_t)p * n + ga) * W32 + k];
                if (gb >= 0) vb = planes[((int64_t)pb * n + gb) * W32 + k];
            }
            if (!JC && p == 0) { va = ~va; vb = ~vb; }
            sA[p][kk][sq] = va;
            sB[p][kk][sq] = vb;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < kKC; ++kk) {
            uint32_t av[SUB], al[SUB], ah[SUB], bv[SUB], bl[SUB], bh[SUB];
            if (SUB == 4) {
                const uint4 aV = *reinterpret_cast<const uint4*>(&sA[0][kk][ty * 4]);
                const uint4 aL = *reinterpret_cast<const uint4*>(&sA[1][kk][ty * 4]);
                const uint4 aH = *reinterpret_cast<const uint4*>(&sA[2][kk][ty * 4]);
                const uint4 bV = *reinterpret_cast<const uint4*>(&sB[0][kk][tx * 4]);
                const uint4 bL = *reinterpret_cast<const uint4*>(&sB[1][kk][tx * 4]);
                const uint4 bH = *reinterpret_cast<const uint4*>(&sB[2][kk][tx * 4]);
                av[0] = aV.x; av[1] = aV.y; av[SUB - 2] = aV.z; av[SUB - 1] = aV.w;
                al[0] = aL.x; al[1] = aL.y; al[SUB - 2] = aL.z; al[SUB - 1] = aL.w;
                ah[0] = aH.x; ah[1] = aH.y; ah[SUB - 2] = aH.z; ah[SUB - 1] = aH.w;
                bv[0] = bV.x; bv[1] = bV.y; bv[SUB - 2] = bV.z; bv[SUB - 1] = bV.w;
                bl[0] = bL.x; bl[1] = bL.y; bl[SUB - 2] = bL.z; bl[SUB - 1] = bL.w;
                bh[0] = bH.x; bh[1] = bH.y; bh[SUB - 2] = bH.z; bh[SUB - 1] = bH.w;
            } else {
                const uint2 aV = *reinterpret_cast<const uint2*>(&sA[0][kk][ty * 2]);
                const uint2 aL = *reinterpret_cast<const uint2*>(&sA[1][kk][ty * 2]);
                const uint2 aH = *reinterpret_cast<const uint2*>(&sA[2][kk][ty * 2]);
                const uint2 bV = *reinterpret_cast<const uint2*>(&sB[0][kk][tx * 2]);
                const uint2 bL = *reinterpret_cast<const uint2*>(&sB[1][kk][tx * 2]);
                const uint2 bH = *reinterpret_cast<const uint2*>(&sB[2][kk][tx * 2]);
                av[0] = aV.x; av[1] = aV.y; al[0] = aL.x; al[1] = aL.y; ah[0] = aH.x; ah[1] = aH.y;
                bv[0] = bV.x; bv[1] = bV.y; bl[0] = bL.x; bl[1] = bL.y; bh[0] = bH.x; bh[1] = bH.y;
            }
#pragma unroll
            for (int r = 0; r < SUB; ++r)
#pragma unroll
                for (int c = 0; c < SUB; ++c) acc[r][c].add(av[r], al[r], ah[r], bv[c], bl[c], bh[c]);
        }
        __syncthreads();
    }
    if constexpr (TYPE == DPR_DIST_JC) {
        const int sites = 32 * kKC * (int)((W32 + kKC - 1) / kKC);   // every word fed, padding included
#pragma unroll
        for (int r = 0; r < SUB; ++r)
#pragma unroll
            for (int c = 0; c < SUB; ++c) acc[r][c].sites = sites;
    }
    // distances into the LDS tile (row stride PT+1 doubles), then coalesced rows in both orientations
    double* T = reinterpret_cast<double*>(smem);
#pragma unroll
    for (int r = 0; r < SUB; ++r)
#pragma unroll
        for (int c = 0; c < SUB; ++c) {
            const int rr = ty * SUB + r, cc = tx * SUB + c;
            double d = 0.0;
            if (rr < o.nr && cc < o.nc && rr + o.diag != cc) d = acc[r][c].value(dist_type, o.tab, o.tab_ld);
            T[rr * (PT + 1) + cc] = d;
        }
    __syncthreads();
    if (!o.skip_main)
        for (int e = tid; e < PT * PT; e += kThreads) {
            const int rr = e / PT, cc = e % PT;
            if (rr < o.nr && cc < o.nc && (o.lower_base < 0 || o.c_org + cc < o.lower_base + o.r_org + rr))
                o.out[(int64_t)rr * o.ld + cc] = T[rr * (PT + 1) + cc];
        }
    if (o.mir)
        for (int e = tid; e < PT * PT; e += kThreads) {
            const int cc = e / PT, rr = e % PT;
            if (rr < o.nr && cc < o.nc) o.mir[(int64_t)cc * o.mir_ld + rr] = T[rr * (PT + 1) + cc];
        }
}

template <int TYPE> constexpr size_t msa_tile_lds()
{
    constexpr int PT = 16 * TileOf<TYPE>::SUB;
    constexpr size_t stage = sizeof(uint32_t) * 2 * 3 * kKC * (PT + 4), tile = sizeof(double) * PT * (PT + 1);
    return stage > tile ? stage : tile;
}

// Matrix front-end: local rows l0.. of (rank, world) (world > 0) or tips row0 + l (world == 0)
// against columns col0 + [0, ncols).  world == 1: only tiles on or below the diagonal are computed and
// mirrored (counts are symmetric, src/MSA.cu:121-122).  transposed: out[(c - col0) * ld + l].
template <int TYPE>
__global__ __launch_bounds__(kThreads, MSA_WAVES) void msa_dist_kernel(const uint32_t* __restrict__ planes, int64_t n, int64_t W32,
                                                            int dist_type, double* __restrict__ D, int64_t ld,
                                                            int64_t rows_local, int rank, int world, int64_t row0,
                                                            int64_t col0, int64_t ncols, int transposed, const double* __restrict__ tab, int tab_ld,
                                                            MsaSparseX xs)
{
    constexpr int PT = 16 * TileOf<TYPE>::SUB;
    __shared__ __attribute__((aligned(16))) char smem[msa_tile_lds<TYPE>()];
    __shared__ int32_t s_rid[PT], s_cid[PT];
    const int64_t l0 = (int64_t)blockIdx.y * PT;   // PT divides the ownership block: one owner per row tile
    const int64_t c0 = col0 + (int64_t)blockIdx.x * PT;
    const int64_t g0 = world > 0 ? shard_global_row(l0, rank, world) : row0 + l0;
    const bool mirror = (world == 1);
    if (mirror && c0 > g0 + PT - 1) return;
    if (threadIdx.x < PT) {
        const int64_t ga = g0 + threadIdx.x, gb = c0 + threadIdx.x;
        s_rid[threadIdx.x] = (ga < n && l0 + threadIdx.x < rows_local) ? (int32_t)ga : -1;
        s_cid[threadIdx.x] = (gb < n && gb < col0 + ncols) ? (int32_t)gb : -1;
    }
    __syncthreads();
    TileOut o;
    const int64_t nr = rows_local - l0 < n - g0 ? rows_local - l0 : n - g0;
    const int64_t ncl = col0 + ncols < n ? col0 + ncols : n;
    o.nr = (int)(nr < PT ? nr : PT);
    o.nc = (int)(ncl - c0 < PT ? ncl - c0 : PT);
    o.lower_base = -1; o.r_org = 0; o.c_org = 0;
    o.diag = g0 - c0;
    o.tab = tab; o.tab_ld = tab_ld; o.xstage = xs.stage;
    if (transposed) {
        o.skip_main = true; o.out = nullptr; o.ld = 0;
        o.mir = D + (c0 - col0) * ld + l0; o.mir_ld = ld;
    } else {
        o.skip_main = false; o.out = D + l0 * ld + c0; o.ld = ld;
        const bool below = mirror && c0 + PT - 1 < g0;
        o.mir = below ? D + c0 * ld + g0 : nullptr; o.mir_ld = ld;
    }
    msa_tile<TYPE>(planes, n, W32, dist_type, s_rid, s_cid, o, smem);
}

// Job front-end (divide-and-conquer cluster distances): job -> (cluster, row tile, column tile);
// rows = the cluster's members, columns = its leaf list (ids, -1 = empty); element (t, u) with
// u < 10 + t goes to out[cl_out + t * cl_ld + u].
template <int TYPE>
__global__ __launch_bounds__(kThreads, MSA_WAVES) void msa_dist_jobs_kernel(const uint32_t* __restrict__ planes, int64_t n,
                                                                 int64_t W32, int dist_type, PairJobs J, const double* __restrict__ tab, int tab_ld,
                                                                 MsaSparseX xs)
{
    constexpr int PT = 16 * TileOf<TYPE>::SUB;
    __shared__ __attribute__((aligned(16))) char smem[msa_tile_lds<TYPE>()];
    __shared__ int32_t s_rid[PT], s_cid[PT];
    const int4 job = J.jobs[blockIdx.x];             // cluster, first row, first column, unused
    const int ci = job.x, t0 = job.y, u0 = job.z;
    const int m = J.cl_m[ci], ncols = m + kDcLeaves;
    if (threadIdx.x < PT) {
        const int t = t0 + threadIdx.x, u = u0 + threadIdx.x;
        s_rid[threadIdx.x] = t < m ? J.members[J.cl_moff[ci] + t] : -1;
        s_cid[threadIdx.x] = u < ncols ? J.cols[J.cl_coff[ci] + u] : -1;
    }
    __syncthreads();
    TileOut o;
    o.nr = m - t0 < PT ? m - t0 : PT;
    o.nc = ncols - u0 < PT ? ncols - u0 : PT;
    o.lower_base = kDcLeaves; o.r_org = t0; o.c_org = u0;
    o.diag = kNoDiag; o.skip_main = false;
    o.tab = tab; o.tab_ld = tab_ld; o.xstage = xs.stage;
    o.out = J.out + J.cl_out[ci] + (int64_t)t0 * J.cl_ld[ci] + u0; o.ld = J.cl_ld[ci];
    o.mir = nullptr; o.mir_ld = 0;
    msa_tile<TYPE>(planes, n, W32, dist_type, s_rid, s_cid, o, smem);
}

static const double* msa_jc_tab(const MsaBuffers& m, int dist_type)
{
    if (!m.jc_tab) return nullptr;
    return m.jc_tab + (dist_type == DPR_DIST_JC ? (m.L + 1) * (m.L <= kMsaTabSites ? m.L + 1 : kMsaBand + 1) : 0);
}
// row stride of the full table (short alignments), or -(L + 1) for the band table of a long one (PairCounts<JC>::value)
static int msa_jc_tab_ld(const MsaBuffers& m) { return m.L <= kMsaTabSites ? (int)m.L + 1 : -((int)m.L + 1); }

template <int TYPE>
static int launch_matrix(dim3 grid, hipStream_t s, const uint32_t* planes, int64_t n, int64_t W32, int dist_type,
                         double* D, int64_t ld, int64_t rows, int rank, int world, int64_t row0, int64_t col0,
                         int64_t ncols, int transposed, const double* tab, int tab_ld, MsaSparseX xs = MsaSparseX{ nullptr })
{
    hipLaunchKernelGGL(msa_dist_kernel<TYPE>, grid, dim3(kThreads), 0, s, planes, n, W32, dist_type, D, ld, rows, rank,
                       world, row0, col0, ncols, transposed, tab, tab_ld, xs);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

static int msa_launch(int dist_type, hipStream_t s, const MsaBuffers& m, double* D, int64_t ld, int64_t rows, int rank,
                      int world, int64_t row0, int64_t col0, int64_t ncols, int transposed)
{
    const int pt = (dist_type == DPR_DIST_UNCORRECTED || dist_type == DPR_DIST_JC) ? 64 : 32;
    dim3 g((unsigned)((ncols + pt - 1) / pt), (unsigned)((rows + pt - 1) / pt));
    switch (dist_type) {
    case DPR_DIST_UNCORRECTED:
    case DPR_DIST_JC:        return launch_matrix<DPR_DIST_JC>(g, s, m.planes, m.n, m.W32, dist_type, D, ld, rows, rank, world, row0, col0, ncols, transposed, msa_jc_tab(m, dist_type), msa_jc_tab_ld(m), MsaSparseX{ m.W32 > kKC ? m.xstage : nullptr });
    case DPR_DIST_TAJIMANEI: return launch_matrix<DPR_DIST_TAJIMANEI>(g, s, m.planes, m.n, m.W32, dist_type, D, ld, rows, rank, world, row0, col0, ncols, transposed, nullptr, 0);
    case DPR_DIST_K2P:       return launch_matrix<DPR_DIST_K2P>(g, s, m.planes, m.n, m.W32, dist_type, D, ld, rows, rank, world, row0, col0, ncols, transposed, nullptr, 0);
    case DPR_DIST_TAMURA:    return launch_matrix<DPR_DIST_TAMURA>(g, s, m.planes, m.n, m.W32, dist_type, D, ld, rows, rank, world, row0, col0, ncols, transposed, nullptr, 0);
    case DPR_DIST_JINNEI:    return launch_matrix<DPR_DIST_JINNEI>(g, s, m.planes, m.n, m.W32, dist_type, D, ld, rows, rank, world, row0, col0, ncols, transposed, nullptr, 0);
    default: set_error("unknown distance type (valid: 1-6)"); return DPR_ERR_ARG;
    }
}

int msa_dist_tile_edge(int dist_type) { return (dist_type == DPR_DIST_UNCORRECTED || dist_type == DPR_DIST_JC) ? 64 : 32; }

int msa_dist_jobs(const MsaBuffers& m, int dist_type, const PairJobs& J, int njobs, hipStream_t s)
{
    if (njobs <= 0) return DPR_OK;
    switch (dist_type) {
    case DPR_DIST_UNCORRECTED:
    case DPR_DIST_JC:        hipLaunchKernelGGL(msa_dist_jobs_kernel<DPR_DIST_JC>, dim3(njobs), dim3(kThreads), 0, s, m.planes, m.n, m.W32, dist_type, J, msa_jc_tab(m, dist_type), msa_jc_tab_ld(m), MsaSparseX{ m.W32 > kKC ? m.xstage : nullptr }); break;
    case DPR_DIST_TAJIMANEI: hipLaunchKernelGGL(msa_dist_jobs_kernel<DPR_DIST_TAJIMANEI>, dim3(njobs), dim3(kThreads), 0, s, m.planes, m.n, m.W32, dist_type, J, (const double*)nullptr, 0, MsaSparseX{ nullptr }); break;
    case DPR_DIST_K2P:       hipLaunchKernelGGL(msa_dist_jobs_kernel<DPR_DIST_K2P>, dim3(njobs), dim3(kThreads), 0, s, m.planes, m.n, m.W32, dist_type, J, (const double*)nullptr, 0, MsaSparseX{ nullptr }); break;
    case DPR_DIST_TAMURA:    hipLaunchKernelGGL(msa_dist_jobs_kernel<DPR_DIST_TAMURA>, dim3(njobs), dim3(kThreads), 0, s, m.planes, m.n, m.W32, dist_type, J, (const double*)nullptr, 0, MsaSparseX{ nullptr }); break;
    case DPR_DIST_JINNEI:    hipLaunchKernelGGL(msa_dist_jobs_kernel<DPR_DIST_JINNEI>, dim3(njobs), dim3(kThreads), 0, s, m.planes, m.n, m.W32, dist_type, J, (const double*)nullptr, 0, MsaSparseX{ nullptr }); break;
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
        const uint32_t av = ~planes[(0 * n + row) * W32 + k], al = planes[(1 * n + row) * W32 + k],
                       ah = planes[(2 * n + row) * W32 + k];
        const uint32_t bv = ~planes[(0 * n + c) * W32 + k], bl = planes[(1 * n + c) * W32 + k],
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
    DPR_HIP(hipMalloc(&m.planes, sizeof(uint32_t) * (size_t)(4 * n * m.W32)));
    const int64_t total = n * m.W32;
    const unsigned grid = (unsigned)((total + kThreads - 1) / kThreads > 8192 ? 8192 : (total + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(msa_planes_kernel, dim3(grid ? grid : 1), dim3(kThreads), 0, s, d_in, n, L, W64,
                       m.W32, m.planes);
    DPR_HIP(hipGetLastError());
    if (L <= kMsaTabSites) {
        const int64_t cells = (L + 1) * (L + 1);
        DPR_HIP(hipMalloc(&m.jc_tab, sizeof(double) * (size_t)(2 * cells)));
        hipLaunchKernelGGL(msa_jc_table_kernel, dim3((unsigned)((cells + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, (int)L, m.jc_tab);
        DPR_HIP(hipGetLastError());
    } else if (L < (1 << 30) && !std::getenv("DPR_MSA_NO_BAND")) {
        const int64_t cells = (L + 1) * (kMsaBand + 1);
        DPR_HIP(hipMalloc(&m.jc_tab, sizeof(double) * (size_t)(2 * cells)));
        hipLaunchKernelGGL(msa_jc_band_kernel, dim3((unsigned)((cells + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, (int)L, m.jc_tab);
        DPR_HIP(hipGetLastError());
    }
    if (!std::getenv("DPR_MSA_NO_FAST")) {      // (the switch exists for the A/B runs of profiles/msa_block_bench.py and for the tests)
        DPR_HIP(hipMalloc(&m.xstage, sizeof(unsigned long long) * (size_t)n));
        hipLaunchKernelGGL(msa_xstage_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, (const uint32_t*)m.planes, n, m.W32, L, m.xstage);
        DPR_HIP(hipGetLastError());
    }
    DPR_HIP(hipStreamSynchronize(s));
    DPR_HIP(hipFree(d_in));
    return DPR_OK;
}

void msa_free(MsaBuffers& m)
{
    if (m.planes) (void)hipFree(m.planes);
    if (m.jc_tab) (void)hipFree(m.jc_tab);
    if (m.xstage) (void)hipFree(m.xstage);
    m = MsaBuffers();
}

int msa_dist_rows(const MsaBuffers& m, NjBuffers& b, int dist_type, hipStream_t s)
{
    if (b.rows_local == 0) return DPR_OK;
    return msa_launch(dist_type, s, m, b.D, b.ld, b.rows_local, b.rank, b.world, 0, 0, m.n, 0);
}

int msa_dist_block_rows(const MsaBuffers& m, int64_t r0, int64_t nr, int rank, int world, int64_t ncols,
                        int dist_type, double* out, int64_t ld, hipStream_t s, bool transposed)
{
    if (nr <= 0 || ncols <= 0) return DPR_OK;
    (void)rank;
    return msa_launch(dist_type, s, m, out, ld, nr, 0, world, r0, 0, ncols, transposed ? 1 : 0);
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
