// Exact PRUNED neighbor joining (single GPU).  Same arithmetic, tie-breaking and merge log as the
// streaming path of nj.hip (and therefore as the reference), but the per-iteration Q-argmin no
// longer streams the whole triangle:
//
//  * the matrix lives in POSITION space: tips are permuted by ascending initial row sum, rows never
//    move afterwards (the reference's "last slot moves into y" becomes a relabel of slot_of_pos /
//    pos_of_slot); every key uses the reference's slot numbers, so ties resolve as in the reference;
//  * the triangle is cut into units of 16 rows x 512 columns, each with four 128-column sub-units (the columns of
//    one wave of the scan kernel) that carry a lazily maintained lower bound umin[unit][w] <= min D over the
//    sub-unit (exact whenever it is scanned; lowered by the prep lane that tests the unit when the new node's
//    row / column crosses it);
//  * for a unit, every candidate obeys  q = fl(fl(D - Ur_a) - Ur_b) >= fl(fl(umin - rmax) - cmax)
//    (and the other association order) because fl(x - y) is monotone in x and -y; units whose bound
//    exceeds a known upper bound of the optimum (the best of the previous iteration's per-block
//    winners re-evaluated with the current row sums) cannot contain the winner NOR A TIE and are
//    skipped.  Sorting by row sum makes Ur homogeneous inside units, which is what makes the bound
//    tight.
//
// Per iteration: njp_scan_kernel (scan the surviving sub-units of the listed units, refresh their bounds) -> njp_post_kernel
// (select + merge + update, indexed by reference slot so that the canonical U[x] summation order is
// unchanged) -> njp_prep_kernel (finish U[x], seed bound, test every unit, list the survivors).
// The kernels take no per-iteration arguments (they read the iteration index from the device
// state), so 32 iterations are captured into one hipGraph and replayed.
#include "nj_dev.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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

// unit ownership of the unit-sharded mode (also exported for the CPU tests of the N > 1 logic)
int njp_unit_owner(int64_t strip, int64_t group, int64_t P, int world)
{
    const int64_t G16 = (P + kUR - 1) / kUR;
    if (strip < 0 || group < 0 || group >= G16 || group < 32 * strip || strip * kTileCols >= P - 1 || world < 1) return -1;
    return (int)((strip * G16 + group) % world);
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

// Latency is what matters in these three kernels (a few hundred KB of data per iteration): every
// kernel issues all of its global loads up front, in as few dependent hops as possible.
//
// prep = finish U[px] + seed bound + unit tests, one lane per valid unit.  Every block derives what
// it needs -- the new node's row sum, the bound, the maxima of Ur over its units' rows and strips --
// directly from Ur, so no cross-block hand-off is needed.  Runs after post(it-1); `it` = st->it is
// stable while it runs.  It publishes st->itb = it, the iteration index the following scan/post
// kernels read (they must not read st->it, which the post kernel advances while it runs).
// For it >= it_limit only U[px] is materialised (no list).
__global__ __launch_bounds__(kThreads) void njp_prep_kernel(const double* __restrict__ D, int64_t ld,
                                                            NjState* __restrict__ st, double* __restrict__ U_w,
                                                            double* __restrict__ Ur_w, const double* __restrict__ Ur,
                                                            const double* __restrict__ xpart,
                                                            const NjRecord* __restrict__ partials,
                                                            unsigned long long* umin,
                                                            int64_t P, const int32_t* __restrict__ blk_cb,
                                                            const int32_t* __restrict__ blk_g0, int scan_grid,
                                                            int32_t* __restrict__ list, int sh_rank, int sh_world,
                                                            unsigned long long* __restrict__ cnt_rank, int nrec_fixed,
                                                            unsigned long long* __restrict__ clk)
{
    __shared__ double s[kThreads];
    __shared__ double sseed[kThreads / 64];
    __shared__ double scm[kThreads / 64];
    __shared__ double snew[kThreads / 64];
    const int tid = threadIdx.x;
    // optional phase clocks (DPR_NJ_ITERSTATS; profiles/nj_clocks.py): thread 0 of block 0 and of the middle block
    const bool clocked = clk != nullptr && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2);
    unsigned long long ck[7] = { 0, 0, 0, 0, 0, 0, 0 };
    if (clocked) ck[0] = wall_clock64();
    // hop 1: the state line, this block's (strip, first group) -- blocks never span strips -- and, speculatively,
    // this thread's seed record (its index depends on the launch arguments only)
    const int64_t nrec_arg = nrec_fixed >= 0 ? (int64_t)(nrec_fixed < scan_grid ? nrec_fixed : scan_grid) : -1;
    const int64_t stride = nrec_arg >= 2 * kThreads ? nrec_arg / kThreads : 1;   // gathered records of several ranks: every (nrec/256)-th
    NjRecord cand = partials[(int64_t)tid * stride];
    const int cb = blk_cb[blockIdx.x];
    const int64_t g = (int64_t)blk_g0[blockIdx.x] + tid;
    const int64_t it = st->it, limit = st->it_limit, N = st->N;
    const int64_t px = it > 0 ? (int64_t)st->pad : -1;
    // unit-sharded mode (sh_world > 1): this launch tests only the units it owns and appends to its own list /
    // counter; the seed records are the gathered records of ALL ranks, nrec_fixed of them, each written
    unsigned long long* cntp = cnt_rank ? cnt_rank : st->cnt_list;
    const unsigned long long nlist_prev = it > 0 ? (nrec_fixed >= 0 ? (unsigned long long)nrec_fixed : cntp[(it - 1) & 1]) : 0ull;
    if (st->status != 0) return;
    const bool beyond = it >= limit;
    if (blockIdx.x == 0 && tid == 0) st->itb = it;            // never read by this kernel
    if (beyond && (blockIdx.x != 0 || it == 0)) return;       // only block 0 materialises U[px]
    const int64_t n = N - it;
    const double NINF = -__builtin_inf(), PINF = __builtin_inf();
    if (clocked) { __builtin_amdgcn_s_waitcnt(0); ck[1] = wall_clock64(); }

    // hop 2: every other load of this block
    const int64_t nchunk = it > 0 ? (n + 1 + kThreads - 1) / kThreads : 0;
    double acc = 0.0;
    for (int64_t c = tid; c < nchunk; c += kThreads) acc += xpart[c];       // chunk sums of U[px]
    const int64_t nrec = (int64_t)(nlist_prev < (unsigned long long)scan_grid ? nlist_prev : (unsigned long long)scan_grid);
    if (beyond || (int64_t)tid * stride >= nrec) cand.key = ~0ull;   // stale or absent record
    const int64_t G16 = (P + kUR - 1) / kUR;
    const bool have = !beyond && g < G16;
    // The new node's row (written by the post kernel) may lower the bound of the units it crosses: the units
    // (strip of px, groups behind px) through their 16 rows, the units (strips before px, group of px) through
    // their 512 columns.  The lane that tests a unit folds that minimum into umin itself (no atomics in post).
    const bool px_strip = px >= 0 && px / kTileCols == cb;                       // block-uniform
    const int64_t gx = px >= 0 ? px / kUR : -1;
    const bool gx_here = px >= 0 && gx >= (int64_t)blk_g0[blockIdx.x] && gx < (int64_t)blk_g0[blockIdx.x] + kThreads;   // block-uniform
    const double* __restrict__ rowx = D + (px >= 0 ? px : 0) * ld;
    // Every unit carries FOUR bounds, one per 128-column sub-strip (the columns of one wave of the scan kernel):
    // the test keeps a unit if any sub-unit survives and hands the scan a 4-bit mask, so a wave whose sub-unit
    // cannot hold the winner neither loads nor evaluates its 16 x 128 block.
    const int wpx = px >= 0 ? (int)((px % kTileCols) / (kTileCols / 4)) : -1;    // sub-strip of the new node's column
    double u4[4] = { PINF, PINF, PINF, PINF };
    double rmax = NINF; bool px_in_group = false;
    double newminA = PINF;                                  // new row x this group's rows (sub-strip wpx of px's strip)
    unsigned long long* up4 = umin + ((int64_t)cb * G16 + (have ? g : 0)) * 4;
    // All loads of this round trip are issued before the first use, without data-dependent branches in between
    // (measured with the phase clocks: as a loop of "load, test, use" the sixteen row sums of a lane arrived one
    // after the other, 4.3 us; the vectors are padded with NaN behind P, so every address is valid).
    v2d urow[kUR / 2], drow[kUR / 2];
    ulonglong2 um0 = make_ulonglong2(0ull, 0ull), um1 = um0;
    const int64_t a0 = (have ? g : 0) * kUR;                       // 128-byte aligned: eight 16-byte loads
    const int64_t pc0 = (int64_t)cb * kTileCols + 2 * tid;         // this thread's two strip columns (< P + 512)
    if (have) { um0 = *reinterpret_cast<const ulonglong2*>(up4); um1 = *reinterpret_cast<const ulonglong2*>(up4 + 2); }
#pragma unroll
    for (int r = 0; r < kUR / 2; ++r) urow[r] = *reinterpret_cast<const v2d*>(Ur + a0 + 2 * r);
    const v2d ucol = *reinterpret_cast<const v2d*>(Ur + pc0);
    v2d dcol; dcol.x = PINF; dcol.y = PINF;
    if (px_strip) {                                                // block-uniform
#pragma unroll
        for (int r = 0; r < kUR / 2; ++r) drow[r] = *reinterpret_cast<const v2d*>(rowx + a0 + 2 * r);
    } else {
#pragma unroll
        for (int r = 0; r < kUR / 2; ++r) { drow[r].x = PINF; drow[r].y = PINF; }
    }
    if (gx_here) dcol = *reinterpret_cast<const v2d*>(rowx + pc0);  // block-uniform; row px is readable up to its padded end
    // candidate gathers (same round trip: the records were loaded up front)
    double cd = 0.0, cua = __builtin_nan(""), cub = __builtin_nan("");
    if (cand.key != ~0ull) {
        const int64_t pi = (int64_t)(cand.pad & 0xffffffffull), pj = (int64_t)(cand.pad >> 32);
        const int64_t pa = pi > pj ? pi : pj, pb = pi > pj ? pj : pi;
        // the record carries D[pa][pb]; the entry is unchanged since the scan (neither end is the new node, and a
        // dead end gives a NaN row sum) -- no gather into the 7 GB matrix (a cold TLB walk per launch)
        if (pa < P && pb < pa && pa != px && pb != px) { cd = cand.d; cua = Ur[pa]; cub = Ur[pb]; }
    }
    if (have) {
        u4[0] = dec_f64(um0.x); u4[1] = dec_f64(um0.y); u4[2] = dec_f64(um1.x); u4[3] = dec_f64(um1.y);
#pragma unroll
        for (int r = 0; r < kUR; ++r) {
            const int64_t p = a0 + r;
            const double v = (r & 1) ? urow[r >> 1].y : urow[r >> 1].x;
            const double dv = (r & 1) ? drow[r >> 1].y : drow[r >> 1].x;
            px_in_group |= p == px;
            const bool live = (v == v) & (p != px);                // dead positions and the padding behind P carry NaN
            rmax = live ? fmax(rmax, v) : rmax;
            newminA = (live & (p > px)) ? fmin(newminA, dv) : newminA;
        }
    }
    // column maximum of each sub-strip (wave w reads columns 128w .. 128w+127 of the strip), px excluded for now;
    // minimum of the new row over the sub-strip's live columns
    double cm_part = NINF, colmin = PINF;
    if (!beyond) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t p = pc0 + k;
            const double v = k ? ucol.y : ucol.x;
            const double dv = k ? dcol.y : dcol.x;
            const bool live = (v == v) & (p != px);
            cm_part = live ? fmax(cm_part, v) : cm_part;
            colmin = (live & (p < px)) ? fmin(colmin, dv) : colmin;
        }
    }
    double qc = PINF;
    if (cua == cua && cub == cub) qc = fmin((cd - cua) - cub, (cd - cub) - cua);
    if (clocked) { __builtin_amdgcn_s_waitcnt(0); ck[2] = wall_clock64(); }
    // ---- reductions: the wave results go to LDS before the tree, whose barriers publish them as well
    if (!beyond) {
        qc = wave_fmin(qc);
        cm_part = wave_fmax(cm_part);
        if (gx_here) colmin = wave_fmin(colmin);
        if ((tid & 63) == 0) { sseed[tid >> 6] = qc; scm[tid >> 6] = cm_part; snew[tid >> 6] = colmin; }
    }
    double urx = 0.0;
    if (it > 0) {
        double ux = block_tree256_lane0(acc, s);
        if (tid == 0) s[0] = ux;
        __syncthreads();
        ux = s[0];
        urx = ux / (double)(n - 2);
        if (blockIdx.x == 0 && tid == 0) { U_w[px] = ux; Ur_w[px] = urx; }
    } else {
        __syncthreads();
    }
    if (beyond) return;
    if (clocked) ck[3] = wall_clock64();
    const double bound = fmin(fmin(sseed[0], sseed[1]), fmin(sseed[2], sseed[3]));
    // the new node's Ur joins the maxima of its group and of its sub-strip
    if (px_in_group) rmax = fmax(rmax, urx);
    const bool mine = have && (sh_world <= 1 || (int)(((int64_t)cb * G16 + g) % sh_world) == sh_rank);
    int submask = 0;
    if (mine) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            double cmw = scm[w];
            if (px_strip && w == wpx) cmw = fmax(cmw, urx);
            double nm = (gx_here && g == gx) ? snew[w] : PINF;            // the unit (this strip, group of px)
            if (px_strip && w == wpx) nm = fmin(nm, newminA);
            if (nm < u4[w]) {                          // persist the lowered bound (this lane is the unit's only writer here)
                u4[w] = nm;
                up4[w] = enc_f64(nm);
            }
            const double lb = fmin((u4[w] - rmax) - cmw, (u4[w] - cmw) - rmax);
            if ((cmw > NINF) && (lb <= bound)) submask |= 1 << w;
        }
    }
    const bool keep = mine && (rmax > NINF) && submask != 0;
    if (clocked) ck[4] = wall_clock64();
    const int par = (int)(it & 1);
    const unsigned long long mask = __ballot(keep);
    const int lane = tid & 63;
    unsigned long long base = 0;
    if (lane == 0 && mask) base = atomicAdd(&cntp[par], (unsigned long long)__popcll(mask));
    base = __shfl(base, 0, 64);
    if (clocked) { __builtin_amdgcn_s_waitcnt(0); ck[5] = wall_clock64(); }
    if (keep) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)(((uint32_t)submask << 28) | ((uint32_t)cb << 18) | (uint32_t)g);   // sub-unit mask | strip | group
    if (clocked) {
        __builtin_amdgcn_s_waitcnt(0);
        ck[6] = wall_clock64();
        unsigned long long* o = clk + (blockIdx.x == 0 ? 0 : 8);
        atomicAdd(&o[0], 1ull);
        for (int k = 1; k < 7; ++k) atomicAdd(&o[k], ck[k] - ck[k - 1]);
    }
}

// scan the listed units: block b takes entries b, b+G, ... and always writes partials[b] when it had
// work, so the records of a scan are partials[0 .. min(cnt, grid)).  The lane-level best carries the
// positions of the pair and its distance, so nothing is looked up after the reduction.
__device__ __forceinline__ void best_update4(double& bq, uint64_t& bk, uint64_t& bp, double& bd, double q, uint64_t k,
                                             uint64_t pp, double d)
{
    const bool take = (q < bq) | ((q == bq) & (k < bk));
    bq = take ? q : bq; bk = take ? k : bk; bp = take ? pp : bp; bd = take ? d : bd;
}

__global__ __launch_bounds__(kThreads) void njp_scan_kernel(const double* __restrict__ D, int64_t ld,
                                                            NjState* __restrict__ st,
                                                            const double* __restrict__ Ur,
                                                            const uint64_t* __restrict__ KA,
                                                            const uint64_t* __restrict__ KB,
                                                            unsigned long long* __restrict__ umin,
                                                            int64_t P, const int32_t* __restrict__ list,
                                                            NjRecord* __restrict__ partials,
                                                            unsigned long long* __restrict__ iterstats,
                                                            const unsigned long long* __restrict__ cnt_rank, int write_null)
{
    __shared__ double sq[kThreads / 64], sd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64], sp[kThreads / 64];

    const int tid = threadIdx.x;
    const bool clocked = iterstats != nullptr && tid == 0 && blockIdx.x == 0;     // phase clocks (profiles/nj_clocks.py)
    unsigned long long ck[5] = { 0, 0, 0, 0, 0 };
    if (clocked) ck[0] = wall_clock64();
    // hop 1: state line and (speculatively) this block's first list entry
    const int64_t it = st->itb, limit = st->it_limit;
    const int32_t first = list[blockIdx.x];
    if (it >= limit || st->status != 0) return;
    const int64_t cnt = (int64_t)(cnt_rank ? cnt_rank[it & 1] : st->cnt_list[it & 1]);
    if ((int64_t)blockIdx.x >= cnt) {
        if (write_null && tid == 0) {      // unit-sharded mode: every record of the gathered array is defined
            NjRecord rec; rec.q = 10000.0; rec.key = ~0ull; rec.d = 0.0; rec.pad = 0ull;
            partials[blockIdx.x] = rec;
        }
        return;
    }
    const int64_t G16 = (P + kUR - 1) / kUR;
    if (clocked) { __builtin_amdgcn_s_waitcnt(0); ck[1] = wall_clock64(); }
    // Two passes per unit instead of a (q, key, positions, d) compare-and-select per candidate (64 candidates
    // per lane and unit, ~20 VALU instructions each, were 2 us of this kernel's critical path): pass 1 computes
    // the candidates' q and their minimum over the WAVE; pass 2 -- only when that minimum reaches the wave's
    // best so far -- looks for the candidates equal to it (rare, wave-uniform branch) and keeps the smallest key.
    // bq is wave-uniform; (bk, bp, bd) is this lane's best candidate AT q == bq (bk == ~0: none).
    double bq = 10000.0, bd = 0.0;  // the reference's init value
    uint64_t bk = ~0ull, bp = 0;
    int64_t scanned = 0;

    const int wv = tid >> 6;                         // this wave's sub-strip of every unit
    for (int64_t e = blockIdx.x; e < cnt; e += gridDim.x, ++scanned) {
        const uint32_t code = (uint32_t)__builtin_amdgcn_readfirstlane(e == (int64_t)blockIdx.x ? first : list[e]);
        if (!((code >> (28 + wv)) & 1u)) continue;   // the bound of this wave's sub-unit rules it out (wave-uniform)
        const int cb = (int)((code >> 18) & 1023u);
        const int64_t g_s = (int64_t)(code & 0x3FFFFu);
        const int64_t c0 = (int64_t)cb * kTileCols, a0 = g_s * kUR;
        // Rows a0 .. a0+15 are always read: the matrix has a zeroed group of rows behind position P and the
        // vectors carry NaN row sums there (njp_alloc_epoch), so rows >= P behave like dead rows.
        const int64_t b0 = c0 + 2 * tid, b1 = b0 + 1;
        const v2d ubv = *reinterpret_cast<const v2d*>(Ur + b0);
        const ulonglong2 kav = *reinterpret_cast<const ulonglong2*>(KA + b0), kbv = *reinterpret_cast<const ulonglong2*>(KB + b0);
        const double ub0 = ubv.x, ub1 = ubv.y;
        const uint64_t ka0 = kav.x, ka1 = kav.y, kb0 = kbv.x, kb1 = kbv.y;
        const v2d* basep = reinterpret_cast<const v2d*>(D + a0 * ld + c0) + tid;
        const int64_t ld2 = ld >> 1;
        const bool diag = a0 < c0 + kTileCols;
        const bool live0 = ub0 == ub0, live1 = ub1 == ub1;   // dead columns carry NaN row sums
        v2d v[kUR];
#pragma unroll
        for (int u8 = 0; u8 < kUR; ++u8) v[u8] = __builtin_nontemporal_load(basep + (int64_t)u8 * ld2);
        // the 16 rows' row sums and keys: lane l holds row a0 + (l & 15) (three coalesced loads, no scalar-register
        // pressure); v_readlane hands them out -- the keys only in the rare branch of pass 2
        const double ua_l = Ur[a0 + (tid & 15)];
        const uint64_t kaa_l = KA[a0 + (tid & 15)], kba_l = KB[a0 + (tid & 15)];
        double ua[kUR];
#pragma unroll
        for (int u8 = 0; u8 < kUR; ++u8) ua[u8] = readlane_f64(ua_l, u8);
        if (clocked && ck[2] == 0) { __builtin_amdgcn_s_waitcnt(0); ck[2] = wall_clock64(); }
        if (diag) {   // block-uniform and rare (units on the diagonal): mask the entries with column >= row
            const int ib0 = (int)b0, ia0 = (int)a0;
#pragma unroll
            for (int u8 = 0; u8 < kUR; ++u8) {
                const int a = ia0 + u8;
                v[u8].x = (ib0 < a) ? v[u8].x : __builtin_nan("");
                v[u8].y = (ib0 + 1 < a) ? v[u8].y : __builtin_nan("");
            }
        }
        // pass 1: q of the four ordered candidates of every loaded pair; a dead row or column (NaN row sum) and a
        // masked entry (NaN distance) give a NaN q, which fmin drops and no comparison selects.  m0 / m1: the
        // exact minimum over the live rows of this lane's two columns (dead rows add NaN, which fmin drops).
        double lm = __builtin_inf(), m0 = __builtin_inf(), m1 = __builtin_inf();
        double rowq[kUR];                          // this lane's smallest q per row: pass 2 looks only at rows that reach wm
#pragma unroll
        for (int u8 = 0; u8 < kUR; ++u8) {
            const double d0 = v[u8].x, d1 = v[u8].y;
            const double rn = ua[u8] == ua[u8] ? 0.0 : __builtin_nan("");   // wave-uniform
            m0 = fmin(m0, d0 + rn);
            m1 = fmin(m1, d1 + rn);
            const double q0 = (d0 - ua[u8]) - ub0;   // (i=a,  j=b0)
            const double q1 = (d0 - ub0) - ua[u8];   // (i=b0, j=a)
            const double q2 = (d1 - ua[u8]) - ub1;
            const double q3 = (d1 - ub1) - ua[u8];
            rowq[u8] = fmin(fmin(q0, q1), fmin(q2, q3));
            lm = fmin(lm, rowq[u8]);
        }
        double m = fmin(fmin(live0 ? m0 : __builtin_nan(""), live1 ? m1 : __builtin_nan("")), __builtin_inf());   // +inf: no live pair
        const double wm = wave_fmin(lm);          // +inf when the wave saw no valid candidate
        if (wm <= bq) {                           // wave-uniform; the q are recomputed (same operations, same bits)
            if (wm < bq) { bq = wm; bk = ~0ull; }   // rather than kept: 128 registers less, twice the blocks per CU
#pragma unroll
            for (int u8 = 0; u8 < kUR; ++u8) {
                if (__builtin_amdgcn_ballot_w64(rowq[u8] == wm) != 0ull) {      // rare; the four q are recomputed (same bits)
                    double d0 = v[u8].x, d1 = v[u8].y;
                    asm volatile("" : "+v"(d0), "+v"(d1));    // opaque: keeps the compiler from holding pass 1's 64 q alive
                    const bool e0 = (d0 - ua[u8]) - ub0 == wm, e1 = (d0 - ub0) - ua[u8] == wm;
                    const bool e2 = (d1 - ua[u8]) - ub1 == wm, e3 = (d1 - ub1) - ua[u8] == wm;
                    const uint64_t pa = (uint64_t)(a0 + u8);
                    const uint64_t kaa = readlane_u64(kaa_l, u8), kba = readlane_u64(kba_l, u8);
                    const uint64_t k0 = kaa | kb0, k1 = ka0 | kba, k2 = kaa | kb1, k3 = ka1 | kba;
                    if (e0 & (k0 < bk)) { bk = k0; bp = pa | ((uint64_t)b0 << 32); bd = d0; }
                    if (e1 & (k1 < bk)) { bk = k1; bp = (uint64_t)b0 | (pa << 32); bd = d0; }
                    if (e2 & (k2 < bk)) { bk = k2; bp = pa | ((uint64_t)b1 << 32); bd = d1; }
                    if (e3 & (k3 < bk)) { bk = k3; bp = (uint64_t)b1 | (pa << 32); bd = d1; }
                }
            }
        }
        // exact minimum of this wave's sub-unit -> its bound (no cross-wave step any more)
        m = wave_fmin(m);
        if ((tid & 63) == 0) umin[((int64_t)cb * G16 + g_s) * 4 + wv] = enc_f64(m);
    }

    if (clocked) ck[3] = wall_clock64();
    // wave winner: the smallest key among the lanes' candidates at bq, then that lane's positions and distance
    {
        const uint64_t wk = wave_umin64(bk);
        if (wk != ~0ull) {
            const unsigned long long own = __builtin_amdgcn_ballot_w64(bk == wk);
            const int src = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(own));
            bp = readlane_u64(bp, src);
            bd = readlane_f64(bd, src);
        }
        bk = wk;
    }
    if ((tid & 63) == 0) { sq[tid >> 6] = bq; sk[tid >> 6] = bk; sp[tid >> 6] = bp; sd[tid >> 6] = bd; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) best_update4(bq, bk, bp, bd, sq[w], sk[w], sp[w], sd[w]);
        NjRecord rec;
        rec.q = bq; rec.key = bk; rec.d = bd; rec.pad = bp;   // pad = pos_i | pos_j << 32
        partials[blockIdx.x] = rec;
        if (iterstats) { atomicAdd(&iterstats[2 * it], (unsigned long long)scanned); atomicMax(&iterstats[2 * it + 1], (unsigned long long)scanned); }
        if (blockIdx.x == 0) atomicAdd(&st->units_scanned, (unsigned long long)cnt);   // statistics; no load on this block's tail
        if (clocked) {
            __builtin_amdgcn_s_waitcnt(0);
            ck[4] = wall_clock64();
            unsigned long long* o = iterstats + 2 * st->N + 2 + 16;
            atomicAdd(&o[0], 1ull);
            if (ck[2] == 0) ck[2] = ck[1];
            for (int k = 1; k < 5; ++k) atomicAdd(&o[k], ck[k] - ck[k - 1]);
        }
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
                                                            double* __restrict__ xpart,
                                                            const NjRecord* __restrict__ partials, int scan_grid,
                                                            int64_t P,
                                                            int32_t* __restrict__ log_x, int32_t* __restrict__ log_y,
                                                            double* __restrict__ log_bx, double* __restrict__ log_by,
                                                            int nrec_fixed, unsigned long long* __restrict__ cnt_all,
                                                            int cnt_ranks, unsigned long long* __restrict__ clk)
{
    __shared__ double s[kThreads];
    __shared__ double sq[kThreads / 64], sdd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64], spp[kThreads / 64];
    const bool clocked = clk != nullptr && threadIdx.x == 0 && blockIdx.x == 0;   // phase clocks (profiles/nj_clocks.py)
    unsigned long long ck[6] = { 0, 0, 0, 0, 0, 0 };
    if (clocked) ck[0] = wall_clock64();
    // hop 1: state line, this thread's slot -> position, and (speculatively) the scan records
    const int64_t it = st->itb;   // stable: the writer below only advances st->it / st->n
    const int64_t limit = st->it_limit, N = st->N;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    // (none of these addresses depends on a loaded value: the slot arrays are padded past N, the record array holds
    // scan_grid entries; what a load was worth is decided afterwards)
    const int64_t p = (int64_t)pos_of_slot[i];
    NjRecord r0; r0.q = 10000.0; r0.key = ~0ull; r0.d = 0; r0.pad = 0;
    NjRecord mine[4] = { r0, r0, r0, r0 };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = threadIdx.x + k * kThreads;
        if (idx < scan_grid) mine[k] = partials[idx];
    }
    const unsigned long long cnt_raw = nrec_fixed >= 0 ? (unsigned long long)nrec_fixed : st->cnt_list[it & 1];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((unsigned long long)(threadIdx.x + k * kThreads) >= cnt_raw) mine[k] = r0;   // not written by this iteration's scan
    // hop 1b: this position's row sum (its address needs the position only; in flight during the record reduction)
    const double up = (i < N && p >= 0) ? U[p] : 0.0;
    if (st->status != 0 || it >= limit) return;
    const int64_t n = N - it;
    if (n < 3 || (int64_t)blockIdx.x * kThreads >= n) return;
    const int par = (int)(it & 1);

    if (clocked) { __builtin_amdgcn_s_waitcnt(0); ck[1] = wall_clock64(); }
    // select: reduce the records (thrust::min_element, src/neighborJoining.cu:214)
    double bq = 10000.0, d = 0.0; uint64_t bk = ~0ull, bp = 0;
    best_update4(bq, bk, bp, d, mine[0].q, mine[0].key, mine[0].pad, mine[0].d);
    if (cnt_raw > (unsigned long long)kThreads) {      // block-uniform; mostly false: a scan rarely lists more than 256 units
#pragma unroll
        for (int k = 1; k < 4; ++k) best_update4(bq, bk, bp, d, mine[k].q, mine[k].key, mine[k].pad, mine[k].d);
    }
    {
        const int64_t nrec = (int64_t)(cnt_raw < (unsigned long long)scan_grid ? cnt_raw : (unsigned long long)scan_grid);
        for (int64_t idx = threadIdx.x + 4 * kThreads; idx < nrec; idx += kThreads)
            best_update4(bq, bk, bp, d, partials[idx].q, partials[idx].key, partials[idx].pad, partials[idx].d);
    }
    wave_best4(bq, bk, bp, d);
    if ((threadIdx.x & 63) == 0) { sq[threadIdx.x >> 6] = bq; sk[threadIdx.x >> 6] = bk; spp[threadIdx.x >> 6] = bp; sdd[threadIdx.x >> 6] = d; }
    __syncthreads();
    bq = sq[0]; bk = sk[0]; bp = spp[0]; d = sdd[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) best_update4(bq, bk, bp, d, sq[w], sk[w], spp[w], sdd[w]);

    if (clocked) ck[2] = wall_clock64();
    const int64_t last = n - 1;
    if (bk == ~0ull) {
        if (i == last) st->status = 1;
        return;
    }
    const int64_t ki = (int64_t)(bk & 0xFFFFFFull), kj = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t pi = (int64_t)(bp & 0xffffffffull), pj = (int64_t)(bp >> 32);
    const int64_t x = ki < kj ? ki : kj, y = ki < kj ? kj : ki;
    const int64_t px = ki < kj ? pi : pj, py = ki < kj ? pj : pi;
    const int64_t n1 = n - 1;
    const double r1 = (double)(n1 - 2);

    double val = 0.0;
    if (i < n) {
        if (i == last) {
            // single writer of the log and the state (reads U[px], U[py]; nobody rewrites them here)
            const double r = (double)(n - 2);
            double blX = (d + U[px] / r - U[py] / r) * 0.5;
            double blY = d - blX;
            if (blX < 0) { blY += blX; blX = 0; }
            if (blY < 0) { blX += blY; blY = 0; }
            log_x[it] = (int32_t)x; log_y[it] = (int32_t)y; log_bx[it] = blX; log_by[it] = blY;
            st->x = (int32_t)x; st->y = (int32_t)y; st->d = d; st->q = bq;
            st->n = n1; st->it = it + 1; st->pad = (int32_t)px;
            st->cnt_list[1 - par] = 0ull;   // list counter of the NEXT scan (nobody reads it in this launch)
            for (int v = 0; v < cnt_ranks; ++v) cnt_all[2 * v + (1 - par)] = 0ull;
        }
        int64_t new_slot = i;
        if (i != x && i != y) {
            const double dxi = D[px * ld + p], dyi = D[py * ld + p];
            if (clocked) { __builtin_amdgcn_s_waitcnt(0); ck[3] = wall_clock64(); }
            val = (dxi + dyi - d) * 0.5;
            const double u = up + (-dxi - dyi + val);   // i == last: "U[y] = U[last] + ..." of the reference's tail
            U[p] = u;
            Ur[p] = u / r1;
            D[px * ld + p] = val;
            D[p * ld + px] = val;
            // (the bounds of the units this pair belongs to are lowered by the prep kernel that follows)
            if (i == last) {           // relabel: the node of the last slot now lives in slot y
                new_slot = y;
                slot_of_pos[p] = (int32_t)y;
                pos_of_slot[y] = (int32_t)p;
            }
        } else if (i == y) {
            // (py from the winning record, not this thread's pos_of_slot[y]: the thread of the last slot rewrites that entry)
            Ur[py] = __builtin_nan("");   // dead: every q it takes part in is NaN, every unit minimum skips it
            if (y != last) slot_of_pos[py] = -1;
            new_slot = -1;
        }
        if (new_slot >= 0) { KA[p] = nj_key_a(new_slot, n1); KB[p] = nj_key_b(new_slot); }
    }
    if (clocked) ck[4] = wall_clock64();
    const double cs = block_tree256_lane0(val, s);
    if (threadIdx.x == 0) xpart[blockIdx.x] = cs;
    if (clocked) {
        __builtin_amdgcn_s_waitcnt(0);
        ck[5] = wall_clock64();
        unsigned long long* o = clk + 32;
        atomicAdd(&o[0], 1ull);
        if (ck[3] == 0) ck[3] = ck[2];
        for (int k = 1; k < 6; ++k) atomicAdd(&o[k], ck[k] - ck[k - 1]);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int64_t round_up16(int64_t v) { return (v + 15) / 16 * 16; }
static int g_njp_grid = 1024;
int njp_scan_grid() { return g_njp_grid; }


// ---- arena -------------------------------------------------------------------------------------------------------
// Everything the pruned path needs is allocated once per (tips, local ranks) and kept until nj_free: hipMalloc /
// hipFree of the 7.2 GB matrices (and of ~15 vectors per epoch, 8 epochs per run) serialise with the device and
// cost more than the distance kernel when a context builds its matrix again (bench.py's steps).
struct SlabPlan {
    size_t U, Ur, KA, KB, slot_of_pos, pos_of_slot, perm, umin, list, blk_cb, blk_g0, cnt_all, total;
    int64_t list_stride;
};
static size_t align256(size_t v) { return (v + 255) / 256 * 256; }
static int64_t prep_blocks(int64_t P, std::vector<int32_t>* hcb, std::vector<int32_t>* hg0)
{
    const int64_t G16 = (P + kUR - 1) / kUR;
    int64_t cnt = 0;
    for (int64_t c = 0; 32 * c < G16 && c * kTileCols < P - 1; ++c)
        for (int64_t g0 = 32 * c; g0 < G16; g0 += kThreads) {
            if (hcb) { hcb->push_back((int32_t)c); hg0->push_back((int32_t)g0); }
            ++cnt;
        }
    if (cnt == 0) { if (hcb) { hcb->push_back(0); hg0->push_back(0); } cnt = 1; }
    return cnt;
}
static SlabPlan slab_plan(int64_t P, int64_t N, int local_ranks)
{
    SlabPlan p;
    const size_t vec = (size_t)(N + kTileCols + 16);
    const int64_t G16 = (P + kUR - 1) / kUR, S = (P + kTileCols - 1) / kTileCols + 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    p.U = take(vec * 8); p.Ur = take(vec * 8); p.KA = take(vec * 8); p.KB = take(vec * 8);
    p.slot_of_pos = take(vec * 4); p.pos_of_slot = take(vec * 4); p.perm = take(vec * 4);
    p.umin = take((size_t)(S * G16 * 4) * 8);
    p.list_stride = unit_total(P) + kScanBlocks + 64;
    p.list = take((size_t)(p.list_stride * local_ranks) * 4);
    const size_t nprep = (size_t)prep_blocks(P, nullptr, nullptr);
    p.blk_cb = take(nprep * 4); p.blk_g0 = take(nprep * 4);
    p.cnt_all = take((size_t)(2 * local_ranks) * 8);
    p.total = off;
    return p;
}
static size_t matrix_bytes(int64_t P)
{
    const int64_t rows_alloc = (P + kUR - 1) / kUR * kUR + kUR;
    return (size_t)(rows_alloc * round_up16(P) + kTileCols + 16) * sizeof(double);
}

static int njp_arena(NjPruned& q, int64_t N, hipStream_t s)
{
    const int local_ranks = q.sh_world > 1 && q.sh_virtual ? q.sh_world : 1;
    const SlabPlan plan = slab_plan(N, N, local_ranks);
    if (q.arena_D && q.arena_N == N && q.arena_slab_bytes >= plan.total) return DPR_OK;
    void* old[] = { q.arena_D, q.arena_slab[0], q.arena_slab[1] };
    for (void* p : old)
        if (p) (void)hipFree(p);
    q.arena_D = nullptr; q.arena_slab[0] = q.arena_slab[1] = nullptr;
    DPR_HIP(hipMalloc(&q.arena_D, matrix_bytes(N)));
    DPR_HIP(hipMemsetAsync(q.arena_D, 0, matrix_bytes(N), s));
    for (int k = 0; k < 2; ++k) DPR_HIP(hipMalloc(&q.arena_slab[k], plan.total));
    q.arena_slab_bytes = plan.total;
    q.arena_N = N;
    return DPR_OK;
}

// point q at the position-space structures of an epoch with P positions (N = total tips: slot arrays) inside
// matrix buffer `Dbuf` and slab `slab`, and initialise them (all fills ordered on s)
static int njp_alloc_epoch(NjPruned& q, int64_t P, int64_t N, double* Dbuf, char* slab, hipStream_t s)
{
    if (P >= (int64_t)kTileCols * 1024) { set_error("pruned NJ: the list encoding holds fewer than 524288 positions"); return DPR_ERR_ARG; }
    const int local_ranks = q.sh_world > 1 && q.sh_virtual ? q.sh_world : 1;
    const SlabPlan plan = slab_plan(P, N, local_ranks);
    q.P = P;
    q.ld = round_up16(P);
    q.D = Dbuf;
    const int64_t rows_alloc = (P + kUR - 1) / kUR * kUR + kUR;
    // what the permute kernel does not write: columns [P, ld), the group of rows behind position P, the tail pad
    if (int rc = nj_fill_pads(q.D, q.ld, P, P, rows_alloc, kTileCols + 16, false, s)) return rc;
    const size_t vec = (size_t)(N + kTileCols + 16);
    q.U = reinterpret_cast<double*>(slab + plan.U);
    q.Ur = reinterpret_cast<double*>(slab + plan.Ur);
    q.KA = reinterpret_cast<uint64_t*>(slab + plan.KA);
    q.KB = reinterpret_cast<uint64_t*>(slab + plan.KB);
    q.slot_of_pos = reinterpret_cast<int32_t*>(slab + plan.slot_of_pos);
    q.pos_of_slot = reinterpret_cast<int32_t*>(slab + plan.pos_of_slot);
    q.perm = reinterpret_cast<int32_t*>(slab + plan.perm);
    q.umin = reinterpret_cast<uint64_t*>(slab + plan.umin);
    q.list = reinterpret_cast<int32_t*>(slab + plan.list);
    q.blk_cb = reinterpret_cast<int32_t*>(slab + plan.blk_cb);
    q.blk_g0 = reinterpret_cast<int32_t*>(slab + plan.blk_g0);
    q.cnt_all = reinterpret_cast<unsigned long long*>(slab + plan.cnt_all);
    DPR_HIP(hipMemsetAsync(q.U, 0, vec * sizeof(double), s));
    DPR_HIP(hipMemsetAsync(q.Ur, 0xff, vec * sizeof(double), s));   // NaN beyond P
    DPR_HIP(hipMemsetAsync(q.KA, 0, vec * sizeof(uint64_t), s));
    DPR_HIP(hipMemsetAsync(q.KB, 0, vec * sizeof(uint64_t), s));
    DPR_HIP(hipMemsetAsync(q.slot_of_pos, 0xff, sizeof(int32_t) * vec, s));
    DPR_HIP(hipMemsetAsync(q.pos_of_slot, 0xff, sizeof(int32_t) * vec, s));   // -1: slot not alive
    const int64_t G16 = (P + kUR - 1) / kUR, S = (P + kTileCols - 1) / kTileCols + 1;
    q.nunits_alloc = S * G16 * 4;      // four sub-strip bounds per unit
    q.utot = unit_total(P);
    {
        // prep blocks: one strip and up to 256 consecutive row groups each (groups >= 32*cb see the strip)
        std::vector<int32_t> hcb, hg0;
        q.nprep = (int)prep_blocks(P, &hcb, &hg0);
        DPR_HIP(hipMemcpyAsync(q.blk_cb, hcb.data(), sizeof(int32_t) * hcb.size(), hipMemcpyHostToDevice, s));
        DPR_HIP(hipMemcpyAsync(q.blk_g0, hg0.data(), sizeof(int32_t) * hg0.size(), hipMemcpyHostToDevice, s));
        DPR_HIP(hipStreamSynchronize(s));   // the host vectors go out of scope
    }
    // unit-sharded mode: one list and one counter pair per rank held here (all of them for virtual ranks)
    q.list_stride = plan.list_stride;
    DPR_HIP(hipMemsetAsync(q.list, 0, sizeof(int32_t) * (size_t)(q.list_stride * local_ranks), s));
    DPR_HIP(hipMemsetAsync(q.cnt_all, 0, sizeof(unsigned long long) * (size_t)(2 * local_ranks), s));
    hipLaunchKernelGGL(njp_fill_u64_kernel, dim3(256), dim3(256), 0, s, (uint64_t*)q.umin, q.nunits_alloc,
                       enc_f64_host(-__builtin_inf()));
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

static void sort_by_row_sum(std::vector<int32_t>& perm, const std::vector<double>& hU)
{
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t c) {
        const double ua = hU[(size_t)a], uc = hU[(size_t)c];
        if (ua != ua) return false;      // NaN last
        if (uc != uc) return true;
        return ua < uc;
    });
}

int njp_build(NjBuffers& b, hipStream_t s)
{
    {   // blocks of the unit scan (tests shrink it so that every block walks several units and cnt > grid)
        const char* e = std::getenv("DPR_NJP_GRID");
        const int g = e ? std::atoi(e) : 1024;
        g_njp_grid = g < 1 ? 1 : (g > 1024 ? 1024 : g);
    }
    // b.D / b.U hold the matrix and the row sums in tip order (world == 1).  Sort by U ascending.
    const int64_t N = b.N;
    std::vector<double> hU((size_t)N);
    DPR_HIP(hipMemcpyAsync(hU.data(), b.U, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, s));
    DPR_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> perm((size_t)N);
    std::iota(perm.begin(), perm.end(), 0);
    sort_by_row_sum(perm, hU);
    NjPruned& q = b.pr;
    if (int rc = njp_arena(q, N, s)) return rc;
    q.epoch_index = 0;
    if (int rc = njp_alloc_epoch(q, N, N, q.arena_D, q.arena_slab[0], s)) return rc;
    q.utot0 = q.utot;
    DPR_HIP(hipMemcpyAsync(q.perm, perm.data(), sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, s));
    if (std::getenv("DPR_NJ_ITERSTATS")) {
        DPR_HIP(hipMalloc(&q.iterstats, sizeof(uint64_t) * (size_t)(2 * N + 2 + 64)));      // + phase clocks of the prep kernel
        DPR_HIP(hipMemsetAsync(q.iterstats, 0, sizeof(uint64_t) * (size_t)(2 * N + 2 + 64), s));
    }
    dim3 grid((unsigned)((N + kThreads - 1) / kThreads > 64 ? 64 : (N + kThreads - 1) / kThreads), (unsigned)(N < 32768 ? N : 32768));
    hipLaunchKernelGGL(njp_permute_kernel, grid, dim3(kThreads), 0, s, b.D, b.ld, q.D, q.ld, q.perm, N);
    hipLaunchKernelGGL(njp_init_vectors_kernel, dim3((unsigned)((N + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       b.U, q.perm, (const int32_t*)nullptr, N, N, q.U, q.Ur, q.KA, q.KB, q.slot_of_pos, q.pos_of_slot);
    DPR_HIP(hipGetLastError());
    DPR_HIP(hipStreamSynchronize(s));   // `perm` goes out of scope
    // (the tip-order matrix in b.D is dead from here on: the odd epochs use its storage)
    q.active = true;
    return DPR_OK;
}

// New epoch: the live positions (n of them), re-sorted by their CURRENT row sums, become a dense n x n
// position space.  Merges put new nodes into the positions of merged ones, so the order by row sum -- which
// is what keeps the unit bounds tight -- decays, and dead positions still occupy scanned units; late in a
// run most units were scanned every iteration.  Costs one n^2 copy; the first scan of the epoch is a full
// one (bounds start at -inf, no seed).  Called between iterations with the stream idle.
// *rebuilt = false: nothing was done (no candidate left, or fewer than three active nodes).
static int njp_rebuild_epoch(NjBuffers& b, hipStream_t s, bool* rebuilt)
{
    NjPruned& q = b.pr;
    *rebuilt = false;
    NjState st;
    DPR_HIP(hipMemcpy(&st, b.st, sizeof(NjState), hipMemcpyDeviceToHost));
    const int64_t n = st.n, Pold = q.P;
    if (st.status != 0 || n < 3) return DPR_OK;
    std::vector<double> hU((size_t)Pold), hUr((size_t)Pold);
    DPR_HIP(hipMemcpy(hU.data(), q.U, sizeof(double) * (size_t)Pold, hipMemcpyDeviceToHost));
    DPR_HIP(hipMemcpy(hUr.data(), q.Ur, sizeof(double) * (size_t)Pold, hipMemcpyDeviceToHost));
    std::vector<int32_t> perm;
    perm.reserve((size_t)n);
    for (int64_t p = 0; p < Pold; ++p)
        if (hUr[(size_t)p] == hUr[(size_t)p]) perm.push_back((int32_t)p);      // dead positions carry NaN
    if ((int64_t)perm.size() != n) { set_error("njp_rebuild_epoch: live positions do not match the active size"); return DPR_ERR_STATE; }
    sort_by_row_sum(perm, hU);
    int32_t new_px = -1;
    for (int64_t a = 0; a < n; ++a)
        if (perm[(size_t)a] == st.pad) new_px = (int32_t)a;
    if (q.graph) { (void)hipGraphExecDestroy(q.graph); q.graph = nullptr; }
    const NjPruned old = q;              // the old epoch's pointers (read by the permute / init kernels below)
    const int e = old.epoch_index + 1;
    q.epoch_index = e;
    if (int rc = njp_alloc_epoch(q, n, b.N, (e & 1) ? b.D : q.arena_D, q.arena_slab[e & 1], s)) return rc;
    DPR_HIP(hipMemcpyAsync(q.perm, perm.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
    dim3 grid((unsigned)((n + kThreads - 1) / kThreads > 64 ? 64 : (n + kThreads - 1) / kThreads), (unsigned)(n < 32768 ? n : 32768));
    hipLaunchKernelGGL(njp_permute_kernel, grid, dim3(kThreads), 0, s, old.D, old.ld, q.D, q.ld, q.perm, n);
    hipLaunchKernelGGL(njp_init_vectors_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       old.U, q.perm, (const int32_t*)old.slot_of_pos, n, n, q.U, q.Ur, q.KA, q.KB, q.slot_of_pos, q.pos_of_slot);
    DPR_HIP(hipGetLastError());
    // iteration state: position of the last new node in the new space; no seed records, empty lists
    st.pad = new_px;
    st.cnt_list[0] = 0ull; st.cnt_list[1] = 0ull;
    DPR_HIP(hipMemcpyAsync(b.st, &st, sizeof(NjState), hipMemcpyHostToDevice, s));
    if (q.sh_world > 1) {   // gathered records refer to old positions: make them null (key = ~0)
        std::vector<NjRecord> nul((size_t)kScanBlocks);
        for (auto& r : nul) { r.q = 10000.0; r.key = ~0ull; r.d = 0.0; r.pad = 0ull; }
        DPR_HIP(hipMemcpyAsync(b.partials, nul.data(), sizeof(NjRecord) * nul.size(), hipMemcpyHostToDevice, s));
        DPR_HIP(hipStreamSynchronize(s));
    }
    DPR_HIP(hipStreamSynchronize(s));
    *rebuilt = true;
    return DPR_OK;
}

// epoch state only: the next njp_build finds the arena in place
void njp_reset(NjPruned& q)
{
    if (q.graph) { (void)hipGraphExecDestroy(q.graph); q.graph = nullptr; }
    if (q.iterstats) { (void)hipFree(q.iterstats); q.iterstats = nullptr; }
    NjPruned fresh;
    fresh.arena_D = q.arena_D; fresh.arena_slab[0] = q.arena_slab[0]; fresh.arena_slab[1] = q.arena_slab[1];
    fresh.arena_slab_bytes = q.arena_slab_bytes; fresh.arena_N = q.arena_N;
    q = fresh;
}

void njp_free(NjPruned& q)
{
    njp_reset(q);
    void* ptrs[] = { q.arena_D, q.arena_slab[0], q.arena_slab[1] };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    q = NjPruned();
}

// unit-sharded mode: records per rank and in total
static int njp_grid_rank(const NjPruned& q) { return q.sh_world > 1 ? (g_njp_grid / q.sh_world > 0 ? g_njp_grid / q.sh_world : 1) : g_njp_grid; }
static int njp_grid_total(const NjPruned& q) { return q.sh_world > 1 ? njp_grid_rank(q) * q.sh_world : g_njp_grid; }

// prep of rank v (its own units, list and counters); v is ignored outside the unit-sharded mode
static int njp_launch_prep(NjBuffers& b, hipStream_t s, int v = 0)
{
    NjPruned& q = b.pr;
    const bool sh = q.sh_world > 1;
    const int slot = sh && q.sh_virtual ? v : 0;             // local storage index of this rank
    hipLaunchKernelGGL(njp_prep_kernel, dim3((unsigned)q.nprep), dim3(kThreads), 0, s, q.D, q.ld, b.st, q.U, q.Ur, q.Ur,
                       b.xpart, b.partials, (unsigned long long*)q.umin, q.P, q.blk_cb, q.blk_g0, njp_grid_total(q),
                       q.list + (int64_t)slot * q.list_stride, sh ? v : 0, sh ? q.sh_world : 1,
                       sh ? q.cnt_all + 2 * slot : (unsigned long long*)nullptr, sh ? njp_grid_total(q) : -1,
                       q.iterstats ? (unsigned long long*)q.iterstats + 2 * b.N + 2 : (unsigned long long*)nullptr);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

static int njp_launch_post(NjBuffers& b, hipStream_t s)
{
    NjPruned& q = b.pr;
    const bool sh = q.sh_world > 1;
    const unsigned pgrid = (unsigned)((b.N + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(njp_post_kernel, dim3(pgrid), dim3(kThreads), 0, s, q.D, q.ld, b.st, q.U, q.Ur, q.KA, q.KB,
                       q.slot_of_pos, q.pos_of_slot, b.xpart, b.partials, njp_grid_total(q), q.P,
                       b.log_x, b.log_y, b.log_bx, b.log_by, sh ? njp_grid_total(q) : -1,
                       sh ? q.cnt_all : (unsigned long long*)nullptr, sh ? (q.sh_virtual ? q.sh_world : 1) : 0,
                       q.iterstats ? (unsigned long long*)q.iterstats + 2 * b.N + 2 : (unsigned long long*)nullptr);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// one iteration: scan -> post -> prep(next); every kernel reads its iteration index from the device state.
// sample: bracket the launches by HIP events (NjKernelTiming; eager runs only)
static int njp_enqueue_iteration(NjBuffers& b, hipStream_t s, bool sample = false)
{
    NjPruned& q = b.pr;
    if (q.sh_world <= 1) {
        auto mark = [&]() -> int {
            if (!sample) return DPR_OK;
            hipEvent_t e = nullptr;
            DPR_HIP(hipEventCreate(&e));
            b.kt->ev.push_back(e);
            DPR_HIP(hipEventRecord(e, s));
            return DPR_OK;
        };
        if (sample) b.kt->nk = 3;
        if (int rc = mark()) return rc;
        hipLaunchKernelGGL(njp_scan_kernel, dim3(g_njp_grid), dim3(kThreads), 0, s, q.D, q.ld, b.st, q.Ur, q.KA, q.KB,
                           (unsigned long long*)q.umin, q.P, q.list, b.partials, (unsigned long long*)q.iterstats,
                           (const unsigned long long*)nullptr, 0);
        if (int rc = mark()) return rc;
        if (int rc = njp_launch_post(b, s)) return rc;
        if (int rc = mark()) return rc;
        if (int rc = njp_launch_prep(b, s)) return rc;
        return mark();
    }
    // Unit-sharded mode (every rank holds the whole position-space matrix): a unit belongs to rank
    // (strip * G16 + group) mod world for good.  Each rank tests and scans only its own units -- a unit that
    // holds the winner always survives its owner's test, whatever the other ranks' bounds are -- so the unit
    // bounds stay private to their owner and ONE small all-gather per iteration (the block records) is the
    // only exchange; select + merge + update run replicated.
    const int gr = njp_grid_rank(q);
    const int v0 = q.sh_virtual ? 0 : q.sh_rank, v1 = q.sh_virtual ? q.sh_world : q.sh_rank + 1;
    for (int v = v0; v < v1; ++v) {
        const int slot = q.sh_virtual ? v : 0;
        hipLaunchKernelGGL(njp_scan_kernel, dim3((unsigned)gr), dim3(kThreads), 0, s, q.D, q.ld, b.st, q.Ur, q.KA, q.KB,
                           (unsigned long long*)q.umin, q.P, q.list + (int64_t)slot * q.list_stride, b.partials + (int64_t)v * gr,
                           (unsigned long long*)q.iterstats, (const unsigned long long*)(q.cnt_all + 2 * slot), 1);
    }
    DPR_HIP(hipGetLastError());
    if (!q.sh_virtual) {
        if (!q.gather) { set_error("njp: unit-sharded mode without a gather callback"); return DPR_ERR_STATE; }
        if (int rc = q.gather(q.gather_ctx, b.partials, sizeof(NjRecord) * (size_t)gr, s)) return rc;
    }
    if (int rc = njp_launch_post(b, s)) return rc;
    for (int v = v0; v < v1; ++v)
        if (int rc = njp_launch_prep(b, s, v)) return rc;
    return DPR_OK;
}

// enqueue `todo` iterations starting at iteration it0.  The four kernels of an iteration take no
// per-iteration arguments, so kGraphIters iterations are captured once into a hipGraph and replayed;
// iterations beyond it_limit are no-ops.
constexpr int kGraphIters = 32;

static int njp_run_segment(NjBuffers& b, int64_t it0, int64_t todo, hipStream_t s)
{
    NjPruned& q = b.pr;
    const int64_t limit = it0 + todo;
    DPR_HIP(hipMemcpyAsync(&b.st->it_limit, &limit, sizeof(int64_t), hipMemcpyHostToDevice, s));
    DPR_HIP(hipStreamSynchronize(s));   // `limit` is a stack variable
    if (todo <= 0) return DPR_OK;
    {   // list + bound for iteration it0
        const int v0 = q.sh_world > 1 && !q.sh_virtual ? q.sh_rank : 0;
        const int v1 = q.sh_world > 1 ? (q.sh_virtual ? q.sh_world : q.sh_rank + 1) : 1;
        for (int v = v0; v < v1; ++v)
            if (int rc = njp_launch_prep(b, s, v)) return rc;
    }
    const bool timing = b.kt && b.kt->stride > 0 && q.sh_world <= 1;
    const bool use_graph = q.sh_world <= 1 && todo >= kGraphIters && !timing && !std::getenv("DPR_NJ_NOGRAPH");
    if (use_graph && !q.graph) {
        const auto tg0 = std::chrono::steady_clock::now();
        hipGraph_t g = nullptr;
        DPR_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = DPR_OK;
        for (int k = 0; k < kGraphIters && rc == DPR_OK; ++k) rc = njp_enqueue_iteration(b, s);
        hipError_t e = hipStreamEndCapture(s, &g);
        if (rc != DPR_OK) return rc;
        if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
        DPR_HIP(hipGraphInstantiate(&q.graph, g, nullptr, nullptr, 0));
        DPR_HIP(hipGraphDestroy(g));
        if (std::getenv("DPR_NJ_EPOCH_LOG"))
            std::fprintf(stderr, "[njp] graph capture + instantiate (P=%lld): %.2f ms\n", (long long)q.P,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg0).count());
    }
    int64_t done = 0;
    if (use_graph)
        for (; done + kGraphIters <= todo; done += kGraphIters) DPR_HIP(hipGraphLaunch(q.graph, s));
    for (; done < todo; ++done)
        if (int rc = njp_enqueue_iteration(b, s, timing && (it0 + done) % b.kt->stride == 0)) return rc;
    return DPR_OK;
}

const char* njp_kernel_name(int idx)
{
    static const char* names[] = { "njp_scan_kernel", "njp_post_kernel", "njp_prep_kernel" };
    return idx >= 0 && idx < 3 ? names[idx] : "";
}

// enqueue `todo` iterations starting at iteration it0, in epochs: whenever the active size has dropped to
// half of the epoch's positions (and the epoch is large enough to matter) the position space is rebuilt
int njp_run(NjBuffers& b, int64_t it0, int64_t todo, hipStream_t s)
{
    const char* e_min = std::getenv("DPR_NJ_EPOCH_MIN");
    const int64_t epoch_min = e_min ? std::atoll(e_min) : 2048;   // epochs smaller than this are not rebuilt
    const char* e_pct = std::getenv("DPR_NJ_EPOCH_PCT");
    const int64_t pct = e_pct ? std::atoll(e_pct) : 80;           // rebuild once n <= pct% of the epoch's positions (sweep: profiles/epoch_sweep2.sh)
    int64_t it = it0, left = todo;
    if (left <= 0) return njp_run_segment(b, it0, 0, s);
    while (left > 0) {
        const int64_t n = b.N - it, P = b.pr.P;
        int64_t seg = left;
        if (epoch_min > 0 && P >= epoch_min) {
            const int64_t target = P * pct / 100;
            if (n <= target && n >= 3) {
                DPR_HIP(hipStreamSynchronize(s));
                const auto t0 = std::chrono::steady_clock::now();
                bool rebuilt = false;
                if (int rc = njp_rebuild_epoch(b, s, &rebuilt)) return rc;
                // not rebuilt: the run has no candidate left (status != 0: any NaN / inf distance at iteration 0 makes every
                // row sum NaN) and every queued kernel is a no-op -- stop here, dpr_nj_run reports DPR_ERR_NOCAND
                if (!rebuilt) return DPR_OK;
                if (std::getenv("DPR_NJ_EPOCH_LOG"))
                    std::fprintf(stderr, "[njp] epoch rebuild at n=%lld: %.2f ms\n", (long long)n,
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                continue;
            }
            if (n - target < seg) seg = n - target;
        }
        if (int rc = njp_run_segment(b, it, seg, s)) return rc;
        it += seg; left -= seg;
    }
    return DPR_OK;
}

}  // namespace dpr
