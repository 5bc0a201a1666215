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
// Per iteration TWO kernels (three until round 2; what an iteration costs is its chain of dependent memory round
// trips plus one kernel boundary per launch):
//   SCAN(it): the surviving sub-units of the listed units (exact bounds refreshed) + the NEW ROW: the node created by
//             merge it-1 is in QUARANTINE during iteration it -- its row sum (a canonical sum over all slots) is only
//             finished here, by dedicated blocks that also evaluate its pairs from a row buffer and move that buffer
//             into the matrix;
//   POST(it): select + merge + update (indexed by reference slot so that the canonical U[x] summation order is
//             unchanged) fused with the unit tests of iteration it+1: the test lanes need the row sums AFTER the
//             merge, which other blocks of the same launch are still storing, so they recompute them for their own
//             16 rows / 2 columns from the (double-buffered) current row sums and rows x, y -- which is why the update
//             writes the new node's row to a buffer instead of over row x.
// The kernels take no per-iteration arguments (they read the iteration index from the device state), so 32
// iterations are captured into one hipGraph and replayed.
#include "njp_args.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <string>

namespace dpr {

// Row groups (= units of one strip) per test block of the post kernel: 64 below 40 000 positions (1024 row positions per
// block: the row phase of a block is short, NJ 500 -> 482 ms at 30 000 tips), 256 above (fewer blocks, less repeated
// column and record work where the kernel is throughput-bound).  The switch: 38 000-50 000 positions measured within
// 1.5 % of each other at 100 000 tips, 24 000 costs 8 % at 30 000 tips; DPR_NJ_BIG_P moves it (the tests run both shapes).
static int64_t njp_big_p()
{
    static const int64_t v = std::getenv("DPR_NJ_BIG_P") ? std::atoll(std::getenv("DPR_NJ_BIG_P")) : 40000;
    return v;
}
// Round 4 measured the alternatives for the SMALL shape and kept none (history: commits 9d05741, 03fdc34; NOTES.md round 4 items
// 5, 8, 20-25): njp_post2_kernel with one strip x 256 row groups per test block (505 vs 475 ms at 30 000 tips), a three-role
// kernel with one cell per unit instead of the atomic list (507 ms), the same with the list and a seed bound without gathers
// (482 ms: neutral); test blocks of 32 / 128 row groups (513 / 480 ms).  Round 5 removed their switches and instantiations:
// the small shape is njp_post_kernel<64, 1>, the large one njp_post2_kernel<4> (DPR_NJP_POST2=0: the fused njp_post_kernel<256, 4>,
// kept as the independent second implementation the tests compare it with).
static bool njp_post2_on()
{
    static const bool on = !(std::getenv("DPR_NJP_POST2") && std::atoi(std::getenv("DPR_NJP_POST2")) == 0);
    return on;
}
static int njp_tg(int64_t P) { return P < njp_big_p() ? 64 : 256; }
// Strips per test block.  In the small shape a test block is one strip x 64 row groups (the post kernel is a chain of
// dependent round trips there and all blocks are resident at once).  In the large shape one strip x 256 groups left
// ~2 900 blocks of ~160 registers per thread, i.e. several rounds of resident blocks, each paying the whole chain (select,
// rows x / y, row sums): 33 us per launch at 100 000 tips.  So a block keeps its 256 row groups and handles kBigNS strips
// with them, all strips' loads in flight together: 23 us per launch (2 strips: +9 %, 8 strips one load ahead: +4 %, 16: +29 %;
// the block that owns 256 row groups then needs 247 registers, two blocks per CU, and the grid still takes two rounds).
constexpr int kBigNS = 4;
static int njp_ns(int64_t P) { return P < njp_big_p() ? 1 : kBigNS; }
// strips that hold a valid unit for some group of the row block [g0, g0 + tg)
__host__ __device__ inline int64_t njp_strips_of_rows(int64_t g0, int64_t tg, int64_t P)
{
    const int64_t G16 = (P + 16 - 1) / 16;
    const int64_t glast = (g0 + tg < G16 ? g0 + tg : G16) - 1;
    if (glast < 0) return 0;
    int64_t c = glast / 32 + 1;                       // strips with 32 c <= glast
    const int64_t cp = (P - 2) / 512 + 1;             // strips with 512 c < P - 1
    return P >= 2 ? (c < cp ? c : cp) : 0;
}

// Unit ownership of the unit-sharded mode (also exported for the CPU tests of the N > 1 logic).  Units are tested in
// blocks of njp_ns(P) strips x njp_tg(P) consecutive row groups (one strip, strip-major, in the small shape; row-block
// major above); test block t belongs to rank t mod world, and with it its units -- so a rank launches (and pays for) only
// its own share of the test blocks.
int njp_unit_owner(int64_t strip, int64_t group, int64_t P, int world)
{
    const int64_t G16 = (P + kUR - 1) / kUR;
    if (strip < 0 || group < 0 || group >= G16 || group < 32 * strip || strip * kTileCols >= P - 1 || world < 1) return -1;
    int64_t t = 0;
    const int64_t tg = njp_tg(P), ns = njp_ns(P);
    if (ns == 1) {
        for (int64_t c = 0; c < strip; ++c) t += (G16 - 32 * c + tg - 1) / tg;     // test blocks of the strips before
        t += (group - 32 * strip) / tg;
    } else {                                                                        // row-block major, ns strips per block
        const int64_t k = group / tg;
        for (int64_t kk = 0; kk < k; ++kk) t += (njp_strips_of_rows(kk * tg, tg, P) + ns - 1) / ns;
        t += strip / ns;
    }
    return (int)(t % world);
}

// ------------------------------------------------------------------------------------------------
// epoch build: B[a][b] = A[perm[a]][perm[b]]
// ------------------------------------------------------------------------------------------------
// Range of the entries a run has held, for njp_post2_kernel's proof: mm[0] = enc_f64(m'), mm[1] = enc_f64(A') with
//   every entry >= 2 min(m', 0)   and   every |entry| <= 2 A'.
// njp_range_kernel reduces the initial matrix once; the M part of every post launch adds the values it creates through
// njp_note_range: a wave only writes when it moves a word by more than a factor of two -- or across zero, the one fact that
// matters for non-negative input (no wave ever writes then).  The header is CARRIED from epoch to epoch (the entries of a new
// epoch are a subset of the old one's).  (First version: the epoch build's permute kernel reported per wave.  The waves of
// seven of the eight XCDs never saw the others' updates -- plain loads, per-XCD L2 -- so nearly all of 8 M waves went to the
// atomic: an epoch build of 24 000 positions 65 ms instead of 3.6, NJ at 30 000 tips 476 -> 590 ms.)
__device__ __forceinline__ void njp_note_range(unsigned long long* mm, double lo, double hi)
{
    if (lo < 0.0) {
        const double cur = dec_f64(mm[0]);
        if (cur >= 0.0 || lo < 2.0 * cur) atomicMin(&mm[0], (unsigned long long)enc_f64(lo));
    }
    if (hi > 0.0) {
        const double cur = dec_f64(mm[1]);
        if (hi > 2.0 * cur) atomicMax(&mm[1], (unsigned long long)enc_f64(hi));
    }
}

// B[a][b] = A[perm[a]][perm[b]]: a gather of 8-byte elements out of the source row, whose 64-byte lines are each wanted by
// eight different threads.  Every block of an output row runs on ONE XCD (blockIdx.x & 7 = the XCD of a block when gridDim.x
// is a multiple of 8: workgroups go round-robin), so the source row enters one L2 once; with the row's blocks spread over all
// eight XCDs (round 1 - 3) every L2 fetched every row: 8 x the reads, 120 ms for the first epoch of 100 000 tips.
// gridDim.x = 8 x chunks: enough blocks per row that only a few rows (<= 4 MB) are in flight per XCD.
__global__ __launch_bounds__(kThreads) void njp_permute_kernel(const double* __restrict__ A, int64_t lda,
                                                               double* __restrict__ B, int64_t ldb,
                                                               const int32_t* __restrict__ perm, int64_t P)
{
    const int xcd = (int)(blockIdx.x & 7u), chunk = (int)(blockIdx.x >> 3), nchunk = (int)(gridDim.x >> 3);
    for (int64_t a = (int64_t)blockIdx.y * 8 + xcd; a < P; a += (int64_t)gridDim.y * 8) {
        const double* row = A + (int64_t)perm[a] * lda;
        for (int64_t b = (int64_t)chunk * kThreads + threadIdx.x; b < P; b += (int64_t)nchunk * kThreads)
            B[a * ldb + b] = row[perm[b]];
    }
}

static void njp_launch_permute(const double* A, int64_t lda, double* B, int64_t ldb, const int32_t* perm, int64_t n, hipStream_t s)
{
    // blocks per row: one per 1 024 columns, 1 ... 64 (the first epoch of 100 000 tips: 97 / 80 / 70 ms
    // with 24 / 32 / 64, of 30 000 tips: 10.9 / 6.5 / 6.0 ms with 4 / 15 / 32 -- 120 and 9.0 ms with the rows spread over the XCDs)
    int64_t chunks = (n + 1023) / 1024;
    chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
    const int64_t gy = (n + 7) / 8;
    dim3 grid((unsigned)(8 * chunks), (unsigned)(gy < 32768 ? gy : 32768));
    hipLaunchKernelGGL(njp_permute_kernel, grid, dim3(kThreads), 0, s, A, lda, B, ldb, perm, n);
}

// range of the n x n matrix A (slot space, rows contiguous) into mm: once per run, before the first large-shape epoch, and
// after a hand-back from the streaming loop (which creates values nobody tracked).  2 048 blocks, one report each.
__global__ __launch_bounds__(kThreads) void njp_range_kernel(const double* __restrict__ A, int64_t lda, int64_t n, unsigned long long* __restrict__ mm)
{
    __shared__ double slo[kThreads / 64], shi[kThreads / 64];
    double lo = __builtin_inf(), hi = 0.0;
    for (int64_t a = blockIdx.x; a < n; a += gridDim.x) {
        const double* row = A + a * lda;
        for (int64_t b = threadIdx.x; b < n; b += kThreads) {
            const double v = row[b];
            lo = fmin(lo, v); hi = fmax(hi, fabs(v));
        }
    }
    lo = wave_fmin(lo); hi = wave_fmax(hi);
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) { lo = fmin(lo, slo[w]); hi = fmax(hi, shi[w]); }
        atomicMin(&mm[0], (unsigned long long)enc_f64(lo));
        atomicMax(&mm[1], (unsigned long long)enc_f64(hi));
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
    KA[p] = nj_key_a_dev(slot, n);
    KB[p] = nj_key_b(slot);
    slot_of_pos[p] = slot;
    pos_of_slot[slot] = (int32_t)p;
}

__global__ void njp_fill_u64_kernel(uint64_t* __restrict__ a, int64_t cnt, uint64_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) a[i] = v;
}

// ------------------------------------------------------------------------------------------------
// epoch start: every unit is listed with all four sub-units (the bounds start at -inf and there is no
// seed, so no test could rule anything out); nothing is in quarantine
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void njp_list_all_kernel(NjpArgs a)
{
    const int tid = threadIdx.x;
    const int64_t it = a.st->it;
    if (a.st->status != 0) return;
    const int tb = a.sh_rank + (int)blockIdx.x * a.sh_world;      // this rank's blockIdx.x-th test block
    const int cb0 = a.blk_cb[tb];
    const int64_t g = (int64_t)a.blk_g0[tb] + tid;
    const int64_t G16 = (a.P + kUR - 1) / kUR;
    const int tg = a.tg;
    int64_t send = njp_strips_of_rows((int64_t)a.blk_g0[tb], tg, a.P);
    if (a.ns == 1) send = cb0 + 1;
    else if (send > cb0 + a.ns) send = cb0 + a.ns;
    const int lane = tid & 63;
    for (int cb = cb0; cb < (int)send; ++cb) {
        const bool keep = g < G16 && g >= 32 * (int64_t)cb && tid < tg;
        const unsigned long long mask = __ballot(keep);
        unsigned long long base = 0;
        if (lane == 0 && mask) base = atomicAdd(&a.cnt[it % 3], (unsigned long long)__popcll(mask));
        base = __shfl(base, 0, 64);
        if (keep) a.list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)((0xFu << 28) | ((uint32_t)cb << 18) | (uint32_t)g);
    }
}

// ------------------------------------------------------------------------------------------------
// SCAN(it).  Blocks [0, ugrid): the listed units -- block b takes entries b, b+G, ... and always writes
// partials[b] when it had work, so the unit records of a scan are partials[0 .. min(cnt, grid)).  The
// lane-level best carries the positions of the pair and its distance, so nothing is looked up after the
// reduction.  The node in quarantine has a NaN row sum: the unit scans skip its row and column (and the
// exact sub-unit minima they store leave it out).
// Blocks [0, nrb): the NEW ROW.  They finish the row sum U[x] of the node created by the
// previous merge from the chunk partials (canonical order), evaluate its pairs against every live
// position from the row buffer, and move the buffered row into the matrix (nobody reads that row validly
// during this launch); the first of them stores U[x].
// ------------------------------------------------------------------------------------------------
// The leading scalar parameters (NjpHead) repeat what the first round trip needs: with -mllvm -amdgpu-kernarg-preload-count
// (Makefile) they arrive in SGPRs with the wave, so hop 1 leaves without waiting for the s_load of the argument block.
// kRS (row-sharded mode, njr.hip): D holds this rank's chunks of rows (njp_lrow), only the owner of a position stores its row,
// the first new-row block adds the rank's header record (the row sum it derived, for the cross-check of the replicated state),
// and -- mailbox plan -- the last block of the launch to finish pushes the rank's records into every rank's window.
template <bool kRS>
__global__ __launch_bounds__(kThreads) void njp_scan_kernel(NjState* h_st, const int32_t* h_list, const unsigned long long* h_cnt,
                                                            const double* h_xpart, int h_nrl, NjpArgs a)
{
    __shared__ double sq[kThreads / 64], sd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64], sp[kThreads / 64];
    __shared__ double stree[kThreads];
    __shared__ unsigned int s_last;

    const int tid = threadIdx.x;
    // hop 1: state line, all three list counters and (speculatively) this block's first list entry / chunk partial
    const int64_t it = h_st->it, limit = h_st->it_limit, N = h_st->N;
    // (the new-row blocks come FIRST in the grid: a 1000-block grid takes ~1.5 us to start, and their chain -- chunk
    // partials, canonical tree, division, candidates -- is the longer one)
    const int nrl = h_nrl;
    const bool unit_block = (int)blockIdx.x >= nrl;
    const int ub = (int)blockIdx.x - nrl;                        // unit block index
    const int32_t first = unit_block ? h_list[ub] : 0;
    const double xp0 = unit_block ? 0.0 : h_xpart[tid];          // (the array is padded to a multiple of 256 entries)
    const unsigned long long cl0 = h_cnt[0], cl1 = h_cnt[1], cl2 = h_cnt[2];      // (not a second, dependent load behind `it`)
    // new-row blocks: this thread's two positions of the buffered row (BOTH buffers: which one is current needs `it`) and
    // their vectors -- nothing here waits for the state line (the vectors are padded past the last block's columns)
    const int64_t j0 = (int64_t)blockIdx.x * kTileCols + 2 * tid;
    v2d dv0, dv1, uv; ulonglong2 kav, kbv;
    dv0.x = dv0.y = 0.0; dv1 = dv0; uv = dv0; kav = make_ulonglong2(0ull, 0ull); kbv = kav;
    if (!unit_block) {
        dv0 = *reinterpret_cast<const v2d*>(a.R + j0);
        dv1 = *reinterpret_cast<const v2d*>(a.R + a.vstride + j0);
        uv = *reinterpret_cast<const v2d*>(a.Ur + j0);
        kav = *reinterpret_cast<const ulonglong2*>(a.KA + j0); kbv = *reinterpret_cast<const ulonglong2*>(a.KB + j0);
    }
    const int32_t pn0 = h_st->pnew[0], pn1 = h_st->pnew[1];      // (both with the state line)
    const int64_t pz = (int64_t)((it & 1) ? pn1 : pn0);
    if (blockIdx.x == 0 && tid == 0) h_st->itb = it;             // the post kernel's iteration index (it advances `it` itself); also beyond the limit
    if (it >= limit || h_st->status != 0) return;
    NJP_STAMP(0, 0, true);
    const int64_t P = a.P;
    double bq = 10000.0, bd = 0.0;  // the reference's init value
    uint64_t bk = ~0ull, bp = 0;
    NjRecord* rec_out;

    if (!unit_block) {
        // ---------------------------------------------------------------- new-row block
        const int r = (int)blockIdx.x;
        rec_out = a.partials + a.urecs + r;
        if (pz < 0) {
            if (tid == 0) { NjRecord rec; rec.q = 10000.0; rec.key = ~0ull; rec.d = 0.0; rec.pad = 0ull; *rec_out = rec; }
            if (kRS) {
                if (r == 0 && tid == 0) njr_store_record(a.partials + a.rec_off - 1, 10000.0, ~0ull, it == a.rs_fault_it ? 1.0e-3 : 0.0, a.rs_seq_base + (unsigned long long)(it + 1));
                njr_scan_publish(a, it, &s_last);
            }
            return;
        }
        const int64_t n = N - it;
        const int64_t nchunk = (n + 1 + kThreads - 1) / kThreads;        // chunks of the previous update (n + 1 slots)
        double acc = tid < nchunk ? xp0 : 0.0;
        for (int64_t c = tid + kThreads; c < nchunk; c += kThreads) acc += a.xpart[c];
        // hop 2: only the keys of the new node itself (needed after the sum)
        const v2d dv = ((it + 1) & 1) ? dv1 : dv0;                             // buffer (it + 1) & 1 was written by POST(it - 1)
        const uint64_t kax = a.KA[pz], kbx = a.KB[pz];
        NJP_STAMP(0, 1, true);
        double ux = block_tree256_lane0(acc, stree);
        if (tid == 0) stree[0] = ux;
        __syncthreads();
        ux = stree[0];
        NJP_STAMP(0, 2, false);
        const double urx = ux / (double)(n - 2);
        if (r == 0 && tid == 0) a.U[(it & 1) * a.vstride + pz] = ux;
        if (kRS && r == 0 && tid == 0)       // header record: this rank's view of the replicated state (inert for every reduction: key = ~0)
            njr_store_record(a.partials + a.rec_off - 1, 10000.0, ~0ull, it == a.rs_fault_it ? ux * 0.5 + 1.0e-3 : ux, a.rs_seq_base + (unsigned long long)(it + 1));
        if (j0 < P && njp_owns<kRS>(a, pz)) *reinterpret_cast<v2d*>(a.D + njp_lrow<kRS>(a, pz) * a.ld + j0) = dv;        // (R[pz] = 0: the diagonal stays 0)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double d = k ? dv.y : dv.x, uj = k ? uv.y : uv.x;       // dead / padding / pz itself: NaN row sum -> NaN q
            const uint64_t kaj = k ? kav.y : kav.x, kbj = k ? kbv.y : kbv.x;
            const uint64_t pj = (uint64_t)(j0 + k);
            best_update4(bq, bk, bp, bd, (d - urx) - uj, kax | kbj, (uint64_t)pz | (pj << 32), d);   // (i = x, j)
            best_update4(bq, bk, bp, bd, (d - uj) - urx, kaj | kbx, pj | ((uint64_t)pz << 32), d);   // (i = j, j = x)
        }
        wave_best4(bq, bk, bp, bd);
    } else {
        // ---------------------------------------------------------------- unit block
        rec_out = a.partials + a.rec_off + ub;
        const int m3 = (int)(it % 3);
        const int64_t cnt = (int64_t)(m3 == 0 ? cl0 : m3 == 1 ? cl1 : cl2);
        if ((int64_t)ub >= cnt) {
            if (a.all_defined && tid == 0) {      // unit-sharded mode: every record of the gathered array is defined
                if (kRS) njr_store_record(rec_out, 10000.0, ~0ull, 0.0, 0ull);
                else { NjRecord rec; rec.q = 10000.0; rec.key = ~0ull; rec.d = 0.0; rec.pad = 0ull; *rec_out = rec; }
            }
            if (kRS) njr_scan_publish(a, it, &s_last);
            return;
        }
        const int64_t G16 = (P + kUR - 1) / kUR;
        // Two passes per unit instead of a (q, key, positions, d) compare-and-select per candidate (64 candidates
        // per lane and unit, ~20 VALU instructions each, were 2 us of this kernel's critical path): pass 1 computes
        // the candidates' q and their minimum over the WAVE; pass 2 -- only when that minimum reaches the wave's
        // best so far -- looks for the candidates equal to it (rare, wave-uniform branch) and keeps the smallest key.
        // bq is wave-uniform; (bk, bp, bd) is this lane's best candidate AT q == bq (bk == ~0: none).
        int64_t scanned = 0;
        const int wv = tid >> 6;                         // this wave's sub-strip of every unit
        for (int64_t e = ub; e < cnt; e += a.ugrid, ++scanned) {
            const uint32_t code = (uint32_t)__builtin_amdgcn_readfirstlane(e == (int64_t)ub ? first : a.list[e]);
            if (!((code >> (28 + wv)) & 1u)) continue;   // the bound of this wave's sub-unit rules it out (wave-uniform)
            const int cb = (int)((code >> 18) & 1023u);
            const int64_t g_s = (int64_t)(code & 0x3FFFFu);
            const int64_t c0 = (int64_t)cb * kTileCols, a0 = g_s * kUR;
            // Rows a0 .. a0+15 are always read: the matrix has a group of rows behind position P and the
            // vectors carry NaN row sums there (njp_alloc_epoch), so rows >= P behave like dead rows.
            const int64_t b0 = c0 + 2 * tid, b1 = b0 + 1;
            const v2d ubv = *reinterpret_cast<const v2d*>(a.Ur + b0);
            const ulonglong2 kav = *reinterpret_cast<const ulonglong2*>(a.KA + b0), kbv = *reinterpret_cast<const ulonglong2*>(a.KB + b0);
            const double ub0 = ubv.x, ub1 = ubv.y;
            const uint64_t ka0 = kav.x, ka1 = kav.y, kb0 = kbv.x, kb1 = kbv.y;
            const v2d* basep = reinterpret_cast<const v2d*>(a.D + njp_lrow<kRS>(a, a0) * a.ld + c0) + tid;
            const int64_t ld2 = a.ld >> 1;
            const bool diag = a0 < c0 + kTileCols;
            const bool live0 = ub0 == ub0, live1 = ub1 == ub1;   // dead columns carry NaN row sums
            v2d v[kUR];
#pragma unroll
            for (int u8 = 0; u8 < kUR; ++u8) v[u8] = __builtin_nontemporal_load(basep + (int64_t)u8 * ld2);
            // the 16 rows' row sums and keys: lane l holds row a0 + (l & 15) (three coalesced loads, no scalar-register
            // pressure); v_readlane hands them out -- the keys only in the rare branch of pass 2
            const double ua_l = a.Ur[a0 + (tid & 15)];
            const uint64_t kaa_l = a.KA[a0 + (tid & 15)], kba_l = a.KB[a0 + (tid & 15)];
            double ua[kUR];
#pragma unroll
            for (int u8 = 0; u8 < kUR; ++u8) ua[u8] = readlane_f64(ua_l, u8);
            NJP_STAMP(0, 1, true);
            if (diag) {   // block-uniform and rare (units on the diagonal): mask the entries with column >= row
                const int ib0 = (int)b0, ia0 = (int)a0;
#pragma unroll
                for (int u8 = 0; u8 < kUR; ++u8) {
                    const int ar = ia0 + u8;
                    v[u8].x = (ib0 < ar) ? v[u8].x : __builtin_nan("");
                    v[u8].y = (ib0 + 1 < ar) ? v[u8].y : __builtin_nan("");
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
            NJP_STAMP(0, 2, false);
            if (wm <= bq) {                           // wave-uniform; the q are recomputed (same operations, same bits)
                if (wm < bq) { bq = wm; bk = ~0ull; }   // rather than kept: 128 registers less, twice the blocks per CU
                if (a.dbg != nullptr && (it == a.dbg_it || a.dbg_it == -2) && (tid & 63) == 0) atomicAdd(&a.dbg[(a.dbg_it == -2 ? 0 : (int)blockIdx.x) * 8 + 6], 1ull);
#pragma unroll
                for (int u8 = 0; u8 < kUR; ++u8) {
                    if (__builtin_amdgcn_ballot_w64(rowq[u8] == wm) != 0ull) {      // rare; the four q are recomputed (same bits)
                        if (a.dbg != nullptr && (it == a.dbg_it || a.dbg_it == -2) && (tid & 63) == 0) atomicAdd(&a.dbg[(a.dbg_it == -2 ? 0 : (int)blockIdx.x) * 8 + 5], 1ull);
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
            NJP_STAMP(0, 7, false);
            // exact minimum of this wave's sub-unit -> its bound (no cross-wave step)
            m = wave_fmin(m);
            if ((tid & 63) == 0) a.umin[((int64_t)cb * G16 + g_s) * 4 + wv] = enc_f64(m);
        }
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
        if (tid == 0) {
            if (ub == 0) atomicAdd(&a.st->units_scanned, (unsigned long long)cnt);   // statistics
        }
    }
    NJP_STAMP(0, 3, false);
    if ((tid & 63) == 0) { sq[tid >> 6] = bq; sk[tid >> 6] = bk; sp[tid >> 6] = bp; sd[tid >> 6] = bd; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) best_update4(bq, bk, bp, bd, sq[w], sk[w], sp[w], sd[w]);
        if (kRS && unit_block) njr_store_record(rec_out, bq, bk, bd, bp);      // (read by the launch's last block, possibly on another XCD)
        else {
            NjRecord rec;
            rec.q = bq; rec.key = bk; rec.d = bd; rec.pad = bp;   // pad = pos_i | pos_j << 32
            *rec_out = rec;
        }
    }
    if (kRS) njr_scan_publish(a, it, &s_last);
    NJP_STAMP(0, 4, true);
}

// after the last enqueued iteration: what the next scan's new-row blocks would materialise (row sum of the node in
// quarantine, its row in the matrix), so that hooks, epoch rebuilds and a resumed run find them in memory.  The node
// stays in quarantine (Ur = NaN): the next scan repeats the two stores with the same values.
template <bool kRS>
__global__ __launch_bounds__(kThreads) void njp_finish_kernel(NjpArgs a)
{
    __shared__ double stree[kThreads];
    const int tid = threadIdx.x;
    const int64_t it = a.st->it, N = a.st->N;
    const int64_t pz = (int64_t)a.st->pnew[it & 1];
    if (a.st->status != 0 || pz < 0) return;
    const int64_t n = N - it;
    const int64_t nchunk = (n + 1 + kThreads - 1) / kThreads;
    double acc = 0.0;
    for (int64_t c = tid; c < nchunk; c += kThreads) acc += a.xpart[c];
    const double* __restrict__ Rz = a.R + ((it + 1) & 1) * a.vstride;
    const int64_t j0 = (int64_t)blockIdx.x * kTileCols + 2 * tid;
    const v2d dv = *reinterpret_cast<const v2d*>(Rz + j0);
    const double ux = block_tree256_lane0(acc, stree);
    if (blockIdx.x == 0 && tid == 0) a.U[(it & 1) * a.vstride + pz] = ux;
    if (j0 < a.P && njp_owns<kRS>(a, pz)) *reinterpret_cast<v2d*>(a.D + njp_lrow<kRS>(a, pz) * a.ld + j0) = dv;
}

// ------------------------------------------------------------------------------------------------
// POST(it) = select + merge + update + the unit tests of iteration it + 1.
// Every block reduces the scan records to the winner for itself.  Then
//  * update role (blocks [ntest, ntest + ceil(N / 256))), indexed by REFERENCE slot i so that the chunk sums
//    of U[x] keep the canonical order: new row sums into the OTHER U buffer (the test role of this launch
//    reads the current one), the new node's row into the row buffer R[it & 1] and its column into the matrix
//    (rows x and y themselves stay untouched in this launch: the test role reads them), the new node goes
//    into quarantine (Ur = NaN), the node of the last slot is relabelled to slot y;
//  * test role (blocks [0, ntest): one strip, one lane per unit): it needs the row sums AFTER this merge,
//    which other blocks are only just storing, so it recomputes them for its 16 rows and 2 columns from the
//    current buffer and rows x, y (same arithmetic, same bits); seed bound = the scan records re-evaluated
//    the same way; the row of the node that LEAVES quarantine (merge it - 1, row buffer R[(it - 1) & 1]) is
//    folded into the bounds of the sub-units it crosses; survivors are appended to the list of scan it + 1.
// ------------------------------------------------------------------------------------------------
// (large shape: DPR_NJP_BIG_WAVES = waves per SIMD the compiler has to leave room for -- experiment knob)
#ifndef DPR_NJP_BIG_WAVES
#define DPR_NJP_BIG_WAVES 1
#endif
// kRS (row-sharded mode, njr.hip): rows x and y are not read from the matrix -- which holds this rank's chunks only -- but from
// the columns px / py every rank extracted from its own rows and exchanged (a.rs_rows; the matrix is symmetric, bit for bit); the
// new node's column is stored into own rows only; the test blocks are this rank's (blk_cb / blk_g0 list them).
template <int kTG, int kNS, bool kRS>
__global__ __launch_bounds__(kThreads, (kNS > 1 ? DPR_NJP_BIG_WAVES : 1)) void njp_post_kernel(NjState* h_st, const NjRecord* h_partials, const unsigned long long* h_cnt, const int32_t* h_blk_cb,
                                                                                              const int32_t* h_blk_g0, const int32_t* h_pos_of_slot, int h_ntest, int h_nupd, NjpArgs a)
{
    __shared__ double s[kThreads];
    __shared__ double sq[kThreads / 64], sdd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64], spp[kThreads / 64];
    __shared__ double sseed[kThreads / 64];
    const int tid = threadIdx.x;
    // block roles: test blocks first, then the update blocks (the other way round -- the short update blocks ahead of the
    // test blocks when the grid does not fit on the chip at once -- measured 5 % slower at 100 000 tips)
    const int bx = (int)blockIdx.x;
    const bool test_block = bx < h_ntest;
    const int tbi = bx, ubi = bx - h_ntest;
    const int tb = test_block ? a.sh_rank + tbi * a.sh_world : 0;      // this rank's tbi-th test block
    // hop 1: state line, scan records, and what each role can address without knowing the winner
    const int64_t it = h_st->itb;   // stable: the writer below only advances st->it
    const int64_t limit = h_st->it_limit, N = h_st->N;
    const int32_t pn0 = h_st->pnew[0], pn1 = h_st->pnew[1];      // (both with the state line)
    const int64_t pz = (int64_t)((it & 1) ? pn1 : pn0);          // node leaving quarantine (its U was stored by SCAN(it))
    const int nrec_all = a.urecs + a.nrb;
    NjRecord r0; r0.q = 10000.0; r0.key = ~0ull; r0.d = 0; r0.pad = 0;
    constexpr int kMine = 5;           // records per thread loaded up front (1280: the default 1024 unit records + 256 new-row records)
    NjRecord mine[kMine] = { r0, r0, r0, r0, r0 };
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
        const int idx = tid + k * kThreads;
        if (idx < nrec_all) mine[k] = h_partials[idx];
    }
    // seed candidate of the test role, ONE per thread (two fp64 divisions each): the last threads take the new-row
    // records, the others a unit record (every sstride-th when all of them are defined)
    const int nseed_rows = a.nrb < kThreads / 2 ? a.nrb : kThreads / 2;
    const int nseed_units = kThreads - nseed_rows;
    const int64_t sstride = a.all_defined && a.urecs >= 2 * nseed_units ? a.urecs / nseed_units : 1;
    const bool seed_is_unit = tid < nseed_units;
    NjRecord cand = r0;
    if (test_block) {
        if (!seed_is_unit) cand = h_partials[a.urecs + (tid - nseed_units)];
        else if ((int64_t)tid * sstride < a.urecs) cand = h_partials[(int64_t)tid * sstride];
    }
    const unsigned long long cl0 = h_cnt[0], cl1 = h_cnt[1], cl2 = h_cnt[2];      // all three list counters with the first round trip (not a second, dependent load)
    const int m3 = (int)(it % 3);
    const unsigned long long cnt_raw = a.all_defined ? (unsigned long long)a.urecs : (m3 == 0 ? cl0 : m3 == 1 ? cl1 : cl2);
    const int64_t uvalid = (int64_t)(cnt_raw < (unsigned long long)a.urecs ? cnt_raw : (unsigned long long)a.urecs);   // unit records written by SCAN(it)
    const double* __restrict__ Uc = a.U + (it & 1) * a.vstride;
    double* __restrict__ Un = a.U + ((it + 1) & 1) * a.vstride;
    const int64_t P = a.P;
    const int64_t G16 = (P + kUR - 1) / kUR;
    const int64_t pclamp = (P + 1) & ~(int64_t)1;     // an even index behind the last position (16-byte aligned pair loads)
    // update role
    const int64_t i = (int64_t)ubi * kThreads + tid;     // reference slot
    int64_t p = -1;
    double up = 0.0;
    // test role: the block's kTG row groups are 16 kTG consecutive positions, read COALESCED in chunks of 512 (thread t:
    // positions rbase + 512 c + 2 t, + 1) -- one lane per group with 16 consecutive values each would touch 64 lines per
    // wave instruction and push 8 x the bytes through L1.  The block walks nsb <= kNS strips with these rows; what a
    // strip needs (its columns' row sums, rows x / y over its columns, the bounds of its units) is loaded up front.
    struct ColData { v2d ucol, dxc, dyc, rzc; ulonglong2 um0, um1; };
    int cb0 = 0, nsb = 0;
    int64_t g = 0, rbase = 0;
    bool have_g = false;
    constexpr int kRC = kTG * kUR / kTileCols;     // 512-position chunks of the block's row range
    v2d urow[kRC];
    ColData cd[kNS];          // every strip of the block in flight at once (walking them one load ahead cost ~1.5 us per strip)
#pragma unroll
    for (int k = 0; k < kNS; ++k) {
        cd[k].ucol.x = 0.0; cd[k].ucol.y = 0.0; cd[k].dxc = cd[k].ucol; cd[k].dyc = cd[k].ucol; cd[k].rzc = cd[k].ucol;
        cd[k].um0 = make_ulonglong2(0ull, 0ull); cd[k].um1 = cd[k].um0;
    }
    // hop-1 part of a strip: nothing here depends on the winner
    auto load_a = [&](int cb, ColData& c) {
        const int64_t pc0 = (int64_t)cb * kTileCols + 2 * tid;                      // this thread's two strip columns (< P + 512)
        c.ucol = *reinterpret_cast<const v2d*>(Uc + pc0);
        if (have_g && g >= 32 * (int64_t)cb) {
            const unsigned long long* up = a.umin + ((int64_t)cb * G16 + g) * 4;
            c.um0 = *reinterpret_cast<const ulonglong2*>(up); c.um1 = *reinterpret_cast<const ulonglong2*>(up + 2);
        }
    };
    if (test_block) {
        cb0 = h_blk_cb[tb];
        g = (int64_t)h_blk_g0[tb] + tid;
        have_g = g < G16 && tid < kTG;
        rbase = (int64_t)h_blk_g0[tb] * kUR;
        if (kNS == 1) nsb = 1;
        else {
            const int64_t send = njp_strips_of_rows((int64_t)h_blk_g0[tb], kTG, P);
            nsb = (int)(send - cb0 < kNS ? send - cb0 : kNS);
        }
#pragma unroll
        for (int k = 0; k < kNS; ++k)
            if (k < nsb) load_a(cb0 + k, cd[k]);
#pragma unroll
        for (int c = 0; c < kRC; ++c) {
            const int64_t pp = rbase + c * kTileCols + 2 * tid;
            urow[c] = *reinterpret_cast<const v2d*>(Uc + (pp < P ? pp : pclamp));      // behind P: padding (NaN = dead)
        }
    } else {
        p = (int64_t)h_pos_of_slot[i];                 // (the slot arrays are padded past N)
    }
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
        const int idx = tid + k * kThreads;
        if (idx >= uvalid && idx < a.urecs) mine[k] = r0;   // not written by this iteration's scan
    }
    if (!test_block) up = (i < N && p >= 0) ? Uc[p] : 0.0;      // hop 1b (address needs the position only)
    if (h_st->status != 0 || it >= limit) return;
    const int64_t n = N - it;
    if (n < 3) return;
    if (!test_block && (!a.do_update || (int64_t)ubi * kThreads >= n)) return;
    if (test_block && !a.do_tests) return;
    NJP_STAMP(1, 0, false);
    NJP_STAMP(1, 1, true);

    // ---- select: reduce the records (thrust::min_element, src/neighborJoining.cu:214)
    double bq = 10000.0, d = 0.0; uint64_t bk = ~0ull, bp = 0;
    const int nmine = (nrec_all + kThreads - 1) / kThreads;       // block-uniform: records per thread that exist at all
#pragma unroll
    for (int k = 0; k < kMine; ++k)
        if (k < nmine) best_update4(bq, bk, bp, d, mine[k].q, mine[k].key, mine[k].pad, mine[k].d);
    for (int64_t idx = tid + kMine * kThreads; idx < nrec_all; idx += kThreads)
        if (idx < uvalid || idx >= a.urecs) best_update4(bq, bk, bp, d, a.partials[idx].q, a.partials[idx].key, a.partials[idx].pad, a.partials[idx].d);
    wave_best4(bq, bk, bp, d);
    if ((tid & 63) == 0) { sq[tid >> 6] = bq; sk[tid >> 6] = bk; spp[tid >> 6] = bp; sdd[tid >> 6] = d; }
    __syncthreads();
    bq = sq[0]; bk = sk[0]; bp = spp[0]; d = sdd[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) best_update4(bq, bk, bp, d, sq[w], sk[w], spp[w], sdd[w]);

    NJP_STAMP(1, 2, false);
    const int64_t last = n - 1;
    if (bk == ~0ull || !(bq < 10000.0)) {     // (q == 10000.0 is no candidate: src/neighborJoining.cu:134-141 compares with a strict `<`)
        if (!test_block && i == last) a.st->status = 1;
        return;
    }
    const int64_t ki = (int64_t)(bk & 0xFFFFFFull), kj = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t pi = (int64_t)(bp & 0xffffffffull), pj = (int64_t)(bp >> 32);
    const int64_t x = ki < kj ? ki : kj, y = ki < kj ? kj : ki;
    const int64_t px = ki < kj ? pi : pj, py = ki < kj ? pj : pi;
    const int64_t n1 = n - 1;
    const double r1 = (double)(n1 - 2);
    const double* __restrict__ rowx = kRS ? a.rs_rows : a.D + px * a.ld;
    const double* __restrict__ rowy = kRS ? a.rs_rows + a.rs_slice : a.D + py * a.ld;
    // entry of row x / row y at position p (pointer: also used for aligned pairs, p even)
    auto RX = [&](int64_t pp_) -> const double* { return kRS ? rowx + njp_xoff(a, pp_) : rowx + pp_; };
    auto RY = [&](int64_t pp_) -> const double* { return kRS ? rowy + njp_xoff(a, pp_) : rowy + pp_; };
    if (kRS && a.rs_plan == kNjrMailbox) {
        // the column slices of this iteration must have arrived from every rank (njr_extract_kernel sends them)
        __shared__ int s_fail;
        if (tid == 0) s_fail = 0;
        __syncthreads();
        if (tid < a.rs_world) {
            const unsigned long long* f = njr_win_rowflag(a.rs_win[a.rs_rank], a.rs_lay, tid);
            const unsigned long long want = a.rs_seq_base + (unsigned long long)(it + 1), t0 = wall_clock64();
            while (njr_ld_sys(f) != want) {
                if (wall_clock64() - t0 > a.rs_poll_ticks) { s_fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (s_fail) {
            if (!test_block && i == last) a.st->status = 3;
            return;
        }
    }

    if (!test_block) {
        // ------------------------------------------------------------------------------ update role
        double* __restrict__ Rw = a.R + (it & 1) * a.vstride;
        double val = 0.0;
        if (i < n) {
            if (i == last) {
                // single writer of the log and the state (reads U[px], U[py] of the current buffer)
                const double r = (double)(n - 2);
                double blX = (d + Uc[px] / r - Uc[py] / r) * 0.5;
                double blY = d - blX;
                if (blX < 0) { blY += blX; blX = 0; }
                if (blY < 0) { blX += blY; blY = 0; }
                a.log_x[it] = (int32_t)x; a.log_y[it] = (int32_t)y; a.log_bx[it] = blX; a.log_by[it] = blY;
                a.st->x = (int32_t)x; a.st->y = (int32_t)y; a.st->d = d; a.st->q = bq;
                a.st->n = n1; a.st->it = it + 1;
                a.st->pnew[(it + 1) & 1] = (int32_t)px;
                // the dead position carries NaN in BOTH row-sum buffers from now on (it is never written again).  The
                // store into the CURRENT buffer is this thread's, behind its own read of U[py] above: any other thread
                // storing it could overtake that read (seen once in 30 000 iterations as a NaN branch length)
                a.U[(it & 1) * a.vstride + py] = __builtin_nan("");
                // list counters of the scan after next (nobody reads or appends to them in this launch)
                a.st->cnt_list[(it + 2) % 3] = 0ull;
                for (int v = 0; v < a.cnt_ranks; ++v) a.cnt_all[4 * v + (it + 2) % 3] = 0ull;
            }
            int64_t new_slot = i;
            if (i != x && i != y) {
                const double dxi = *RX(p), dyi = *RY(p);
                val = nj_val(dxi, dyi, d);
                const double u = nj_unew(up, dxi, dyi, val);   // i == last: "U[y] = U[last] + ..." of the reference's tail
                Un[p] = u;
                a.Ur[p] = u / r1;
                Rw[p] = val;                   // row of the new node: into the matrix by the next scan's new-row blocks
                // its column: 8 bytes into n different lines.  Stored write-through (sc1): as plain stores they leave n dirty
                // 128-byte lines in L2 that the end-of-kernel write-back has to flush (NJ 515 -> 508 ms at 30 000 tips)
                if (kRS) { if (njp_owns<true>(a, p)) __hip_atomic_store(a.D + njp_lrow<true>(a, p) * a.ld + px, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else __hip_atomic_store(a.D + p * a.ld + px, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (i == last) {           // relabel: the node of the last slot now lives in slot y
                    new_slot = y;
                    a.slot_of_pos[p] = (int32_t)y;
                    a.pos_of_slot[y] = (int32_t)p;
                }
            } else if (i == y) {
                // (py from the winning record, not this thread's pos_of_slot[y]: the thread of the last slot rewrites that entry)
                a.Ur[py] = __builtin_nan("");   // dead: every q it takes part in is NaN, every unit minimum skips it
                Un[py] = __builtin_nan("");     // (the current buffer's entry: the log writer above; the test role of this launch leaves py out by index)
                a.slot_of_pos[py] = -1;
                Rw[py] = 0.0;
                new_slot = -1;
            } else {
                a.Ur[px] = __builtin_nan("");   // quarantine until SCAN(it + 1) has finished its row sum
                Rw[px] = 0.0;                  // diagonal
            }
            if (new_slot >= 0) { a.KA[p] = nj_key_a_dev(new_slot, n1); a.KB[p] = nj_key_b(new_slot); }
        }
        NJP_STAMP(1, 3, false);
        const double cs = block_tree256_lane0(val, s);
        if (tid == 0) a.xpart[ubi] = cs;
        NJP_STAMP(1, 6, true);
        return;
    }

    // ---------------------------------------------------------------------------------- test role
    const double NINF = -__builtin_inf(), PINF = __builtin_inf();
    // hop 2: rows x and y over this block's rows and the first strip's columns; the row buffer of the node leaving
    // quarantine where it crosses this block's units; the seed candidates' row sums
    const bool fold = pz >= 0 && pz != px && pz != py;                       // block-uniform
    const double* __restrict__ Rz = a.R + ((it + 1) & 1) * a.vstride;        // written by POST(it - 1)
    const int64_t gz = fold ? pz / kUR : -1;
    const bool gz_here = fold && gz >= (int64_t)a.blk_g0[tb] && gz < (int64_t)a.blk_g0[tb] + kTG;   // block-uniform
    const int wpz = fold ? (int)((pz % kTileCols) / (kTileCols / 4)) : -1;   // sub-strip of that node's column
    // winner-dependent part of a strip
    auto load_b = [&](int cb, ColData& c) {
        const int64_t pc0 = (int64_t)cb * kTileCols + 2 * tid;
        c.dxc = *reinterpret_cast<const v2d*>(RX(pc0));
        c.dyc = *reinterpret_cast<const v2d*>(RY(pc0));
        c.rzc.x = PINF; c.rzc.y = PINF;
        if (gz_here) c.rzc = *reinterpret_cast<const v2d*>(Rz + pc0);
    };
    // (the ~30 blocks of the strip that holds the node leaving quarantine also need ITS buffered row over their rows: fetched
    //  here, with rows x / y.  Fetched where it is used -- behind the first barrier of the strip loop -- it was one more round
    //  trip to HBM in exactly the blocks that end the launch: tests + list append ended at 4.4 - 5.0 us, theirs at 6.5.)
    constexpr bool kPreRz = kRC <= 4;                                        // (8 registers per 512 rows)
    const bool pz_block = fold && pz / kTileCols >= cb0 && pz / kTileCols < cb0 + nsb;      // block-uniform
    v2d dxr[kRC], dyr[kRC], rzr[kPreRz ? kRC : 1];
#pragma unroll
    for (int c = 0; c < kRC; ++c) {
        const int64_t pp = rbase + c * kTileCols + 2 * tid;
        const int64_t po = pp < P ? pp : pclamp;
        dxr[c] = *reinterpret_cast<const v2d*>(RX(po));
        dyr[c] = *reinterpret_cast<const v2d*>(RY(po));
        if (kPreRz) {
            rzr[c].x = 0.0; rzr[c].y = 0.0;
            if (pz_block) rzr[c] = *reinterpret_cast<const v2d*>(Rz + po);
        }
    }
#pragma unroll
    for (int k = 0; k < kNS; ++k)
        if (k < nsb) load_b(cb0 + k, cd[k]);
    // seed candidate re-evaluated with the row sums after this merge
    double qc = PINF;
    if (seed_is_unit && (int64_t)tid * sstride >= uvalid) cand.key = ~0ull;          // not written by this iteration's scan
    if (cand.key != ~0ull) {
        const int64_t ci = (int64_t)(cand.pad & 0xffffffffull), cj = (int64_t)(cand.pad >> 32);
        // the record carries D of the pair; the entry is unchanged by this merge unless one end is x or y
        if (ci < P && cj < P && ci != px && cj != px && ci != py && cj != py) {
            const double uia = Uc[ci], uib = Uc[cj];
            const double xa = *RX(ci), ya = *RY(ci), xb = *RX(cj), yb = *RY(cj);
            const double ua = nj_unew(uia, xa, ya, nj_val(xa, ya, d)) / r1;
            const double ub = nj_unew(uib, xb, yb, nj_val(xb, yb, d)) / r1;
            const double qk = fmin((cand.d - ua) - ub, (cand.d - ub) - ua);
            qc = qk == qk ? qk : qc;
        }
    }
    NJP_STAMP(1, 3, true);
    // Row sums after this merge for the row positions of the block (U, not U / (n - 3): the division is monotone, so it is
    // done once per group afterwards -- fl(max U / r) == max fl(U / r) bit for bit, and an fp64 division is ~40
    // instructions).  They go through LDS, 16 values + 1 pad per group (both the pairwise writes and the group-wise reads
    // are then conflict-free), and the test lane of a group takes the maximum of its 16.  Dead positions and the padding
    // behind P carry NaN in U; the merged pair is left out by (32-bit, block-local) index.
    __shared__ double s_un[kTG * (kUR + 1)];
    __shared__ double scm2[2][kThreads / 64], snew2[2][kThreads / 64];
    const int lxp = (int)(px - rbase), lyp = (int)(py - rbase), lzp = (int)((fold ? pz : -1) - rbase);
#pragma unroll
    for (int c = 0; c < kRC; ++c) {
        const int lp = c * kTileCols + 2 * tid;
        const double u0 = urow[c].x, u1 = urow[c].y;
        const bool live0 = (u0 == u0) & (lp != lxp) & (lp != lyp);          // (the new node is in quarantine during scan it + 1)
        const bool live1 = (u1 == u1) & (lp + 1 != lxp) & (lp + 1 != lyp);
        const double n0 = nj_unew(u0, dxr[c].x, dyr[c].x, nj_val(dxr[c].x, dyr[c].x, d));
        const double n1v = nj_unew(u1, dxr[c].y, dyr[c].y, nj_val(dxr[c].y, dyr[c].y, d));
        const int li = lp + (lp >> 4);
        s_un[li] = live0 ? n0 : NINF;
        s_un[li + 1] = live1 ? n1v : NINF;
    }
    qc = wave_fmin(qc);
    if ((tid & 63) == 0) sseed[tid >> 6] = qc;
    double bound = PINF, rmax = NINF;                               // both set behind the first strip's barrier
    const int tg = tid < kTG ? tid : 0;
    const int lane = tid & 63;
#pragma unroll
    for (int sidx = 0; sidx < kNS; ++sidx) {
        if (sidx >= nsb) break;                       // block-uniform
        const int cb = cb0 + sidx, par = sidx & 1;
        const ColData& cur = cd[sidx];
        const int64_t pc0 = (int64_t)cb * kTileCols + 2 * tid;
        const bool pz_strip = fold && pz / kTileCols == cb;                      // block-uniform
        const bool have = have_g && g >= 32 * (int64_t)cb;
        unsigned long long* up4 = a.umin + ((int64_t)cb * G16 + (have ? g : 0)) * 4;
        double u4[4] = { PINF, PINF, PINF, PINF };
        if (have) { u4[0] = dec_f64(cur.um0.x); u4[1] = dec_f64(cur.um0.y); u4[2] = dec_f64(cur.um1.x); u4[3] = dec_f64(cur.um1.y); }
        // column maximum of each sub-strip (wave w holds columns 128w .. 128w+127 of the strip); minimum of the row of the
        // node leaving quarantine over the sub-strip's live columns
        double cm_part = NINF, colmin = PINF;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t pp = pc0 + k;
            const double uo = k ? cur.ucol.y : cur.ucol.x;
            const double dx = k ? cur.dxc.y : cur.dxc.x, dy = k ? cur.dyc.y : cur.dyc.x;
            const double rz = k ? cur.rzc.y : cur.rzc.x;
            const bool live = (uo == uo) & (pp != px) & (pp != py) & (pp < P);
            const double un = nj_unew(uo, dx, dy, nj_val(dx, dy, d));      // divided after the wave maximum
            cm_part = live ? fmax(cm_part, un) : cm_part;
            colmin = (live & (pp < pz)) ? fmin(colmin, rz) : colmin;
        }
        cm_part = wave_fmax(cm_part) / r1;            // (-inf stays -inf)
        if (gz_here) colmin = wave_fmin(colmin);
        if (lane == 0) { scm2[par][tid >> 6] = cm_part; snew2[par][tid >> 6] = colmin; }
        __syncthreads();      // (one barrier per strip: the parity keeps a fast wave's next stores off the values still being read)
        NJP_STAMP(1, 4, false);
        if (sidx == 0) {
            bound = fmin(fmin(sseed[0], sseed[1]), fmin(sseed[2], sseed[3]));
#pragma unroll
            for (int k = 0; k < kUR; ++k) rmax = fmax(rmax, s_un[17 * tg + k]);      // group g = blk_g0 + tid
            rmax = rmax / r1;                                                       // (-inf stays -inf)
        }
        double newminA = PINF;
        if (pz_strip) {
            // block-uniform and rare (the strip of the node leaving quarantine): minimum of its buffered row over each
            // group's live rows behind it, same route through LDS
            __syncthreads();
#pragma unroll
            for (int c = 0; c < kRC; ++c) {
                const int lp = c * kTileCols + 2 * tid;
                const int64_t pp = rbase + lp;
                const v2d rz = kPreRz ? rzr[kPreRz ? c : 0] : *reinterpret_cast<const v2d*>(Rz + (pp < P ? pp : pclamp));
                const double u0 = urow[c].x, u1 = urow[c].y;
                const bool live0 = (u0 == u0) & (lp != lxp) & (lp != lyp) & (lp > lzp);
                const bool live1 = (u1 == u1) & (lp + 1 != lxp) & (lp + 1 != lyp) & (lp + 1 > lzp);
                const int li = lp + (lp >> 4);
                s_un[li] = live0 ? rz.x : PINF;
                s_un[li + 1] = live1 ? rz.y : PINF;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kUR; ++k) newminA = fmin(newminA, s_un[17 * tg + k]);
        }
        int submask = 0;
        if (have) {      // (a rank launches only its own test blocks)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const double cmw = scm2[par][w];
                double nm = (gz_here && g == gz) ? snew2[par][w] : PINF;      // the unit (this strip, group of pz)
                if (pz_strip && w == wpz) nm = fmin(nm, newminA);
                if (nm < u4[w]) {                          // persist the lowered bound (this lane is the unit's only writer here)
                    u4[w] = nm;
                    up4[w] = enc_f64(nm);
                }
                const double lb = fmin((u4[w] - rmax) - cmw, (u4[w] - cmw) - rmax);
                if ((cmw > NINF) && (lb <= bound)) submask |= 1 << w;
            }
        }
        const bool keep = have && (rmax > NINF) && submask != 0;
        const unsigned long long mask = __ballot(keep);
        unsigned long long base = 0;
        if (lane == 0 && mask) base = atomicAdd(&a.cnt[(it + 1) % 3], (unsigned long long)__popcll(mask));
        base = __shfl(base, 0, 64);
        if (a.dbg != nullptr && a.dbg_it == -2 && mask) {       // DPR_NJ_PHASES=-2: per iteration the largest number of (sub-)units one test block lists
            int subs = keep ? __popc(submask) : 0;
            for (int off = 32; off > 0; off >>= 1) subs += __shfl_xor(subs, off, 64);
            if (lane == 0 && it < 32768) {
                atomicMax(&a.dbg[32768 + it], ((unsigned long long)__popcll(mask) << 32) | (unsigned long long)subs);
                atomicAdd(&a.dbg[7], (unsigned long long)subs);
            }
        }
        NJP_STAMP(1, 5, true);
        if (keep) a.list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)(((uint32_t)submask << 28) | ((uint32_t)cb << 18) | (uint32_t)g);   // sub-unit mask | strip | group
    }
    NJP_STAMP(1, 6, true);
}

// ------------------------------------------------------------------------------------------------
// POST(it), LARGE shape (P >= njp_big_p(), single rank): ONE launch of light blocks.
// njp_post_kernel's test blocks each recompute the row sums after the merge for their own 4096 rows and 2048 columns (the
// test of iteration it + 1 needs the maxima of the row sums AFTER merge it, which the update blocks of the same launch are
// only just storing) and reduce all scan records: ~220 KB of L2 reads per block, 230 MB per iteration at 100 000 tips, 230
// registers per thread, two rounds of blocks -- 21.5 us per launch, the largest item of that run.
// Here nothing is recomputed per test block.  The maxima come from the PREVIOUS launch plus a proof:
//   * M blocks (one per 512 positions) of POST(it - 1) stored, per 16-row group and per 128-column sub-strip, the maximum of
//     the row sums after merge it - 1 (undivided) -- these ARE the current row sums of iteration it;
//   * merge it changes a row sum by  t_i = fl(fl(-dxi - dyi) + val_i),  val_i = fl(fl(dxi + dyi - d) / 2).  While every live
//     matrix entry is >= 0, t_i <= 0 (val_i <= fl(dxi + dyi) / 2), so fl(U_i + t_i) <= U_i: the maxima of the previous
//     launch are upper bounds of the maxima after this merge, and fl(max / (n - 3)) bounds every fl(U'_i / (n - 3)) because
//     division by a positive number is monotone.  With negative entries (NJ creates them when the triangle inequality
//     fails) the bound gets a slack T >= every t_i:  T = 1.5 |m| (1 + 2^-30) + A 2^-48  with m the smallest entry and A the
//     largest magnitude the epoch has held (the header keeps both within a factor of two, njp_note_range: the epoch build
//     reduces the matrix, the M blocks add every val they create) -- exact value -(s + d) / 2 <= 1.5 |m| for
//     s = fl(dxi + dyi) >= 2 m, d >= m, plus the rounding of three operations on magnitudes <= 3 A;
//   * the node that leaves quarantine (created by merge it - 1) has no entry in those maxima -- its row sum is stored by
//     SCAN(it) -- so its own value is added to its group and its sub-strip;
//   * its row was written by the U blocks of POST(it - 1); the M blocks of that launch stored its minimum per sub-strip (live
//     columns before it) and per group (live rows behind it) over the nodes alive THEN, a superset of those alive now: a
//     smaller minimum, still a valid lower bound for the units it crosses.
// Looser bounds list more units, never fewer: the scan computes exact minima whatever is listed, so the merge log stays the
// reference's bit for bit (the first 20 000 iterations at 100 000 tips list 1.00x as many units as the exact test).
// Roles.  T blocks (first in the grid: the longest chain) of 256 row groups x up to kNS strips, each ONE coarse cell:
// t2_cmin[tb] <= every sub-unit bound of the cell (kept by this block alone) and, by the same monotonicity,
// lb(unit) >= fl(fl(cmin - rmaxC) - cmaxC) -- a cell whose coarse bound exceeds the seed bound reads none of its 32 KB of unit
// bounds; one list append per wave.  UM blocks: block u does, for the reference slots
// [512 u, 512 u + 512), what needs the SLOT order (the chunk sums of the new node's row -- the canonical order of U[x] --, log,
// state, relabel of the last slot) and, for the positions [512 u, 512 u + 512), every per-position store of the update (row
// sums, row buffer, keys, the new node's column) plus the maxima and minima for the next launch: coalesced, where
// njp_post_kernel's update role scatters them through pos_of_slot.  No block waits for another one; 845 blocks of 108
// registers at 100 000 tips, all resident.
// Measured at 100 000 tips x 10 000 sites: 39.7 -> 24.3 us per iteration in the first epoch, NJ 2.68 -> 2.09 s on the same box,
// same merge log (profiles/r3/nj100k_post2.txt, nj100k_fused.txt, nj_kt_100k.txt; nj_phases2_100k.txt: the launch ends 11.8 us
// after its first block starts).  Steps on the way: separate U and M blocks -- 1 041 blocks for 1 024 resident ones, the
// stragglers started 8 us late; one list atomic per strip -- four dependent round trips; 16 fp64 divisions per thread for the
// 16 block-uniform column maxima -- now 16 lanes, through LDS.
// (Tried before, profiles/r3/nj_kt_100k_post2_split.txt: exact maxima handed from producer blocks to the test blocks of the
//  SAME launch through a tag -- every hand-over step is a round trip across the XCDs, 40 us per launch; as two launches --
//  14.6 + 8.8 us, each pays launch + two dependent misses + drain: 2.60 s instead of 2.68 s for the whole run.)
// ------------------------------------------------------------------------------------------------
struct Post2Hdr { unsigned long long min_enc, maxabs_enc; };      // enc_f64 order; over all entries the epoch has held

// epoch start: maxima of the row sums as they stand (buffer `par`), nothing in quarantine
__global__ __launch_bounds__(kThreads) void njp_t2_init_kernel(const double* __restrict__ Uc, int64_t P, double* __restrict__ rmaxU,
                                                               double* __restrict__ cmaxU)
{
    const int tid = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * kTileCols + 2 * tid;
    const v2d uc = *reinterpret_cast<const v2d*>(Uc + p0);
    const double NINF = -__builtin_inf();
    const double m2 = fmax(((uc.x == uc.x) & (p0 < P)) ? uc.x : NINF, ((uc.y == uc.y) & (p0 + 1 < P)) ? uc.y : NINF);
    double gm = m2;
    gm = fmax(gm, __shfl_xor(gm, 1, 64)); gm = fmax(gm, __shfl_xor(gm, 2, 64)); gm = fmax(gm, __shfl_xor(gm, 4, 64));
    const double cm = wave_fmax(m2);
    if ((tid & 7) == 0) rmaxU[p0 >> 4] = gm;
    if ((tid & 63) == 0) cmaxU[4 * blockIdx.x + (tid >> 6)] = cm;
}

template <int kNS>
__global__ __launch_bounds__(kThreads) void njp_post2_kernel(NjState* h_st, const NjRecord* h_partials, const unsigned long long* h_cnt, const int32_t* h_blk_cb,
                                                         const int32_t* h_blk_g0, const int32_t* h_pos_of_slot, int h_ntest, int h_nupd, NjpArgs a)
{
    constexpr int kTG = 256;
    __shared__ double s[kThreads];
    __shared__ double sq[kThreads / 64], sdd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64], spp[kThreads / 64];
    __shared__ double sseed[kThreads / 64], srC[kThreads / 64], scm[kNS * 4], s_colmin[kNS * 4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int bx = (int)blockIdx.x;
    // T blocks FIRST (the longest chain), then the UM blocks: block u does the slot-order part of the update for the reference
    // slots [512 u, 512 u + 512) and the position-order part for the positions [512 u, 512 u + 512) -- one select for both,
    // and 845 blocks at 100 000 tips where separate U and M blocks made 1 041 for 1 024 resident ones (110 registers): the
    // blocks of the second round started 8 us late and set the length of the launch (profiles/r3/nj_phases2_100k.txt)
    const bool trole = bx < h_ntest;
    const int tb = a.sh_rank + bx * a.sh_world;      // (unit-sharded plan: a rank launches only its own test blocks; tb is the global index)
    const int umb = bx - h_ntest;
    Post2Hdr* hdr = reinterpret_cast<Post2Hdr*>(a.t2_hdr);
    const double NINF = -__builtin_inf(), PINF = __builtin_inf();

    // hop 1: state line, scan records, and what each role can address without knowing the winner
    const int64_t it = h_st->itb;
    const int64_t limit = h_st->it_limit, N = h_st->N;
    const int32_t pn0 = h_st->pnew[0], pn1 = h_st->pnew[1];      // (both with the state line)
    const int64_t pz = (int64_t)((it & 1) ? pn1 : pn0);
    const int64_t P = a.P;
    const int64_t G16 = (P + kUR - 1) / kUR;
    const int nrec_all = a.urecs + a.nrb;
    NjRecord r0; r0.q = 10000.0; r0.key = ~0ull; r0.d = 0; r0.pad = 0;
    constexpr int kMine = 5;
    NjRecord mine[kMine] = { r0, r0, r0, r0, r0 };
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
        const int idx = tid + k * kThreads;
        if (idx < nrec_all) mine[k] = h_partials[idx];
    }
    // T: the seed candidate of this thread (as in njp_post_kernel)
    const int nseed_rows = a.nrb < kThreads / 2 ? a.nrb : kThreads / 2;
    const int nseed_units = kThreads - nseed_rows;
    const int64_t sstride = a.all_defined && a.urecs >= 2 * nseed_units ? a.urecs / nseed_units : 1;
    const bool seed_is_unit = tid < nseed_units;
    NjRecord cand = r0;
    int cb0 = 0;
    int64_t g0 = 0;
    // UM: the block's two chunks of 256 reference slots (the chunk sums stay per 256 slots: the canonical order of U[x]) and
    // its 512 positions, two per thread
    const int64_t i = (int64_t)umb * (2 * kThreads) + tid;     // first reference slot; the second one is i + 256
    const int64_t p0 = (int64_t)umb * kTileCols + 2 * tid;     // first position (< P + 512 when the M part is active)
    const bool m_part = !trole && umb < a.nrb && a.do_update;      // (a tests-only launch -- virtual ranks -- has no UM blocks at all)
    const bool u_slots = !trole && (int64_t)umb * (2 * kThreads) < N;
    int64_t p = -1, pb = -1;
    int2 sl = make_int2(-1, -1);                             // reference slots of the two positions (-1: dead / padding)
    if (trole) {
        if (!seed_is_unit) cand = h_partials[a.urecs + (tid - nseed_units)];
        else if ((int64_t)tid * sstride < a.urecs) cand = h_partials[(int64_t)tid * sstride];
        cb0 = h_blk_cb[tb];
        g0 = (int64_t)h_blk_g0[tb];
    } else {
        if (u_slots) { p = (int64_t)h_pos_of_slot[i]; pb = (int64_t)h_pos_of_slot[i + kThreads]; }      // (padded past N)
        if (m_part) sl = *reinterpret_cast<const int2*>(a.slot_of_pos + p0);
    }
    // hop 2: needs the iteration index
    const int par = (int)(it & 1);                           // T reads the maxima of buffer par, M writes buffer 1 - par
    const double* __restrict__ Uc = a.U + (it & 1) * a.vstride;
    double* __restrict__ Un = a.U + ((it + 1) & 1) * a.vstride;
    const unsigned long long cl0 = h_cnt[0], cl1 = h_cnt[1], cl2 = h_cnt[2];      // all three list counters with the first round trip (not a second, dependent load)
    const int m3 = (int)(it % 3);
    const unsigned long long cnt_raw = a.all_defined ? (unsigned long long)a.urecs : (m3 == 0 ? cl0 : m3 == 1 ? cl1 : cl2);
    const int64_t uvalid = (int64_t)(cnt_raw < (unsigned long long)a.urecs ? cnt_raw : (unsigned long long)a.urecs);
    const double emin = dec_f64(hdr->min_enc), eabs = dec_f64(hdr->maxabs_enc);
    v2d uc; uc.x = 0.0; uc.y = 0.0;
    const int64_t g = g0 + tid;                                  // (T) this lane's row group
    const bool have_g = trole && g < G16;
    int nsb = 0;
    double uz = NINF;
    double sUa = 0.0, sUb = 0.0;
    int64_t ci = -1, cj = -1;
    if (m_part) uc = *reinterpret_cast<const v2d*>(Uc + p0);
    if (trole) {
        const int64_t send = njp_strips_of_rows(g0, kTG, P);
        nsb = (int)(send - cb0 < kNS ? send - cb0 : kNS);
        if (pz >= 0) uz = Uc[pz];
        if (seed_is_unit && (int64_t)tid * sstride >= uvalid) cand.key = ~0ull;          // not written by this iteration's scan
        if (cand.key != ~0ull) {
            ci = (int64_t)(cand.pad & 0xffffffffull); cj = (int64_t)(cand.pad >> 32);
            if (!(ci < P && cj < P)) ci = -1;
        }
    }
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
        const int idx = tid + k * kThreads;
        if (idx >= uvalid && idx < a.urecs) mine[k] = r0;   // not written by this iteration's scan
    }
    NJP_STAMP(1, 0, false);
    NJP_STAMP(1, 1, true);
    if (h_st->status != 0 || it >= limit) return;
    const int64_t n = N - it;
    if (n < 3) return;
    const bool u_part = u_slots && a.do_update && (int64_t)umb * (2 * kThreads) < n;
    if (trole ? !a.do_tests : (!u_part && !m_part)) return;

    // ---- select (thrust::min_element, src/neighborJoining.cu:214)
    double bq = 10000.0, d = 0.0; uint64_t bk = ~0ull, bp = 0;
    const int nmine = (nrec_all + kThreads - 1) / kThreads;
#pragma unroll
    for (int k = 0; k < kMine; ++k)
        if (k < nmine) best_update4(bq, bk, bp, d, mine[k].q, mine[k].key, mine[k].pad, mine[k].d);
    for (int64_t idx = tid + kMine * kThreads; idx < nrec_all; idx += kThreads)
        if (idx < uvalid || idx >= a.urecs) best_update4(bq, bk, bp, d, a.partials[idx].q, a.partials[idx].key, a.partials[idx].pad, a.partials[idx].d);
    wave_best4(bq, bk, bp, d);
    if (lane == 0) { sq[tid >> 6] = bq; sk[tid >> 6] = bk; spp[tid >> 6] = bp; sdd[tid >> 6] = d; }
    __syncthreads();
    bq = sq[0]; bk = sk[0]; bp = spp[0]; d = sdd[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) best_update4(bq, bk, bp, d, sq[w], sk[w], spp[w], sdd[w]);

    NJP_STAMP(1, 2, false);
    const int64_t last = n - 1;
    if (bk == ~0ull || !(bq < 10000.0)) {
        if (u_part && (i == last || i + kThreads == last)) a.st->status = 1;
        return;
    }
    const int64_t ki = (int64_t)(bk & 0xFFFFFFull), kj = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t pi = (int64_t)(bp & 0xffffffffull), pj = (int64_t)(bp >> 32);
    const int64_t x = ki < kj ? ki : kj, y = ki < kj ? kj : ki;
    const int64_t px = ki < kj ? pi : pj, py = ki < kj ? pj : pi;
    const int64_t n1 = n - 1;
    const double r1 = (double)(n1 - 2);
    const double* __restrict__ rowx = a.D + px * a.ld;
    const double* __restrict__ rowy = a.D + py * a.ld;
    const int64_t GS = 32 * ((P + kTileCols - 1) / kTileCols + 2), SS4 = 4 * ((P + kTileCols - 1) / kTileCols + 2);

    if (!trole) {
        // hop 3, both parts: rows x / y at this thread's two slots' positions (a gather) and at its two positions (coalesced)
        double gx[2] = { 0.0, 0.0 }, gy[2] = { 0.0, 0.0 };
        v2d dx, dy; dx.x = dx.y = dy.x = dy.y = 0.0;
        if (u_part) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t ii = i + h * kThreads, pp = h ? pb : p;
                if (ii < n && ii != x && ii != y) { gx[h] = rowx[pp]; gy[h] = rowy[pp]; }
            }
        }
        if (m_part) { dx = *reinterpret_cast<const v2d*>(rowx + p0); dy = *reinterpret_cast<const v2d*>(rowy + p0); }
        if (m_part) {
            // ------------------------------------------------------------------ M: the update by POSITION, and the maxima for POST(it + 1)
            const bool live0 = (uc.x == uc.x) & (p0 != px) & (p0 != py) & (p0 < P);
            const bool live1 = (uc.y == uc.y) & (p0 + 1 != px) & (p0 + 1 != py) & (p0 + 1 < P);
            const double v0 = nj_val(dx.x, dy.x, d), v1 = nj_val(dx.y, dy.y, d);       // the new node's row
            const double un0 = nj_unew(uc.x, dx.x, dy.x, v0);
            const double un1 = nj_unew(uc.y, dx.y, dy.y, v1);
            NJP_STAMP(2, 4, false);           // (finer stamps of this kernel live in group 2: rows x / y have arrived)
            if (a.do_update) {
                // same values, same destinations as njp_post_kernel's update role (which reaches them through pos_of_slot): new row
                // sums into the other buffer, the new node's row into the row buffer and its column into the matrix, keys from the
                // slot (the node of the last slot is relabelled to slot y), quarantine / death marks of the merged pair
                double* __restrict__ Rw = a.R + (it & 1) * a.vstride;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int64_t pp = p0 + k;
                    const int slot = k ? sl.y : sl.x;
                    const double v = k ? v1 : v0, u = k ? un1 : un0;
                    if (pp >= P) continue;
                    if (pp == px) {
                        a.Ur[px] = __builtin_nan("");   // quarantine until SCAN(it + 1) has finished its row sum
                        Rw[px] = 0.0;                  // diagonal
                        a.KA[px] = nj_key_a_dev(x, n1); a.KB[px] = nj_key_b(x);
                    } else if (pp == py) {
                        a.Ur[py] = __builtin_nan("");   // dead
                        Un[py] = __builtin_nan("");
                        Rw[py] = 0.0;
                    } else if (slot >= 0) {
                        Un[pp] = u;
                        a.Ur[pp] = u / r1;
                        Rw[pp] = v;
                        __hip_atomic_store(a.D + pp * a.ld + px, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // (slot_of_pos of the last slot's position is being rewritten to y by the U part of some block: either value gives y here)
                        const int64_t new_slot = slot == (int)last ? y : (int64_t)slot;
                        a.KA[pp] = nj_key_a_dev(new_slot, n1); a.KB[pp] = nj_key_b(new_slot);
                    }
                }
            }
            NJP_STAMP(2, 5, false);           // (the position-order stores are issued)
            const double m2 = fmax(live0 ? un0 : NINF, live1 ? un1 : NINF);
            double gm = m2;                                                   // group of 16 positions = 8 lanes
            gm = fmax(gm, __shfl_xor(gm, 1, 64)); gm = fmax(gm, __shfl_xor(gm, 2, 64)); gm = fmax(gm, __shfl_xor(gm, 4, 64));
            const double cm = wave_fmax(m2);                                  // sub-strip of 128 positions = this wave
            const int64_t gidx = p0 >> 4;
            const int wq = 1 - par;
            if ((tid & 7) == 0) a.t2_rmax[wq * GS + gidx] = gm;                 // undivided
            if (lane == 0) a.t2_cmax[wq * SS4 + 4 * umb + (tid >> 6)] = cm;
            // the new node's row (position px): minimum per sub-strip over the live columns before it, per group over the live
            // rows behind it
            const double c2 = fmin((live0 & (p0 < px)) ? v0 : PINF, (live1 & (p0 + 1 < px)) ? v1 : PINF);
            double r2 = fmin((live0 & (p0 > px)) ? v0 : PINF, (live1 & (p0 + 1 > px)) ? v1 : PINF);
            r2 = fmin(r2, __shfl_xor(r2, 1, 64)); r2 = fmin(r2, __shfl_xor(r2, 2, 64)); r2 = fmin(r2, __shfl_xor(r2, 4, 64));
            const double cz = wave_fmin(c2);
            if ((tid & 7) == 0) a.t2_rowmin[wq * GS + gidx] = r2;
            if (lane == 0) a.t2_colmin[wq * SS4 + 4 * umb + (tid >> 6)] = cz;
            // smallest entry / largest magnitude the epoch has held: this block's new values
            double vmin = fmin(live0 ? v0 : PINF, live1 ? v1 : PINF);
            double vabs = fmax(live0 ? fabs(v0) : 0.0, live1 ? fabs(v1) : 0.0);
            vmin = wave_fmin(vmin); vabs = wave_fmax(vabs);
            if (lane == 0) njp_note_range(&hdr->min_enc, vmin, vabs);
        }
        NJP_STAMP(1, 3, true);
        if (u_part) {
            // ------------------------------------------------------------------ U: what needs the SLOT order
            // the chunk sums of the new node's row (canonical order of U[x]), the log, the state, the relabel of the last slot
            double val2[2] = { 0.0, 0.0 };
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t ii = i + h * kThreads;
                const int64_t pp = h ? pb : p;
                if (ii >= n) continue;
                if (ii == last) {
                    const double r = (double)(n - 2);
                    double blX = (d + Uc[px] / r - Uc[py] / r) * 0.5;
                    double blY = d - blX;
                    if (blX < 0) { blY += blX; blX = 0; }
                    if (blY < 0) { blX += blY; blY = 0; }
                    a.log_x[it] = (int32_t)x; a.log_y[it] = (int32_t)y; a.log_bx[it] = blX; a.log_by[it] = blY;
                    a.st->x = (int32_t)x; a.st->y = (int32_t)y; a.st->d = d; a.st->q = bq;
                    a.st->n = n1; a.st->it = it + 1;
                    a.st->pnew[(it + 1) & 1] = (int32_t)px;
                    a.U[(it & 1) * a.vstride + py] = __builtin_nan("");       // (this thread's own store, behind its read of U[py] above)
                    a.st->cnt_list[(it + 2) % 3] = 0ull;
                    for (int v = 0; v < a.cnt_ranks; ++v) a.cnt_all[4 * v + (it + 2) % 3] = 0ull;
                }
                if (ii != x && ii != y) {
                    val2[h] = nj_val(gx[h], gy[h], d);
                    if (ii == last) {           // relabel: the node of the last slot now lives in slot y
                        a.slot_of_pos[pp] = (int32_t)y;
                        a.pos_of_slot[y] = (int32_t)pp;
                    }
                } else if (ii == y) {
                    a.slot_of_pos[py] = -1;
                }
            }
            const double cs0 = block_tree256_lane0(val2[0], s);
            if (tid == 0) a.xpart[2 * umb] = cs0;
            if ((int64_t)(2 * umb + 1) * kThreads < n) {      // block-uniform: the second chunk holds live slots
                __syncthreads();
                const double cs1 = block_tree256_lane0(val2[1], s);
                if (tid == 0) a.xpart[2 * umb + 1] = cs1;
            }
        }
        NJP_STAMP(1, 6, true);
        if (a.dbg != nullptr && it == a.dbg_it && tid == 0 && bx < 2048) a.dbg[(2048 + bx) * 8 + 7] = 1ull;
        return;
    }

    // ---------------------------------------------------------------------------------- T: tests of iteration it + 1
    // hop 3: the maxima of the previous launch (issued here, not with hop 2: the scan records are dead by now and their
    // 40 registers free -- the chain has this hop anyway) and the seed candidates re-evaluated with the row sums after this merge
    const double rmaxU = have_g ? a.t2_rmax[par * GS + g] : NINF;
    double cmU = NINF;                                                 // lanes 0 .. 4 kNS - 1: one sub-strip of the cell each
    if (tid < 4 * kNS && (tid >> 2) < nsb) cmU = a.t2_cmax[par * SS4 + 4 * (cb0 + (tid >> 2)) + (tid & 3)];
    const double cmin = a.t2_cmin[tb];
    // Seed bound.  Any live pair gives a valid upper bound of the next optimum, so not every record has to be re-evaluated:
    // of each group of 16 lanes only the candidate with the smallest OLD q (pairs touching x or y left out) gathers its six
    // values (two row sums, rows x / y at both ends) and is re-evaluated -- 16 candidates per block instead of 256.  Round 4:
    // every block gathering the same 256 x 6 scattered words was ~70 % of the L2 requests of a post launch (673 k per launch
    // at 30 000 tips) and what stretched its second round trip from 1.1 to 3.5 us; the iteration's minimum by old q is in
    // the set by construction and a merge moves a q by ~1 / n of its size, so the bound hardly ever loosens
    // (one candidate per wave, per 8 lanes, all of them: measured equal or slower, NOTES.md round 4 item 4).
    double qc = PINF;
    {
        const bool live = ci >= 0 && ci != px && cj != px && ci != py && cj != py;
        const double oq = live ? cand.q : PINF;
        constexpr int sgm = 0;      // (round 4: one candidate per wave / per 8 lanes / all of them measured equal or slower)
        bool lead = live;
        if (sgm != 3) {
            double gmin = oq;
            gmin = fmin(gmin, dpp_f64<kDppXor1>(gmin));
            gmin = fmin(gmin, dpp_f64<kDppXor2>(gmin));
            gmin = fmin(gmin, dpp_f64<kDppHalfMirror>(gmin));
            if (sgm != 2) gmin = fmin(gmin, dpp_f64<kDppMirror>(gmin));
            if (sgm == 1) gmin = fmin(fmin(readlane_f64(gmin, 0), readlane_f64(gmin, 16)), fmin(readlane_f64(gmin, 32), readlane_f64(gmin, 48)));
            const int sgl = sgm == 1 ? 64 : sgm == 2 ? 8 : 16;                   // lanes per group
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(live & (oq == gmin));
            const int gbase = lane & ~(sgl - 1);
            const unsigned long long mine_g = (hit >> gbase) & (sgl == 64 ? ~0ull : ((1ull << sgl) - 1ull));
            lead = live && mine_g != 0ull && (lane - gbase) == (int)__builtin_ctzll(mine_g);
        }
        if (lead) {
            sUa = Uc[ci]; sUb = Uc[cj];
            const double xa = rowx[ci], ya = rowy[ci], xb = rowx[cj], yb = rowy[cj];
            const double ua = nj_unew(sUa, xa, ya, nj_val(xa, ya, d)) / r1;
            const double ub = nj_unew(sUb, xb, yb, nj_val(xb, yb, d)) / r1;
            const double qk = fmin((cand.d - ua) - ub, (cand.d - ub) - ua);
            qc = qk == qk ? qk : qc;
        }
    }
    // small shape (one strip per block): the unit bounds travel with this round trip -- waiting for the coarse decision first
    // is a dependent round trip, and a block holds 8 KB of them
    ulonglong2 um0[kNS], um1[kNS];
#pragma unroll
    for (int k = 0; k < kNS; ++k) { um0[k] = make_ulonglong2(0ull, 0ull); um1[k] = um0[k]; }
    if (kNS == 1 && nsb > 0 && have_g && g >= 32 * (int64_t)cb0) {
        const unsigned long long* up = a.umin + ((int64_t)cb0 * G16 + g) * 4;
        um0[0] = *reinterpret_cast<const ulonglong2*>(up); um1[0] = *reinterpret_cast<const ulonglong2*>(up + 2);
    }
    qc = wave_fmin(qc);
    if (lane == 0) sseed[tid >> 6] = qc;
    NJP_STAMP(1, 3, true);
    // upper bounds of the maxima after this merge (see the header)
    const double slack = emin >= 0.0 ? 0.0 : (3.0 * -emin) * (1.0 + 0x1p-30) + eabs * 0x1p-47;      // (the header's values are within a factor of two)
    const bool zlive = pz >= 0 && pz != px && pz != py;                        // the node leaving quarantine stays
    uz = (uz == uz) ? uz : PINF;                                               // (its row sum is always in memory here; NaN would silently drop out of fmax)
    const int64_t gz = zlive ? pz / kUR : -1;
    const int64_t sz = zlive ? pz / (kTileCols / 4) : -1;                      // its sub-strip (global index)
    double rU = rmaxU;
    if (have_g && g == gz) rU = fmax(rU, uz);
    const double rmax = (rU + slack) / r1;                                     // (-inf stays -inf)
    if (tid < 4 * kNS) {
        double cU = cmU;
        if (4 * (int64_t)(cb0 + (tid >> 2)) + (tid & 3) == sz) cU = fmax(cU, uz);
        scm[tid] = (cU + slack) / r1;
    }
    const bool fold = zlive;                                                   // block-uniform
    const bool gz_here = fold && gz >= g0 && gz < g0 + kTG;
    const int wpz = fold ? (int)((pz % kTileCols) / (kTileCols / 4)) : -1;
    const int64_t cbz = fold ? pz / kTileCols : -1;
    const bool pz_here = fold && cbz >= cb0 && cbz < cb0 + nsb;              // block-uniform
    double rC = wave_fmax(rmax);
    if (lane == 0) srC[tid >> 6] = rC;
    __syncthreads();
    const double bound = fmin(fmin(sseed[0], sseed[1]), fmin(sseed[2], sseed[3]));
    double cm4[kNS][4];
#pragma unroll
    for (int k = 0; k < kNS; ++k)
#pragma unroll
        for (int w = 0; w < 4; ++w) cm4[k][w] = scm[4 * k + w];
    {
        rC = fmax(fmax(srC[0], srC[1]), fmax(srC[2], srC[3]));
        double cC = NINF;
#pragma unroll
        for (int k = 0; k < kNS; ++k)
#pragma unroll
            for (int w = 0; w < 4; ++w) cC = fmax(cC, cm4[k][w]);
        const double lbC = fmin((cmin - rC) - cC, (cmin - cC) - rC);
        // the node leaving quarantine lowers unit bounds of this cell: those lanes must run (block-uniform decision)
        if (!gz_here && !pz_here && !(lbC <= bound)) {
            NJP_STAMP(1, 6, false);
            if (a.dbg != nullptr && it == a.dbg_it && tid == 0) a.dbg[(2048 + bx) * 8 + 7] = 3ull;
            return;
        }
        NJP_STAMP(1, 4, false);
    }
    double colz = PINF;            // (issued with the unit bounds; see below)
    if (gz_here && tid < 4 * kNS && (tid >> 2) < nsb) colz = a.t2_colmin[par * SS4 + 4 * (cb0 + (tid >> 2)) + (tid & 3)];
    if (kNS > 1) {
#pragma unroll
        for (int k = 0; k < kNS; ++k) {
            if (k < nsb && have_g && g >= 32 * (int64_t)(cb0 + k)) {
                const unsigned long long* up = a.umin + ((int64_t)(cb0 + k) * G16 + g) * 4;
                um0[k] = *reinterpret_cast<const ulonglong2*>(up); um1[k] = *reinterpret_cast<const ulonglong2*>(up + 2);
            }
        }
    }
    NJP_STAMP(1, 5, true);
    if (gz_here) {                 // block-uniform: the ~20 blocks whose row groups hold the node leaving quarantine
        // its column minima for this block's 4 x kNS sub-strips in ONE round trip: loaded where they are used -- one lane, inside
        // the loop below, each behind a wait and in front of a conditional store -- they were sixteen round trips in a row in
        // exactly the blocks that end the launch (ISA: load, s_waitcnt vmcnt(0), store, load, ...)
        if (tid < 4 * kNS) s_colmin[tid] = colz;
        __syncthreads();
    }
    double mymin = PINF;           // minimum of this lane's sub-unit bounds after the fold: the cell's new coarse bound
    int sub[kNS];                  // per strip: sub-unit mask of this lane's unit, 0 = not listed
#pragma unroll
    for (int sidx = 0; sidx < kNS; ++sidx) {
        sub[sidx] = 0;
        if (sidx >= nsb) continue;                    // block-uniform
        const int cb = cb0 + sidx;
        const bool pz_strip = fold && cbz == cb;                                 // block-uniform
        const bool have = have_g && g >= 32 * (int64_t)cb;
        unsigned long long* up4 = a.umin + ((int64_t)cb * G16 + (have ? g : 0)) * 4;
        double u4[4] = { PINF, PINF, PINF, PINF };
        if (have) { u4[0] = dec_f64(um0[sidx].x); u4[1] = dec_f64(um0[sidx].y); u4[2] = dec_f64(um1[sidx].x); u4[3] = dec_f64(um1[sidx].y); }
        const double newminA = (pz_strip && have_g) ? a.t2_rowmin[par * GS + g] : PINF;
        int submask = 0;
        if (have) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const double cmw = cm4[sidx][w];
                double nm = (gz_here && g == gz) ? s_colmin[4 * sidx + w] : PINF;      // the unit (this strip, group of pz)
                if (pz_strip && w == wpz) nm = fmin(nm, newminA);
                if (nm < u4[w]) {                          // persist the lowered bound (this lane is the unit's only writer here)
                    u4[w] = nm;
                    up4[w] = enc_f64(nm);
                }
                mymin = fmin(mymin, u4[w]);
                const double lb = fmin((u4[w] - rmax) - cmw, (u4[w] - cmw) - rmax);
                if ((cmw > NINF) && (lb <= bound)) submask |= 1 << w;
            }
        }
        if (have && (rmax > NINF)) sub[sidx] = submask;
    }
    // (finer stamps, group 2.  A stamp is a store, and the next wait for "all memory operations" includes it: ~0.5 us each --
    //  read differences between stamps with that in mind; two more stamps inside the loop above made every strip "cost" 0.6 us)
    NJP_STAMP(2, 0, false);                   // (the unit tests are done)
    // ONE list append per wave for all its strips (an atomic per strip is a chain of kNS dependent round trips)
    {
        unsigned long long masks[kNS];
        int total = 0;
#pragma unroll
        for (int sidx = 0; sidx < kNS; ++sidx) { masks[sidx] = __ballot(sub[sidx] != 0); total += __popcll(masks[sidx]); }
        if (total > 0) {                              // wave-uniform
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&a.cnt[(it + 1) % 3], (unsigned long long)total);
            base = __shfl(base, 0, 64);
            NJP_STAMP(2, 1, false);           // (wave 0 only, if it lists: the atomic has returned)
#pragma unroll
            for (int sidx = 0; sidx < kNS; ++sidx) {
                if (sub[sidx] != 0)
                    a.list[base + __popcll(masks[sidx] & ((1ull << lane) - 1ull))] =
                        (int32_t)(((uint32_t)sub[sidx] << 28) | ((uint32_t)(cb0 + sidx) << 18) | (uint32_t)g);
                base += (unsigned long long)__popcll(masks[sidx]);
            }
        }
    }
    NJP_STAMP(2, 2, false);                   // (the list stores are issued)
    mymin = wave_fmin(mymin);
    __syncthreads();               // (srC: the coarse maxima above were read by every thread)
    if (lane == 0) srC[tid >> 6] = mymin;
    __syncthreads();
    NJP_STAMP(2, 3, false);                   // (every wave of the block has got here)
    if (tid == 0) a.t2_cmin[tb] = fmin(fmin(srC[0], srC[1]), fmin(srC[2], srC[3]));
    NJP_STAMP(1, 6, true);
    if (a.dbg != nullptr && it == a.dbg_it && tid == 0) a.dbg[(2048 + bx) * 8 + 7] = 4ull;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int64_t round_up16(int64_t v) { return (v + 15) / 16 * 16; }
static bool njp_use_post2(const NjPruned& q);
// (the scan grid, the graph length and the debug buffer are per-context state of NjPruned: two contexts of one process
// may run different plans, from different host threads)
// the stamps of the last context that ran with DPR_NJ_PHASES (debug hook njp_phase_stamps; a process-wide pointer to a
// per-context buffer that stays allocated until the process ends)
static unsigned long long* g_njp_dbg_last = nullptr;

// ---- arena -------------------------------------------------------------------------------------------------------
// Everything the pruned path needs is allocated once per (tips, local ranks) and kept until nj_free: hipMalloc /
// hipFree of the 7.2 GB matrices (and of ~15 vectors per epoch, 8 epochs per run) serialise with the device and
// cost more than the distance kernel when a context builds its matrix again (bench.py's steps).
struct SlabPlan {
    size_t U, R, Ur, KA, KB, slot_of_pos, pos_of_slot, perm, umin, list, blk_cb, blk_g0, cnt_all, t2_hdr, t2_rmax, t2_cmax, t2_colmin, t2_rowmin, t2_cmin, total;
    int64_t list_stride;
};
static size_t align256(size_t v) { return (v + 255) / 256 * 256; }
static int64_t prep_blocks(int64_t P, std::vector<int32_t>* hcb, std::vector<int32_t>* hg0)
{
    const int64_t G16 = (P + kUR - 1) / kUR;
    int64_t cnt = 0;
    if (njp_ns(P) == 1) {
        for (int64_t c = 0; 32 * c < G16 && c * kTileCols < P - 1; ++c)
            for (int64_t g0 = 32 * c; g0 < G16; g0 += njp_tg(P)) {
                if (hcb) { hcb->push_back((int32_t)c); hg0->push_back((int32_t)g0); }
                ++cnt;
            }
    } else {
        for (int64_t g0 = 0; g0 < G16; g0 += njp_tg(P))
            for (int64_t c0 = 0; c0 < njp_strips_of_rows(g0, njp_tg(P), P); c0 += njp_ns(P)) {
                if (hcb) { hcb->push_back((int32_t)c0); hg0->push_back((int32_t)g0); }
                ++cnt;
            }
    }
    if (cnt == 0) { if (hcb) { hcb->push_back(0); hg0->push_back(0); } cnt = 1; }
    return cnt;
}
static int64_t vec_len(int64_t N) { return (N + kTileCols + 16 + 31) / 32 * 32; }    // 256-byte multiple per vector
static SlabPlan slab_plan(int64_t P, int64_t N, int local_ranks)
{
    SlabPlan p;
    const size_t vec = (size_t)vec_len(N);
    const int64_t G16 = (P + kUR - 1) / kUR, S = (P + kTileCols - 1) / kTileCols + 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    p.U = take(2 * vec * 8); p.R = take(2 * vec * 8);
    p.Ur = take(vec * 8); p.KA = take(vec * 8); p.KB = take(vec * 8);
    p.slot_of_pos = take(vec * 4); p.pos_of_slot = take(vec * 4); p.perm = take(vec * 4);
    p.umin = take((size_t)(S * G16 * 4) * 8);
    p.list_stride = unit_total(P) + kScanBlocks + 64;
    p.list = take((size_t)(p.list_stride * local_ranks) * 4);
    // (capacity for the finest test-block size: a later, smaller epoch of the same arena may use it)
    size_t nprep = 1;
    {
        const int64_t G16c = (P + kUR - 1) / kUR;
        for (int64_t c = 0; 32 * c < G16c && c * kTileCols < P - 1; ++c) nprep += (size_t)((G16c - 32 * c + 31) / 32);
    }
    p.blk_cb = take(nprep * 4); p.blk_g0 = take(nprep * 4);
    p.cnt_all = take((size_t)(4 * local_ranks) * 8);
    // njp_post2_kernel: header, per-group and per-sub-strip values (M block m writes groups 32 m .. 32 m + 31, sub-strips 4 m .. 4 m + 3)
    p.t2_hdr = take(256);
    p.t2_rmax = take((size_t)(2 * 32 * (S + 1)) * 8); p.t2_rowmin = take((size_t)(2 * 32 * (S + 1)) * 8);       // [2]: by iteration parity
    p.t2_cmax = take((size_t)(2 * 4 * (S + 1)) * 8); p.t2_colmin = take((size_t)(2 * 4 * (S + 1)) * 8);
    p.t2_cmin = take(nprep * 8);       // one coarse bound per test block
    p.total = off;
    return p;
}
// (the same size as NjBuffers::D for N tips, nj_alloc: the two matrix buffers of a context are interchangeable -- the
//  hand-over to the streaming loop swaps them when the epoch of the moment lives in NjBuffers::D)
static size_t matrix_bytes(int64_t N)
{
    const int64_t rows_alloc = (N + kRowBlock - 1) / kRowBlock * kRowBlock + 32;
    return (size_t)(rows_alloc * round_up16(N) + kTileCols + 16) * sizeof(double);
}

static int njp_arena(NjPruned& q, int64_t N, hipStream_t s)
{
    const int local_ranks = q.sh_world > 1 && q.sh_virtual ? q.sh_world : 1;
    const SlabPlan plan = slab_plan(N, N, local_ranks);
    if (q.arena_D && q.arena_N == N && q.arena_ranks >= local_ranks && q.arena_slab_bytes >= plan.total) return DPR_OK;
    void* old[] = { q.arena_D, q.arena_slab[0], q.arena_slab[1] };
    for (void* p : old)
        if (p) (void)hipFree(p);
    q.arena_D = nullptr; q.arena_slab[0] = q.arena_slab[1] = nullptr;
    DPR_HIP(hipMalloc(&q.arena_D, matrix_bytes(N)));
    DPR_HIP(hipMemsetAsync(q.arena_D, 0, matrix_bytes(N), s));
    for (int k = 0; k < 2; ++k) DPR_HIP(hipMalloc(&q.arena_slab[k], plan.total));
    q.arena_slab_bytes = plan.total;
    q.arena_N = N;
    q.arena_ranks = local_ranks;
    return DPR_OK;
}

int njp_reserve(NjPruned& q, int64_t N, hipStream_t s) { return njp_arena(q, N, s); }

// point q at the position-space structures of an epoch with P positions (N = total tips: slot arrays) inside
// matrix buffer `Dbuf` and slab `slab`, and initialise them (all fills ordered on s)
// rs_world > 1 (row-sharded mode, njr.hip): Dbuf holds the chunks of rank rs_rank only (the caller has cleared it), and the
// test blocks are this rank's -- one strip x 64 row groups ALIGNED to the ownership chunks (a strip's first block may start
// up to 32 groups in front of the strip's first valid group: those lanes are masked by g >= 32 cb in the kernels)
static int njp_alloc_epoch(NjPruned& q, int64_t P, int64_t N, double* Dbuf, char* slab, hipStream_t s, const void* hdr_from = nullptr,
                           int rs_world = 1, int rs_rank = 0)
{
    if (P >= (int64_t)kTileCols * 1024) { set_error("pruned NJ: the list encoding holds fewer than 524288 positions"); return DPR_ERR_ARG; }
    const int local_ranks = q.sh_world > 1 && q.sh_virtual ? q.sh_world : 1;
    const SlabPlan plan = slab_plan(P, N, local_ranks);
    q.P = P;
    q.ld = round_up16(P);
    q.D = Dbuf;
    const int64_t rows_alloc = (P + kUR - 1) / kUR * kUR + kUR;
    // what the permute kernel does not write: columns [P, ld), the group of rows behind position P, the tail pad
    if (rs_world <= 1)
        if (int rc = nj_fill_pads(q.D, q.ld, P, P, rows_alloc, kTileCols + 16, false, s)) return rc;
    const size_t vec = (size_t)vec_len(N);
    q.vstride = (int64_t)vec;
    q.U = reinterpret_cast<double*>(slab + plan.U);
    q.R = reinterpret_cast<double*>(slab + plan.R);
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
    q.t2_hdr = slab + plan.t2_hdr;
    q.t2_rmax = reinterpret_cast<double*>(slab + plan.t2_rmax); q.t2_rowmin = reinterpret_cast<double*>(slab + plan.t2_rowmin);
    q.t2_cmax = reinterpret_cast<double*>(slab + plan.t2_cmax); q.t2_colmin = reinterpret_cast<double*>(slab + plan.t2_colmin);
    q.t2_cmin = reinterpret_cast<double*>(slab + plan.t2_cmin);
    if (hdr_from != nullptr) {
        DPR_HIP(hipMemcpyAsync(q.t2_hdr, hdr_from, 16, hipMemcpyDeviceToDevice, s));      // the range of the run so far
    } else {
        const unsigned long long h0[2] = { 0xFFF0000000000000ull /* enc(+inf) */, 0x8000000000000000ull /* enc(0.0) */ };
        DPR_HIP(hipMemcpyAsync(q.t2_hdr, h0, sizeof h0, hipMemcpyHostToDevice, s));
        DPR_HIP(hipStreamSynchronize(s));
    }
    DPR_HIP(hipMemsetAsync(q.U, 0xff, 2 * vec * sizeof(double), s));   // NaN = dead / padding, in both buffers
    DPR_HIP(hipMemsetAsync(q.R, 0, 2 * vec * sizeof(double), s));
    DPR_HIP(hipMemsetAsync(q.Ur, 0xff, vec * sizeof(double), s));   // NaN beyond P
    DPR_HIP(hipMemsetAsync(q.KA, 0, vec * sizeof(uint64_t), s));
    DPR_HIP(hipMemsetAsync(q.KB, 0, vec * sizeof(uint64_t), s));
    DPR_HIP(hipMemsetAsync(q.slot_of_pos, 0xff, sizeof(int32_t) * vec, s));   // -1: dead / padding
    DPR_HIP(hipMemsetAsync(q.pos_of_slot, 0xff, sizeof(int32_t) * vec, s));   // -1: slot not alive
    const int64_t G16 = (P + kUR - 1) / kUR, S = (P + kTileCols - 1) / kTileCols + 1;
    q.nunits_alloc = S * G16 * 4;      // four sub-strip bounds per unit
    q.utot = unit_total(P);
    {
        // test blocks: one strip and up to 256 consecutive row groups each (groups >= 32*cb see the strip)
        std::vector<int32_t> hcb, hg0;
        if (rs_world > 1) {
            const int64_t G16r = (P + kUR - 1) / kUR, gpc = kNjrChunk / kUR;      // row groups per ownership chunk (64)
            for (int64_t c = 0; 32 * c < G16r && c * kTileCols < P - 1; ++c)
                for (int64_t g0 = (32 * c) / gpc * gpc; g0 < G16r; g0 += gpc)
                    if (njr_owner(g0 * kUR, rs_world) == rs_rank) { hcb.push_back((int32_t)c); hg0.push_back((int32_t)g0); }
            q.nprep = (int)hcb.size();
            if (hcb.empty()) { hcb.push_back(0); hg0.push_back(0); }      // (a rank without units: nprep = 0, nothing is launched for it)
        } else
        q.nprep = (int)prep_blocks(P, &hcb, &hg0);
        DPR_HIP(hipMemcpyAsync(q.blk_cb, hcb.data(), sizeof(int32_t) * hcb.size(), hipMemcpyHostToDevice, s));
        DPR_HIP(hipMemcpyAsync(q.blk_g0, hg0.data(), sizeof(int32_t) * hg0.size(), hipMemcpyHostToDevice, s));
        DPR_HIP(hipStreamSynchronize(s));   // the host vectors go out of scope
    }
    // unit-sharded mode: one list and one counter quadruple per rank held here (all of them for virtual ranks)
    q.list_stride = plan.list_stride;
    DPR_HIP(hipMemsetAsync(q.list, 0, sizeof(int32_t) * (size_t)(q.list_stride * local_ranks), s));
    DPR_HIP(hipMemsetAsync(q.cnt_all, 0, sizeof(unsigned long long) * (size_t)(4 * local_ranks), s));
    hipLaunchKernelGGL(njp_fill_u64_kernel, dim3(256), dim3(256), 0, s, (uint64_t*)q.umin, q.nunits_alloc,
                       enc_f64_host(-__builtin_inf()));
    hipLaunchKernelGGL(njp_fill_u64_kernel, dim3(16), dim3(256), 0, s, (uint64_t*)q.t2_cmin, (int64_t)q.nprep, 0xFFF0000000000000ull);   // -inf (plain doubles)
    DPR_HIP(hipGetLastError());
    q.fresh = true;
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
        // default: 256 blocks while an iteration lists ~100 units (a 1000-block grid takes ~1.5 us just to start; NJ 515 ->
        // 512 ms at 30 000 tips), 512 above (round 2: 1024; every block of the post kernels reduces one record per scan block --
        // NJ at 100 000 tips 2.10 / 2.07 / 2.06 / 2.08 / 2.10 s with 256 / 384 / 512 / 768 / 1024, round 3)
        // Round 4: 512 for every size.  The listing rate depends on the data and on the age of the epoch (4 - 60 units per
        // iteration in a fresh epoch, 150 - 350 in an old one at 30 000 tips, 770 - 1 180 at 100 000; branch lengths x 5 / x 25:
        // 500 / 820 on average): with 256 blocks the diverged inputs took 574 / 663 ms, with 512 blocks 544 / 614 ms, the bench
        // input 479.5 vs 481.5 ms.  A grid that followed the watched rate (256 / 512 / 1 024, graph re-captured) was slower than
        // 512 throughout: 2.08 vs 2.05 s at 100 000 tips, 657 vs 614 ms on the x 25 input (profiles/r4/scan_grid_*.txt).
        b.pr.scan_grid = njp_scan_grid_default();
    }
    if (const char* e = std::getenv("DPR_NJ_ADAPTIVE")) b.pr.adaptive = std::atoi(e) != 0 ? 1 : 0;
    if (const char* e = std::getenv("DPR_NJ_STREAM_FRAC")) b.pr.stream_frac = std::atof(e);
    b.pr.stream_iterations = 0; b.pr.stream_epochs = 0;
    if (const char* e = std::getenv("DPR_NJ_GRAPH_ITERS")) { const int v = std::atoi(e); if (v >= 1 && v <= 4096) b.pr.graph_iters = v; }
    if (const char* e = std::getenv("DPR_NJ_PHASES")) {
        b.pr.dbg_it = std::atoll(e);
        if (!b.pr.dbg) DPR_HIP(hipMalloc(&b.pr.dbg, sizeof(unsigned long long) * 4 * 2048 * 8));
        DPR_HIP(hipMemsetAsync(b.pr.dbg, 0, sizeof(unsigned long long) * 4 * 2048 * 8, s));
        g_njp_dbg_last = b.pr.dbg;
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
    njp_launch_permute(b.D, b.ld, q.D, q.ld, q.perm, N, s);
    q.range_known = false;
    if (njp_use_post2(q)) {        // the large-shape post kernel's bounds need the range of the entries (one pass, once per run)
        hipLaunchKernelGGL(njp_range_kernel, dim3(2048), dim3(kThreads), 0, s, (const double*)b.D, b.ld, N, (unsigned long long*)q.t2_hdr);
        q.range_known = true;
    }
    // (iteration 0 reads U buffer 0)
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
// one (bounds start at -inf, no seed).  Called between iterations with the stream idle and the node in
// quarantine materialised (njp_finish_kernel).
// *rebuilt = false: nothing was done (no candidate left, or fewer than three active nodes).
static int njp_rebuild_epoch(NjBuffers& b, hipStream_t s, bool* rebuilt)
{
    NjPruned& q = b.pr;
    *rebuilt = false;
    const bool tlog = log_level("epoch") >= 3;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!tlog) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[njp]   rebuild: %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - tprev).count());
        tprev = t;
    };
    NjState st;
    DPR_HIP(hipMemcpy(&st, b.st, sizeof(NjState), hipMemcpyDeviceToHost));
    const int64_t n = st.n, Pold = q.P;
    if (st.status != 0 || n < 3) return DPR_OK;
    std::vector<double> hU((size_t)Pold);
    std::vector<int32_t> hslot((size_t)Pold);
    const double* Ucur = q.U + (st.it & 1) * q.vstride;
    DPR_HIP(hipMemcpy(hU.data(), Ucur, sizeof(double) * (size_t)Pold, hipMemcpyDeviceToHost));
    DPR_HIP(hipMemcpy(hslot.data(), q.slot_of_pos, sizeof(int32_t) * (size_t)Pold, hipMemcpyDeviceToHost));
    lap("state + vectors to the host");
    std::vector<int32_t> perm;
    perm.reserve((size_t)n);
    for (int64_t p = 0; p < Pold; ++p)
        if (hslot[(size_t)p] >= 0) perm.push_back((int32_t)p);      // (Ur does not tell: the node in quarantine carries NaN there)
    if ((int64_t)perm.size() != n) { set_error("njp_rebuild_epoch: live positions do not match the active size"); return DPR_ERR_STATE; }
    sort_by_row_sum(perm, hU);
    lap("sort");
    if (q.graph) { (void)hipGraphExecDestroy(q.graph); q.graph = nullptr; }
    const NjPruned old = q;              // the old epoch's pointers (read by the permute / init kernels below)
    const int e = old.epoch_index + 1;
    q.epoch_index = e;
    if (int rc = njp_alloc_epoch(q, n, b.N, (e & 1) ? b.D : q.arena_D, q.arena_slab[e & 1], s, old.range_known ? old.t2_hdr : nullptr)) return rc;
    lap("njp_alloc_epoch");
    DPR_HIP(hipMemcpyAsync(q.perm, perm.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
    njp_launch_permute(old.D, old.ld, q.D, q.ld, q.perm, n, s);
    hipLaunchKernelGGL(njp_init_vectors_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       Ucur, q.perm, (const int32_t*)old.slot_of_pos, n, n, q.U + (st.it & 1) * q.vstride, q.Ur, q.KA, q.KB,
                       q.slot_of_pos, q.pos_of_slot);
    DPR_HIP(hipGetLastError());
    // iteration state: nothing in quarantine (every row sum is in memory), empty lists
    st.pnew[0] = -1; st.pnew[1] = -1;
    for (auto& c : st.cnt_list) c = 0ull;
    DPR_HIP(hipMemcpyAsync(b.st, &st, sizeof(NjState), hipMemcpyHostToDevice, s));
    lap("enqueue");
    DPR_HIP(hipStreamSynchronize(s));    // `st`, `perm` are host objects
    lap("device work");
    *rebuilt = true;
    return DPR_OK;
}

// ---- adaptive plan: hand-over between the pruned path (position space) and nj.hip's streaming loop (slot space) ----
__global__ __launch_bounds__(kThreads) void njp_gather_u_kernel(const double* __restrict__ Ucur, const int32_t* __restrict__ pos_of_slot,
                                                                int64_t n, double* __restrict__ U)
{
    const int64_t s = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (s < n) U[s] = Ucur[pos_of_slot[s]];
}

// position space -> dense slot space (b.D, b.U, b.Ur, b.KA).  Called between iterations with the node in quarantine
// materialised (njp_finish_kernel) and the stream idle.
static int njp_to_slots(NjBuffers& b, hipStream_t s)
{
    NjPruned& q = b.pr;
    NjState st;
    DPR_HIP(hipMemcpy(&st, b.st, sizeof(NjState), hipMemcpyDeviceToHost));
    const int64_t n = st.n;
    if (st.status != 0 || n < 3) return DPR_OK;
    if (q.graph) { (void)hipGraphExecDestroy(q.graph); q.graph = nullptr; }
    // the slot-space matrix goes into the buffer the epoch does not live in; the streaming kernels read b.D
    if (q.D == b.D) std::swap(b.D, q.arena_D);
    njp_launch_permute((const double*)q.D, q.ld, b.D, b.ld, (const int32_t*)q.pos_of_slot, n, s);
    hipLaunchKernelGGL(njp_gather_u_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       njp_current_u(q, st.it), (const int32_t*)q.pos_of_slot, n, b.U);
    DPR_HIP(hipGetLastError());
    if (int rc = nj_prepare(b, s)) return rc;        // Ur, KA for the active size in the state
    q.slots_mode = true;
    ++q.stream_epochs;
    return DPR_OK;
}

// dense slot space -> a fresh pruned epoch over the n active slots (the counterpart of njp_build for a run in progress)
static int njp_from_slots(NjBuffers& b, hipStream_t s)
{
    NjPruned& q = b.pr;
    NjState st;
    DPR_HIP(hipStreamSynchronize(s));
    DPR_HIP(hipMemcpy(&st, b.st, sizeof(NjState), hipMemcpyDeviceToHost));
    const int64_t n = st.n;
    if (st.status != 0 || n < 3) return DPR_OK;
    std::vector<double> hU((size_t)n);
    DPR_HIP(hipMemcpy(hU.data(), b.U, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    std::vector<int32_t> perm((size_t)n);
    std::iota(perm.begin(), perm.end(), 0);
    sort_by_row_sum(perm, hU);
    q.epoch_index = 0;                       // even: the epoch lives in arena_D (b.D holds the slot-space matrix it is built from)
    if (int rc = njp_alloc_epoch(q, n, b.N, q.arena_D, q.arena_slab[0], s)) return rc;
    DPR_HIP(hipMemcpyAsync(q.perm, perm.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
    njp_launch_permute((const double*)b.D, b.ld, q.D, q.ld, (const int32_t*)q.perm, n, s);
    q.range_known = false;
    if (njp_use_post2(q)) {        // (the streaming iterations created values nobody tracked: reduce the matrix again)
        hipLaunchKernelGGL(njp_range_kernel, dim3(2048), dim3(kThreads), 0, s, (const double*)b.D, b.ld, n, (unsigned long long*)q.t2_hdr);
        q.range_known = true;
    }
    hipLaunchKernelGGL(njp_init_vectors_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       (const double*)b.U, (const int32_t*)q.perm, (const int32_t*)nullptr, n, n, q.U + (st.it & 1) * q.vstride, q.Ur, q.KA, q.KB,
                       q.slot_of_pos, q.pos_of_slot);
    DPR_HIP(hipGetLastError());
    st.pnew[0] = -1; st.pnew[1] = -1;
    for (auto& c : st.cnt_list) c = 0ull;
    DPR_HIP(hipMemcpyAsync(b.st, &st, sizeof(NjState), hipMemcpyHostToDevice, s));
    DPR_HIP(hipStreamSynchronize(s));        // `st`, `perm` are host objects
    q.slots_mode = false;
    return DPR_OK;
}

// `todo` iterations of nj.hip's streaming loop on the slot-space matrix (two launches per iteration, a full Q scan each)
static int njp_run_slots(NjBuffers& b, int64_t it0, int64_t todo, hipStream_t s)
{
    const int64_t n0 = b.N - it0;
    for (int64_t k = 0; k < todo; ++k) {
        if (int rc = nj_launch_scan(b, false, n0 - k, it0 + k, s)) return rc;
        if (int rc = nj_launch_post(b, n0 - k, it0 + k, s)) return rc;
    }
    b.pr.stream_iterations += todo;
    return nj_launch_finish(b, n0 - todo, it0 + todo, s);
}

// epoch state only: the next njp_build finds the arena in place
void njp_reset(NjPruned& q)
{
    if (q.graph) { (void)hipGraphExecDestroy(q.graph); q.graph = nullptr; }
    NjPruned fresh;
    fresh.arena_D = q.arena_D; fresh.arena_slab[0] = q.arena_slab[0]; fresh.arena_slab[1] = q.arena_slab[1];
    fresh.arena_slab_bytes = q.arena_slab_bytes; fresh.arena_N = q.arena_N; fresh.arena_ranks = q.arena_ranks;
    fresh.adaptive = q.adaptive;      // (a setting of the context)
    fresh.dbg = q.dbg;      // (debug buffer: kept for njp_phase_stamps, a few hundred KB, DPR_NJ_PHASES runs only)
    q = fresh;
}

void njp_free(NjPruned& q)
{
    njp_reset(q);
    if (q.dbg) { if (g_njp_dbg_last == q.dbg) g_njp_dbg_last = nullptr; (void)hipFree(q.dbg); q.dbg = nullptr; }
    void* ptrs[] = { q.arena_D, q.arena_slab[0], q.arena_slab[1] };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    q = NjPruned();
}

// unit-sharded mode: unit-scan blocks per rank and unit records in total
static int njp_grid_rank(const NjPruned& q) { return q.sh_world > 1 ? (q.scan_grid / q.sh_world > 0 ? q.scan_grid / q.sh_world : 1) : q.scan_grid; }
static int njp_grid_total(const NjPruned& q) { return q.sh_world > 1 ? njp_grid_rank(q) * q.sh_world : q.scan_grid; }

// kernel arguments of rank v (its own list and counters; v is ignored outside the unit-sharded mode)
static NjpArgs njp_args(NjBuffers& b, int v)
{
    NjPruned& q = b.pr;
    const bool sh = q.sh_world > 1;
    const int slot = sh && q.sh_virtual ? v : 0;             // local storage index of this rank
    NjpArgs a;
    a.D = q.D; a.ld = q.ld; a.st = b.st;
    a.U = q.U; a.R = q.R; a.vstride = q.vstride;
    a.Ur = q.Ur; a.KA = q.KA; a.KB = q.KB; a.slot_of_pos = q.slot_of_pos; a.pos_of_slot = q.pos_of_slot;
    a.xpart = b.xpart; a.partials = b.partials; a.umin = (unsigned long long*)q.umin;
    a.P = q.P;
    a.blk_cb = q.blk_cb; a.blk_g0 = q.blk_g0;
    a.tg = njp_tg(q.P); a.ns = njp_ns(q.P); a.nupd = 0;
    a.ntest = sh ? (q.nprep - v + q.sh_world - 1) / q.sh_world : q.nprep;     // test blocks v, v + world, ... are this rank's
    a.list = q.list + (int64_t)slot * q.list_stride;
    a.cnt = sh ? q.cnt_all + 4 * slot : b.st->cnt_list;
    a.ugrid = njp_grid_rank(q);
    a.urecs = njp_grid_total(q);
    a.nrb = (int)((q.P + kTileCols - 1) / kTileCols);
    a.rec_off = sh ? v * a.ugrid : 0;
    a.all_defined = sh ? 1 : 0;
    a.sh_rank = sh ? v : 0; a.sh_world = sh ? q.sh_world : 1;
    a.cnt_all = sh ? q.cnt_all : nullptr;
    a.cnt_ranks = sh ? (q.sh_virtual ? q.sh_world : 1) : 0;
    a.do_update = 1; a.do_tests = 1; a.do_rows = 1;
    a.log_x = b.log_x; a.log_y = b.log_y; a.log_bx = b.log_bx; a.log_by = b.log_by;
    a.dbg = q.dbg; a.dbg_it = q.dbg_it;
    a.t2_hdr = q.t2_hdr; a.t2_rmax = q.t2_rmax; a.t2_cmax = q.t2_cmax; a.t2_colmin = q.t2_colmin; a.t2_rowmin = q.t2_rowmin; a.t2_cmin = q.t2_cmin;
    return a;
}

static int njp_launch_scan(NjBuffers& b, hipStream_t s, int v, bool rows)
{
    NjpArgs a = njp_args(b, v);
    a.do_rows = rows ? 1 : 0;
    hipLaunchKernelGGL((njp_scan_kernel<false>), dim3((unsigned)(a.ugrid + (rows ? a.nrb : 0))), dim3(kThreads), 0, s, a.st, (const int32_t*)a.list,
                       (const unsigned long long*)a.cnt, (const double*)a.xpart, rows ? a.nrb : 0, a);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

static bool njp_use_post2(const NjPruned& q)
{
    const int ns = njp_ns(q.P);
    return njp_post2_on() && ns == kBigNS && (q.dbg == nullptr || q.dbg_it >= 0);
}

static int njp_launch_post(NjBuffers& b, hipStream_t s, int v, bool update)
{
    NjpArgs a = njp_args(b, v);
    a.do_update = update ? 1 : 0;
    const unsigned ublocks = update ? (unsigned)((b.N + kThreads - 1) / kThreads) : 0u;
    a.nupd = (int)ublocks;
    if ((unsigned)a.ntest + ublocks == 0u) return DPR_OK;      // (a rank without test blocks in a tests-only launch)
    // large shape on a single rank: light blocks, maxima of the previous launch (njp_post2_kernel; DPR_NJP_POST2=0: the fused kernel)
    if (njp_use_post2(b.pr)) {
        const unsigned u2 = update ? (unsigned)((b.N + 2 * kThreads - 1) / (2 * kThreads)) : 0u;      // UM blocks: 512 reference slots and 512 positions each
        const unsigned um = !update ? 0u : (u2 > (unsigned)a.nrb ? u2 : (unsigned)a.nrb);
        a.nupd = (int)um;
        if ((unsigned)a.ntest + um == 0u) return DPR_OK;
        hipLaunchKernelGGL((njp_post2_kernel<kBigNS>), dim3((unsigned)a.ntest + um), dim3(kThreads), 0, s, a.st, (const NjRecord*)a.partials, (const unsigned long long*)a.cnt, a.blk_cb, a.blk_g0, (const int32_t*)a.pos_of_slot, a.ntest, a.nupd, a);
        DPR_HIP(hipGetLastError());
        return DPR_OK;
    }
    if (a.tg == 64) hipLaunchKernelGGL((njp_post_kernel<64, 1, false>), dim3((unsigned)a.ntest + ublocks), dim3(kThreads), 0, s, a.st, (const NjRecord*)a.partials, (const unsigned long long*)a.cnt, a.blk_cb, a.blk_g0, (const int32_t*)a.pos_of_slot, a.ntest, a.nupd, a);
    else hipLaunchKernelGGL((njp_post_kernel<256, kBigNS, false>), dim3((unsigned)a.ntest + ublocks), dim3(kThreads), 0, s, a.st, (const NjRecord*)a.partials, (const unsigned long long*)a.cnt, a.blk_cb, a.blk_g0, (const int32_t*)a.pos_of_slot, a.ntest, a.nupd, a);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// ---- the row-sharded instantiations (njr.hip builds the arguments and owns the loop) ----------------------------------
int njp_rs_launch_list_all(const NjpArgs& a, hipStream_t s)
{
    if (a.ntest > 0) hipLaunchKernelGGL(njp_list_all_kernel, dim3((unsigned)a.ntest), dim3(kThreads), 0, s, a);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}
int njp_rs_launch_scan(const NjpArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL((njp_scan_kernel<true>), dim3((unsigned)(a.ugrid + a.nrb)), dim3(kThreads), 0, s, a.st, (const int32_t*)a.list,
                       (const unsigned long long*)a.cnt, (const double*)a.xpart, a.nrb, a);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}
int njp_rs_launch_post(const NjpArgs& a0, int64_t N, hipStream_t s)
{
    NjpArgs a = a0;
    const unsigned ublocks = (unsigned)((N + kThreads - 1) / kThreads);
    a.nupd = (int)ublocks;
    hipLaunchKernelGGL((njp_post_kernel<64, 1, true>), dim3((unsigned)a.ntest + ublocks), dim3(kThreads), 0, s, a.st, (const NjRecord*)a.partials,
                       (const unsigned long long*)a.cnt, a.blk_cb, a.blk_g0, (const int32_t*)a.pos_of_slot, a.ntest, a.nupd, a);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}
int njp_rs_launch_finish(const NjpArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL((njp_finish_kernel<true>), dim3((unsigned)a.nrb), dim3(kThreads), 0, s, a);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}
// epoch e of a row-sharded run on this rank: slabs allocated on first use (the arena of the single-GPU path without its matrix
// buffer), the epoch's structures inside slab e & 1, matrix rows in Dbuf (one half of NjBuffers::D)
int njp_rs_epoch(NjPruned& q, int64_t P, int64_t N, double* Dbuf, int epoch_index, int rs_rank, int rs_world, const void* hdr_from, hipStream_t s)
{
    const SlabPlan plan = slab_plan(N, N, 1);
    if (!q.arena_slab[0] || q.arena_N != N || q.arena_slab_bytes < plan.total) {
        for (int k = 0; k < 2; ++k) { if (q.arena_slab[k]) (void)hipFree(q.arena_slab[k]); q.arena_slab[k] = nullptr; }
        for (int k = 0; k < 2; ++k) DPR_HIP(hipMalloc(&q.arena_slab[k], plan.total));
        q.arena_slab_bytes = plan.total; q.arena_N = N; q.arena_ranks = 1;
    }
    if (q.graph) { (void)hipGraphExecDestroy(q.graph); q.graph = nullptr; }
    q.sh_world = 1; q.sh_rank = 0; q.sh_virtual = false;
    q.epoch_index = epoch_index;
    if (int rc = njp_alloc_epoch(q, P, N, Dbuf, q.arena_slab[epoch_index & 1], s, hdr_from, rs_world, rs_rank)) return rc;
    q.active = true;
    return DPR_OK;
}
// vectors of a new epoch from the previous one's (or from the tip-order row sums: slot_src = nullptr), as njp_build / njp_rebuild_epoch do
int njp_rs_init_vectors(NjPruned& q, const double* U_src, const int32_t* slot_src, int64_t P, int64_t n, int64_t it, hipStream_t s)
{
    hipLaunchKernelGGL(njp_init_vectors_kernel, dim3((unsigned)((P + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       U_src, (const int32_t*)q.perm, slot_src, P, n, q.U + (it & 1) * q.vstride, q.Ur, q.KA, q.KB, q.slot_of_pos, q.pos_of_slot);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}
void njp_rs_sort_by_row_sum(std::vector<int32_t>& perm, const std::vector<double>& hU) { sort_by_row_sum(perm, hU); }
int64_t njp_vec_len(int64_t N) { return vec_len(N); }
int njp_scan_grid_default()
{
    const char* e = std::getenv("DPR_NJP_GRID");
    const int g = e ? std::atoi(e) : 512;
    return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

// one iteration: SCAN -> POST; every kernel reads its iteration index from the device state.
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
        if (sample) { b.kt->nk = 3; njp_set_kernel_names(nullptr); }      // the third interval holds nothing: what an event pair itself costs
        if (int rc = mark()) return rc;
        if (int rc = njp_launch_scan(b, s, 0, true)) return rc;
        if (int rc = mark()) return rc;
        if (int rc = njp_launch_post(b, s, 0, true)) return rc;
        if (int rc = mark()) return rc;
        return mark();
    }
    // Unit-sharded mode (every rank holds the whole position-space matrix): a unit belongs to rank
    // (strip * G16 + group) mod world for good.  Each rank tests and scans only its own units -- a unit that
    // holds the winner always survives its owner's test, whatever the other ranks' bounds are -- so the unit
    // bounds stay private to their owner and ONE small all-gather per iteration (the unit records) is the
    // only exchange; the new-row blocks, select + merge + update run replicated.
    const int gr = njp_grid_rank(q);
    const int v0 = q.sh_virtual ? 0 : q.sh_rank, v1 = q.sh_virtual ? q.sh_world : q.sh_rank + 1;
    for (int v = v0; v < v1; ++v)
        if (int rc = njp_launch_scan(b, s, v, v == v0)) return rc;
    if (!q.sh_virtual) {
        if (!q.gather) { set_error("njp: unit-sharded mode without a gather callback"); return DPR_ERR_STATE; }
        if (int rc = q.gather(q.gather_ctx, b.partials, sizeof(NjRecord) * (size_t)gr, s)) return rc;
    }
    // (virtual ranks: the update runs once, with the first rank's tests; the other ranks' tests follow in launches of
    // their own -- they read only what the update leaves alone: the current U buffer, rows x and y, the scan records)
    for (int v = v0; v < v1; ++v)
        if (int rc = njp_launch_post(b, s, v, v == v0)) return rc;
    return DPR_OK;
}

// enqueue `todo` iterations starting at iteration it0.  The kernels of an iteration take no per-iteration
// arguments, so kGraphIters iterations are captured once into a hipGraph and replayed; iterations beyond
// it_limit are no-ops.  Afterwards the node in quarantine is materialised (row sum, matrix row).
// (NjPruned::graph_iters, default 32; DPR_NJ_GRAPH_ITERS at njp_build)

static int njp_run_segment(NjBuffers& b, int64_t it0, int64_t todo, hipStream_t s)
{
    NjPruned& q = b.pr;
    const int64_t limit = it0 + todo;
    DPR_HIP(hipMemcpyAsync(&b.st->it_limit, &limit, sizeof(int64_t), hipMemcpyHostToDevice, s));
    DPR_HIP(hipStreamSynchronize(s));   // `limit` is a stack variable
    if (todo <= 0) return DPR_OK;
    const int v0 = q.sh_world > 1 && !q.sh_virtual ? q.sh_rank : 0;
    const int v1 = q.sh_world > 1 ? (q.sh_virtual ? q.sh_world : q.sh_rank + 1) : 1;
    if (q.fresh) {      // first scan of an epoch: every unit
        for (int v = v0; v < v1; ++v) {
            NjpArgs a = njp_args(b, v);
            if (a.ntest > 0) hipLaunchKernelGGL(njp_list_all_kernel, dim3((unsigned)a.ntest), dim3(kThreads), 0, s, a);
        }
        DPR_HIP(hipGetLastError());
        if (njp_use_post2(q)) {
            // the maxima the first post launch of the epoch reads: the row sums as they stand (buffer of the current iteration)
            const int64_t S2 = (q.P + kTileCols - 1) / kTileCols + 2;
            const int par = (int)(it0 & 1);
            hipLaunchKernelGGL(njp_t2_init_kernel, dim3((unsigned)((q.P + kTileCols - 1) / kTileCols)), dim3(kThreads), 0, s,
                               (const double*)(q.U + (it0 & 1) * q.vstride), q.P, q.t2_rmax + par * 32 * S2, q.t2_cmax + par * 4 * S2);
            DPR_HIP(hipGetLastError());
        }
        q.fresh = false;
    }
    const bool timing = b.kt && b.kt->stride > 0 && q.sh_world <= 1;
    const int kGraphIters = q.graph_iters;
    const bool use_graph = q.sh_world <= 1 && todo >= kGraphIters && !timing;      // (DPR_NJ_GRAPH_ITERS above `todo`: eager launches)
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
        if (log_level("epoch") > 0)
            std::fprintf(stderr, "[njp] graph capture + instantiate (P=%lld): %.2f ms\n", (long long)q.P,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg0).count());
    }
    int64_t done = 0;
    if (use_graph)
        for (; done + kGraphIters <= todo; done += kGraphIters) DPR_HIP(hipGraphLaunch(q.graph, s));
    for (; done < todo; ++done)
        if (int rc = njp_enqueue_iteration(b, s, timing && (it0 + done) % b.kt->stride == 0)) return rc;
    {
        NjpArgs a = njp_args(b, v0);
        hipLaunchKernelGGL((njp_finish_kernel<false>), dim3((unsigned)a.nrb), dim3(kThreads), 0, s, a);
        DPR_HIP(hipGetLastError());
    }
    return DPR_OK;
}

static const char* const kNjpKernelNames[] = { "njp_scan_kernel", "njp_post_kernel", "(empty event pair)", "" };
static const char* const* g_kernel_names = kNjpKernelNames;
void njp_set_kernel_names(const char* const* names) { g_kernel_names = names ? names : kNjpKernelNames; }
const char* njp_kernel_name(int idx)
{
    if (idx < 0 || idx >= kNjKernelsMax) return "";
    for (int k = 0; k <= idx; ++k)
        if (g_kernel_names[k][0] == 0) return "";
    return g_kernel_names[idx];
}

// debug: the unit list the next scan would walk (codes: sub-unit mask << 28 | strip << 18 | row group) and the row sums by position
// (Ur: U / (n - 2); NaN = dead position or the node in quarantine) of the current epoch, after a dpr_nj_run that stopped early
int njp_debug_list(NjBuffers& b, int32_t* out, int64_t cap, int64_t* count, int64_t* P, double* ur, int64_t urcap)
{
    NjPruned& q = b.pr;
    if (!q.active || q.slots_mode || q.sh_world > 1) { set_error("njp_debug_list: no single-rank pruned epoch"); return DPR_ERR_STATE; }
    DPR_HIP(hipDeviceSynchronize());
    NjState st;
    DPR_HIP(hipMemcpy(&st, b.st, sizeof(NjState), hipMemcpyDeviceToHost));
    const int64_t cnt = (int64_t)st.cnt_list[st.it % 3];
    *count = cnt; *P = q.P;
    const int64_t take = cnt < cap ? cnt : cap;
    if (take > 0) DPR_HIP(hipMemcpy(out, q.list, sizeof(int32_t) * (size_t)take, hipMemcpyDeviceToHost));
    const int64_t tu = q.P < urcap ? q.P : urcap;
    if (ur && tu > 0) DPR_HIP(hipMemcpy(ur, q.Ur, sizeof(double) * (size_t)tu, hipMemcpyDeviceToHost));
    return DPR_OK;
}

// debug: the phase stamps of iteration DPR_NJ_PHASES (2 x 2048 x 8 words), 0 = not stamped
int njp_phase_stamps(unsigned long long* out)
{
    if (!g_njp_dbg_last) { set_error("DPR_NJ_PHASES not set"); return DPR_ERR_STATE; }
    DPR_HIP(hipDeviceSynchronize());
    DPR_HIP(hipMemcpy(out, g_njp_dbg_last, sizeof(unsigned long long) * 4 * 2048 * 8, hipMemcpyDeviceToHost));
    return DPR_OK;
}

// current U buffer of the pruned path (row sums by position) after `it` iterations
const double* njp_current_u(const NjPruned& q, int64_t it) { return q.U + (it & 1) * q.vstride; }

int njp_shape(const NjPruned& q, int64_t* positions, int* row_groups, int* strips, int* post2, int* scan_grid)
{
    if (positions) *positions = q.P;
    if (row_groups) *row_groups = njp_tg(q.P);
    if (strips) *strips = njp_ns(q.P);
    if (post2) *post2 = njp_use_post2(q) ? 1 : 0;
    if (scan_grid) *scan_grid = q.scan_grid;
    return DPR_OK;
}

// enqueue `todo` iterations starting at iteration it0, in epochs: whenever the active size has dropped to
// pct % of the epoch's positions (and the epoch is large enough to matter) the position space is rebuilt; with the
// adaptive plan on, the listing rate is watched and the run handed over to the streaming loop (and back) as described in
// dpr_internal.hpp
int njp_run(NjBuffers& b, int64_t it0, int64_t todo, hipStream_t s)
{
    const char* e_min = std::getenv("DPR_NJ_EPOCH_MIN");
    const int64_t epoch_min = e_min ? std::atoll(e_min) : 2048;   // epochs smaller than this are not rebuilt
    const int64_t pct = 80;           // rebuild once n <= pct% of the epoch's positions (85 / 90 measured: DESIGN.md section 4.4)
    NjPruned& q = b.pr;
    int64_t it = it0, left = todo;
    if (left <= 0) return q.slots_mode ? DPR_OK : njp_run_segment(b, it0, 0, s);
    while (left > 0) {
        const int64_t n = b.N - it;
        if (q.slots_mode) {
            // streaming loop until the next probe point (or the end); probes only while an epoch is worth building
            if (n <= q.slots_probe_n && n >= epoch_min && n >= 3) {
                if (int rc = njp_from_slots(b, s)) return rc;
                if (!q.slots_mode) continue;        // a fresh pruned epoch: probed below
            }
            int64_t seg = left;
            if (n > q.slots_probe_n && q.slots_probe_n >= epoch_min && n - q.slots_probe_n < seg) seg = n - q.slots_probe_n;
            if (int rc = njp_run_slots(b, it, seg, s)) return rc;
            it += seg; left -= seg;
            continue;
        }
        const int64_t P = q.P;
        int64_t seg = left;
        if (epoch_min > 0 && P >= epoch_min) {
            const int64_t target = P * pct / 100;
            if (n <= target && n >= 3) {
                DPR_HIP(hipStreamSynchronize(s));
                const auto t0 = std::chrono::steady_clock::now();
                bool rebuilt = false;
                if (int rc = njp_rebuild_epoch(b, s, &rebuilt)) return rc;
                // not rebuilt: the run has no candidate left (status != 0: e.g. only rows with NaN row sums are still active -- a NaN
                // distance makes the row sums of ITS two rows NaN, those rows never win and stay to the end) and every queued
                // kernel is a no-op -- stop here, dpr_nj_run reports DPR_ERR_NOCAND
                if (!rebuilt) return DPR_OK;
                if (log_level("epoch") > 0)
                    std::fprintf(stderr, "[njp] epoch rebuild at n=%lld: %.2f ms\n", (long long)n,
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                continue;
            }
            if (n - target < seg) seg = n - target;
        }
        // adaptive plan: look at the listing rate after the first graph of an epoch and then every 2 048 iterations
        const bool watch = q.adaptive && q.sh_world <= 1 && q.utot > 0;
        unsigned long long units0 = 0;
        bool first_of_epoch = false;
        if (watch) {
            first_of_epoch = q.fresh;
            const int64_t chunk = first_of_epoch ? (int64_t)q.graph_iters : 2048;
            if (seg > chunk) seg = chunk;
            NjState st0;
            DPR_HIP(hipStreamSynchronize(s));
            DPR_HIP(hipMemcpy(&st0, b.st, sizeof(NjState), hipMemcpyDeviceToHost));
            units0 = st0.units_scanned;
        }
        if (int rc = njp_run_segment(b, it, seg, s)) return rc;
        it += seg; left -= seg;
        if (watch && left > 0) {
            NjState st1;
            DPR_HIP(hipStreamSynchronize(s));
            DPR_HIP(hipMemcpy(&st1, b.st, sizeof(NjState), hipMemcpyDeviceToHost));
            if (st1.status != 0) continue;
            // (the first scan of an epoch lists every unit by construction: left out of the rate)
            const double listed = (double)(st1.units_scanned - units0) - (first_of_epoch ? (double)q.utot : 0.0);
            const double iters = (double)seg - (first_of_epoch ? 1.0 : 0.0);
            const double rate = iters >= 1.0 ? listed / (iters * (double)q.utot) : 0.0;
            const double per_it = iters >= 1.0 ? listed / iters : 0.0;       // units per iteration
            if (log_level("epoch") >= 2)
                std::fprintf(stderr, "[njp] watch at n=%lld: %.0f units per iteration over %.0f iterations (first of epoch %d, P=%lld, utot=%lld, scan grid %d)\n",
                             (long long)(b.N - it), per_it, iters, (int)first_of_epoch, (long long)q.P, (long long)q.utot, q.scan_grid);
            if (iters >= 1.0 && rate > q.stream_frac) {
                // hand over; probe a pruned epoch again after the active size has shrunk by the epoch factor 1, 2, 4, 8 ... times
                const int64_t na = b.N - it;
                int64_t pn = na;
                const int hops = 1 << (q.probe_fail_streak < 4 ? q.probe_fail_streak : 4);
                for (int h = 0; h < hops; ++h) pn = pn * pct / 100;
                ++q.probe_fail_streak;
                q.slots_probe_n = pn;
                if (int rc = njp_to_slots(b, s)) return rc;
                if (log_level("epoch") > 0)
                    std::fprintf(stderr, "[njp] %.0f %% of the units listed per iteration at n=%lld: handed over to the streaming loop; next pruned probe at n <= %lld\n",
                                 100.0 * rate, (long long)na, (long long)pn);
            } else if (!first_of_epoch) {
                q.probe_fail_streak = 0;       // the epoch keeps pruning
            }
        }
    }
    return DPR_OK;
}

}  // namespace dpr
