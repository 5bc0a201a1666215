// ROW-SHARDED exact pruned neighbor joining (several GPUs): north_star's layout -- the N x N matrix as row blocks over the
// ranks -- under the pruned algorithm of njp.hip instead of one full Q scan per iteration (njs.hip).  Same arithmetic, keys and
// merge log as the single-GPU paths, bit for bit; replaces the loop src/neighborJoining.cu:197-249 of the reference.
//
// Why: the pruned loop on ONE GPU needs the whole matrix twice (two epoch buffers: 2 x 80 GB at 100 000 tips, nothing above
// ~130 000), and the row-sharded plan that existed streams 4 n^2 bytes per iteration.  Here a rank holds 1 / G of the rows of
// the position-space matrix, owns the 16 x 512 units of those rows (bounds, tests, lists, scans: all private to the owner) and
// an iteration moves 2 x 8 n bytes plus ~16 KB of block records between the ranks.
//
// Layout.  Positions are dealt in chunks of kNjrChunk = 1024 (64 row groups = the rows of one test block of
// njp_post_kernel<64, 1>): chunk k belongs to rank k % G and is that rank's (k / G)-th chunk; rows at full width.  The
// per-position vectors (row sums, keys, slot maps, the new node's row buffer) and the merge log are REPLICATED: every rank
// applies the whole update, so they are bit-identical everywhere by construction (and checked: below).
//
// One iteration per rank, three launches:
//   SCAN(it)     njp_scan_kernel<kRS>: the listed units of the own rows + the new-row blocks (replicated work: they read only
//                replicated vectors; the owner of the new node's position also moves its buffered row into the matrix).
//                Per rank ugrid = grid / G unit records + one header record (the bits of the row sum U[x] this rank derived).
//   [records]    all ranks' records to every rank: one all-gather, in place (plan COLLECTIVE), or -- plan MAILBOX -- stored
//                straight into every rank's window by the last block of SCAN.
//   EXTRACT(it)  njr_extract_kernel: every rank reduces the G x (ugrid + 1) + new-row records to the winner (px, py), compares
//                the header words (a differing word = some rank's replicated state went wrong: DPR_ERR_COMM on every rank
//                instead of a silently different tree), and extracts COLUMNS px and py from its own rows: the matrix is
//                symmetric bit for bit (column and row of a new node are stored with the same values), so the columns over all
//                ranks' rows ARE rows px and py.  No rank ever reads another rank's matrix during the loop.
//   [slices]     the column slices to every rank: second all-gather (COLLECTIVE), or stored into every rank's window (MAILBOX).
//   POST(it)     njp_post_kernel<64, 1, kRS>: select + merge + update with rows x / y taken from the exchanged slices, the new
//                node's column stored into the own rows, the unit tests of iteration it + 1 for the own test blocks.
// Epoch rebuilds (positions compacted and re-sorted by row sum, njp.hip) are a sharded permute: a rank pulls the source rows
// its new rows need from their owners' buffers (coalesced, into a staging buffer) and gathers the columns locally; the two
// epoch buffers of a rank are halves of one allocation, so one hipIpc mapping per peer serves all epochs.
#include "njp_args.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>

namespace dpr {

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

NjrLayout njr_layout(int64_t N, int world)
{
    NjrLayout l;
    l.rec_stride = 1024 / (world > 0 ? world : 1) + 1;                 // unit-scan blocks per rank (grid <= 1024) + the header record
    l.slice = njr_slice_len(N, world);
    l.off_recflag = 0;                                                  // [2][kNjsMaxWorld] 64-byte lines
    l.off_recs = (int64_t)(2 * 64 * kNjsMaxWorld);                      // [2][kNjsMaxWorld][rec_stride] records
    l.off_rowflag = (int64_t)align256((size_t)l.off_recs + sizeof(NjRecord) * (size_t)(2 * kNjsMaxWorld) * (size_t)l.rec_stride);
    l.off_rows = l.off_rowflag + 64 * kNjsMaxWorld;                     // [world][2][slice] doubles
    l.bytes = l.off_rows + (int64_t)sizeof(double) * 2 * world * l.slice;
    return l;
}

// local rows of rank `rank` whose position is < P
static int64_t njr_local_count(int64_t P, int rank, int world)
{
    const int64_t full = P / kNjrChunk, rem = P % kNjrChunk;
    int64_t cnt = (full / world + ((full % world) > rank ? 1 : 0)) * kNjrChunk;
    if (rem && (full % world) == rank) cnt += rem;
    return cnt;
}

// ------------------------------------------------------------------------------------------------
// EXTRACT(it): see the header.  Grid: a few blocks (each strides over the own rows); the records are reduced by every block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void njr_extract_kernel(NjpArgs a, int64_t rows_local)
{
    __shared__ double sq[kThreads / 64], sdd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64], spp[kThreads / 64];
    __shared__ unsigned long long s_hdr[kNjsMaxWorld], s_seq[kNjsMaxWorld];
    __shared__ int s_fail;
    __shared__ unsigned int s_last;
    const int tid = threadIdx.x;
    const int64_t it = a.st->it, limit = a.st->it_limit, N = a.st->N;
    if (a.st->status != 0 || it >= limit) return;
    const int64_t n = N - it;
    if (n < 3) return;
    const int W = a.rs_world, rank = a.rs_rank, nrec = a.ugrid + 1, par = (int)(it & 1);
    const bool mailbox = a.rs_plan == kNjrMailbox;
    const unsigned long long want = a.rs_seq_base + (unsigned long long)(it + 1);
    if (tid == 0) s_fail = 0;
    __syncthreads();
    if (mailbox && tid < W) {
        const unsigned long long* f = njr_win_recflag(a.rs_win[rank], a.rs_lay, par, tid);
        const unsigned long long t0 = wall_clock64();
        while (njr_ld_sys(f) != want) {
            if (wall_clock64() - t0 > a.rs_poll_ticks) { s_fail = 3; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (s_fail) {
        if (blockIdx.x == 0 && tid == 0) a.st->status = s_fail;
        return;
    }
    // ---- the winner over all ranks' unit records and the (replicated) new-row records
    double bq = 10000.0, d = 0.0; uint64_t bk = ~0ull, bp = 0;
    // (a thread's records are all loaded before the first one is used: one round trip, not one per record)
    constexpr int kPre = 4;
    for (int idx0 = tid; idx0 < W * nrec; idx0 += kPre * kThreads) {
        unsigned long long w4[kPre][4];
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int idx = idx0 + u * kThreads;
            w4[u][0] = (unsigned long long)__double_as_longlong(10000.0); w4[u][1] = ~0ull; w4[u][2] = 0ull; w4[u][3] = 0ull;
            if (idx >= W * nrec) continue;
            const int r = idx / nrec, k = idx - r * nrec;
            if (mailbox) {
                const unsigned long long* w = reinterpret_cast<const unsigned long long*>(njr_win_recs(a.rs_win[rank], a.rs_lay, par, r)) + 4 * k;
#pragma unroll
                for (int c = 0; c < 4; ++c) w4[u][c] = njr_ld_sys(w + c);
            } else {
                const unsigned long long* w = reinterpret_cast<const unsigned long long*>(a.partials + idx);
#pragma unroll
                for (int c = 0; c < 4; ++c) w4[u][c] = w[c];
            }
        }
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int idx = idx0 + u * kThreads;
            if (idx >= W * nrec) continue;
            const int r = idx / nrec, k = idx - r * nrec;
            const double q = __longlong_as_double((long long)w4[u][0]), dd = __longlong_as_double((long long)w4[u][2]);
            if (mailbox && blockIdx.x == 0) { NjRecord rec; rec.q = q; rec.key = w4[u][1]; rec.d = dd; rec.pad = w4[u][3]; a.partials[idx] = rec; }      // POST reads them here
            if (k == 0) { s_hdr[r] = w4[u][2]; s_seq[r] = w4[u][3]; }      // header: row sum bits, sequence number
            best_update4(bq, bk, bp, d, q, w4[u][1], w4[u][3], dd);
        }
    }
    for (int idx = tid; idx < a.nrb; idx += kThreads) {
        const NjRecord rec = a.partials[a.urecs + idx];
        best_update4(bq, bk, bp, d, rec.q, rec.key, rec.pad, rec.d);
    }
    wave_best4(bq, bk, bp, d);
    if ((tid & 63) == 0) { sq[tid >> 6] = bq; sk[tid >> 6] = bk; spp[tid >> 6] = bp; sdd[tid >> 6] = d; }
    __syncthreads();
    bq = sq[0]; bk = sk[0]; bp = spp[0]; d = sdd[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) best_update4(bq, bk, bp, d, sq[w], sk[w], spp[w], sdd[w]);
    // ---- the replicated state must agree: every rank's header carries the bits of the row sum it derived for the node of
    // merge it - 1 and the sequence number of THIS iteration (a record of another iteration: the exchange did not happen)
    if (tid < W) {
        if (s_seq[tid] != want) s_fail = 3;
        else if (s_hdr[tid] != s_hdr[rank]) s_fail = 4;
    }
    __syncthreads();
    if (s_fail) {
        if (blockIdx.x == 0 && tid == 0) {
            a.st->status = s_fail;
            if (s_fail == 4) {
                int other = 0;
                for (int r = 0; r < W; ++r) if (s_hdr[r] != s_hdr[rank]) { other = r; break; }
                a.st->q = __longlong_as_double((long long)s_hdr[rank]); a.st->d = __longlong_as_double((long long)s_hdr[other]);
                a.st->x = rank; a.st->y = other;
            }
        }
        return;
    }
    if (bk == ~0ull || !(bq < 10000.0)) return;                 // no candidate (q == 10000.0 is none: the reference's strict `<`): POST reports it (status 1) on every rank
    const int64_t ki = (int64_t)(bk & 0xFFFFFFull), kj = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t pi = (int64_t)(bp & 0xffffffffull), pj = (int64_t)(bp >> 32);
    const int64_t px = ki < kj ? pi : pj, py = ki < kj ? pj : pi;
    // ---- columns px and py of the own rows
    double* own_x = const_cast<double*>(a.rs_rows) + (int64_t)(2 * rank) * a.rs_slice;
    double* own_y = own_x + a.rs_slice;
    for (int64_t l = (int64_t)blockIdx.x * kThreads + tid; l < rows_local; l += (int64_t)gridDim.x * kThreads) {
        const double vx = a.D[l * a.ld + px], vy = a.D[l * a.ld + py];
        if (mailbox) {
            const unsigned long long ux = (unsigned long long)__double_as_longlong(vx), uy = (unsigned long long)__double_as_longlong(vy);
            for (int r = 0; r < W; ++r) {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.rs_win[r] + a.rs_lay.off_rows) + (int64_t)(2 * rank) * a.rs_slice + l;
                njr_st_sys(dst, ux);
                njr_st_sys(dst + a.rs_slice, uy);
            }
        } else {
            own_x[l] = vx; own_y[l] = vy;
        }
    }
    if (!mailbox) return;
    // the launch's last block tells every rank that this rank's slices are complete
    unsigned int* tk = a.rs_ticket + kNjsTicketBytes / sizeof(unsigned int);      // (the second set: SCAN uses the first)
    if (!njr_last_block(tk, &s_last)) return;
    if (tid <= (int)kNjsTicketGroups) tk[32 * tid] = 0u;
    if (tid < W) njr_st_flag(njr_win_rowflag(a.rs_win[tid], a.rs_lay, rank), want);
}

// barrier over the ranks through the windows (mailbox plan, no RCCL): epoch numbers only grow
__global__ void njr_barrier_kernel(char* const* win, NjrLayout lay, int rank, int world, unsigned long long epoch, unsigned long long poll_ticks, NjState* st)
{
    const int tid = threadIdx.x;
    if (tid >= world) return;
    // (the row flags' lines hold 8 words each; word 1 of a line is the barrier word of that source rank)
    unsigned long long* theirs = njr_win_rowflag(win[tid], lay, rank) + 1;
    njr_st_flag(theirs, epoch);
    const unsigned long long* mine = njr_win_rowflag(win[rank], lay, tid) + 1;
    const unsigned long long t0 = wall_clock64();
    while (njr_ld_sys(mine) < epoch) {
        if (wall_clock64() - t0 > poll_ticks) { st->status = 3; break; }
        __builtin_amdgcn_s_sleep(4);
    }
}

// ------------------------------------------------------------------------------------------------
// epoch builds: B_local[l][b] = A[perm[a]][perm[b]] for the own positions a = global_pos(l)
// ------------------------------------------------------------------------------------------------
// source rows of a batch of own output rows, pulled from their owners (coalesced; system-scope loads: the owner wrote them in
// earlier kernels, behind a barrier over the ranks) into the local staging buffer.  Source row s lives at
// src[(s / blk) % world] + ((s / blk / world) * blk + s % blk) * ld_src  (blk = 64: the tip-order matrix; 1024: an epoch)
__global__ __launch_bounds__(kThreads) void njr_stage_kernel(double* const* __restrict__ src, int64_t ld_src, int blk, int world, int64_t ncols,
                                                             const int32_t* __restrict__ perm, int64_t P, int64_t l0, int count, int rank,
                                                             double* __restrict__ stage, int64_t stage_ld)
{
    for (int i = blockIdx.y; i < count; i += gridDim.y) {
        const int64_t a = njr_global_pos(l0 + i, rank, world);
        if (a >= P) continue;
        const int64_t sr = perm[a], sb = sr / blk;
        const double* row = src[sb % world] + ((sb / world) * blk + sr % blk) * ld_src;
        double* dst = stage + (int64_t)i * stage_ld;
        for (int64_t c = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 2; c < ncols; c += (int64_t)gridDim.x * kThreads * 2) {
            const unsigned long long v0 = njr_ld_sys(reinterpret_cast<const unsigned long long*>(row + c));
            const unsigned long long v1 = njr_ld_sys(reinterpret_cast<const unsigned long long*>(row + c + 1));      // (rows are padded to even lengths)
            dst[c] = __longlong_as_double((long long)v0);
            dst[c + 1] = __longlong_as_double((long long)v1);
        }
    }
}
// ... and the column gather out of the staged rows (the rows of a batch on one XCD each, as njp_permute_kernel does)
__global__ __launch_bounds__(kThreads) void njr_gather_kernel(const double* __restrict__ stage, int64_t stage_ld, const int32_t* __restrict__ perm, int64_t P,
                                                              double* __restrict__ B, int64_t ldb, int64_t l0, int count, int rank, int world)
{
    const int xcd = (int)(blockIdx.x & 7u), chunk = (int)(blockIdx.x >> 3), nchunk = (int)(gridDim.x >> 3);
    for (int i = (int)blockIdx.y * 8 + xcd; i < count; i += (int)gridDim.y * 8) {
        if (njr_global_pos(l0 + i, rank, world) >= P) continue;
        const double* row = stage + (int64_t)i * stage_ld;
        double* out = B + (l0 + i) * ldb;
        for (int64_t b = (int64_t)chunk * kThreads + threadIdx.x; b < P; b += (int64_t)nchunk * kThreads) out[b] = row[perm[b]];
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
void njr_free(NjBuffers& b)
{
    NjRowShard& r = b.rs;
    void* ptrs[] = { r.rows_plain, r.stage, r.d_region, r.d_src, r.ticket };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    NjRowShard fresh;
    fresh.gather = r.gather; fresh.barrier = r.barrier; fresh.cb_ctx = r.cb_ctx;
    r = fresh;
}

static NjpArgs njr_args(NjBuffers& b)
{
    NjPruned& q = b.pr;
    NjRowShard& r = b.rs;
    NjpArgs a{};
    a.D = q.D; a.ld = q.ld; a.st = b.st;
    a.U = q.U; a.R = q.R; a.vstride = q.vstride;
    a.Ur = q.Ur; a.KA = q.KA; a.KB = q.KB; a.slot_of_pos = q.slot_of_pos; a.pos_of_slot = q.pos_of_slot;
    a.xpart = b.xpart; a.partials = b.partials; a.umin = (unsigned long long*)q.umin;
    a.P = q.P;
    a.blk_cb = q.blk_cb; a.blk_g0 = q.blk_g0; a.ntest = q.nprep;
    a.tg = kNjrChunk / kUR; a.ns = 1; a.nupd = 0;
    a.list = q.list; a.cnt = b.st->cnt_list;
    a.ugrid = q.scan_grid / r.world > 0 ? q.scan_grid / r.world : 1;
    a.urecs = (a.ugrid + 1) * r.world;
    a.nrb = (int)((q.P + kTileCols - 1) / kTileCols);
    a.rec_off = r.rank * (a.ugrid + 1) + 1;
    a.all_defined = 1;
    a.sh_rank = 0; a.sh_world = 1;
    a.cnt_all = nullptr; a.cnt_ranks = 0;
    a.do_update = 1; a.do_tests = 1; a.do_rows = 1;
    a.log_x = b.log_x; a.log_y = b.log_y; a.log_bx = b.log_bx; a.log_by = b.log_by;
    a.dbg = nullptr; a.dbg_it = -1;
    a.t2_hdr = q.t2_hdr; a.t2_rmax = q.t2_rmax; a.t2_cmax = q.t2_cmax; a.t2_colmin = q.t2_colmin; a.t2_rowmin = q.t2_rowmin; a.t2_cmin = q.t2_cmin;
    a.rs_world = r.world; a.rs_rank = r.rank;
    a.rs_inv16 = (65536u + (unsigned int)r.world - 1u) / (unsigned int)r.world;
    a.rs_slice = r.lay.slice;
    a.rs_plan = r.plan;
    a.rs_lay = r.lay;
    a.rs_win = reinterpret_cast<char* const*>(r.d_region);
    a.rs_rows = r.plan == kNjrMailbox ? reinterpret_cast<const double*>(b.peer.win + r.win_off + r.lay.off_rows) : r.rows_plain;
    a.rs_ticket = r.ticket;
    a.rs_seq_base = b.peer.run_id << 32;
    a.rs_poll_ticks = b.peer.poll_ticks;
    a.rs_fault_it = b.peer.fault_rank == r.rank ? b.peer.fault_it : -1;
    return a;
}

static int njr_launch_extract(NjBuffers& b, hipStream_t s)
{
    const NjpArgs a = njr_args(b);
    const int64_t rows_local = njr_local_count(b.pr.P, b.rs.rank, b.rs.world);
    int64_t grid = (rows_local + kThreads - 1) / kThreads;
    grid = grid < 1 ? 1 : (grid > 32 ? 32 : grid);
    hipLaunchKernelGGL(njr_extract_kernel, dim3((unsigned)grid), dim3(kThreads), 0, s, a, rows_local);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// all ranks in step, streams idle (epoch builds; start and end of a run segment)
static int njr_barrier(std::vector<NjBuffers*>& ranks, hipStream_t s)
{
    NjBuffers& b0 = *ranks[0];
    if (ranks.size() > 1) return DPR_OK;      // virtual ranks: one stream orders everything
    if (b0.rs.barrier) return b0.rs.barrier(b0.rs.cb_ctx);
    // no callback: ranks joined without RCCL -- barrier through the windows
    hipLaunchKernelGGL(njr_barrier_kernel, dim3(1), dim3(kNjsMaxWorld), 0, s, reinterpret_cast<char* const*>(b0.rs.d_region), b0.rs.lay, b0.rs.rank, b0.rs.world,
                       ++b0.peer.bar_epoch, b0.peer.poll_ticks, b0.st);
    DPR_HIP(hipGetLastError());
    DPR_HIP(hipStreamSynchronize(s));
    return DPR_OK;
}

// matrix rows of a new epoch on one rank: for batches of own output rows, stage the source rows, gather the columns.
// src_half: index of the epoch buffer the source lives in (all ranks'), blk / ncols: its row-block size and width
static int njr_permute(NjBuffers& b, int src_half, int64_t ld_src, int blk, int64_t ncols, double* dst, int64_t ldb, int64_t P, hipStream_t s)
{
    NjRowShard& r = b.rs;
    const int64_t rows_local = njr_local_count(P, r.rank, r.world);
    double** d_src = r.d_src;      // device array [world]: the source half of every rank
    DPR_HIP(hipMemcpyAsync(d_src, r.peer_half[src_half].data(), sizeof(double*) * (size_t)r.world, hipMemcpyHostToDevice, s));
    int64_t chunks = (P + 1023) / 1024;
    chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
    for (int64_t l0 = 0; l0 < rows_local; l0 += r.stage_rows) {
        const int count = (int)(rows_local - l0 < r.stage_rows ? rows_local - l0 : r.stage_rows);
        int64_t gx = (ncols / 2 + kThreads - 1) / kThreads;
        gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
        hipLaunchKernelGGL(njr_stage_kernel, dim3((unsigned)gx, (unsigned)count), dim3(kThreads), 0, s, (double* const*)d_src, ld_src, blk, r.world, ncols,
                           (const int32_t*)b.pr.perm, P, l0, count, r.rank, r.stage, r.stage_ld);
        hipLaunchKernelGGL(njr_gather_kernel, dim3((unsigned)(8 * chunks), (unsigned)((count + 7) / 8)), dim3(kThreads), 0, s, (const double*)r.stage, r.stage_ld,
                           (const int32_t*)b.pr.perm, P, dst, ldb, l0, count, r.rank, r.world);
    }
    DPR_HIP(hipGetLastError());
    DPR_HIP(hipStreamSynchronize(s));      // (d_src is rewritten by the next build; the host vector may change)
    return DPR_OK;
}

static int njr_ensure_buffers(NjBuffers& b, hipStream_t s)
{
    NjRowShard& r = b.rs;
    if (b.twin_rows <= 0 || !b.D) { set_error("njr: the matrix buffers of this rank were not allocated for the row-sharded pruned plan"); return DPR_ERR_STATE; }
    r.lay = njr_layout(b.N, r.world);
    r.half[0] = b.D;
    r.half[1] = reinterpret_cast<double*>(reinterpret_cast<char*>(b.D) + b.half_bytes);
    if (r.plan == kNjrCollective && !r.rows_plain) {
        DPR_HIP(hipMalloc(&r.rows_plain, sizeof(double) * (size_t)(2 * r.world) * (size_t)r.lay.slice));
        DPR_HIP(hipMemsetAsync(r.rows_plain, 0, sizeof(double) * (size_t)(2 * r.world) * (size_t)r.lay.slice, s));
    }
    if (!r.d_region) {
        DPR_HIP(hipMalloc(&r.d_region, sizeof(char*) * kNjsMaxWorld));
        DPR_HIP(hipMalloc(&r.d_src, sizeof(double*) * kNjsMaxWorld));
        DPR_HIP(hipMalloc(&r.ticket, 2 * kNjsTicketBytes));
        DPR_HIP(hipMemsetAsync(r.ticket, 0, 2 * kNjsTicketBytes, s));
    }
    {
        // every rank's window region and epoch buffers as this process sees them (njs_set_peers recorded the mappings)
        if ((int)b.peer.h_D.size() != r.world || (r.plan == kNjrMailbox && (int)b.peer.h_win.size() != r.world)) {
            set_error("njr: the peers' buffers are not mapped on this rank"); return DPR_ERR_STATE;
        }
        std::vector<char*> regions((size_t)r.world, nullptr);
        for (int h = 0; h < 2; ++h) r.peer_half[h].assign((size_t)r.world, nullptr);
        for (int k = 0; k < r.world; ++k) {
            if ((int)b.peer.h_win.size() == r.world && b.peer.h_win[(size_t)k]) regions[(size_t)k] = b.peer.h_win[(size_t)k] + r.win_off;
            r.peer_half[0][(size_t)k] = b.peer.h_D[(size_t)k];
            r.peer_half[1][(size_t)k] = reinterpret_cast<double*>(reinterpret_cast<char*>(b.peer.h_D[(size_t)k]) + b.half_bytes);
        }
        DPR_HIP(hipMemcpyAsync(r.d_region, regions.data(), sizeof(char*) * (size_t)r.world, hipMemcpyHostToDevice, s));
        DPR_HIP(hipStreamSynchronize(s));
    }
    if (!r.stage) {
        // staging rows: ~256 MB, at least 16 rows
        r.stage_ld = (b.N + kTileCols + 16 + 15) / 16 * 16;
        int64_t rows = (int64_t)(256ll << 20) / (r.stage_ld * 8);
        rows = rows < 16 ? 16 : (rows > 1024 ? 1024 : rows);
        r.stage_rows = rows;
        DPR_HIP(hipMalloc(&r.stage, sizeof(double) * (size_t)(r.stage_rows * r.stage_ld)));
    }
    return DPR_OK;
}

// Build epoch 0 on the ranks this process holds (all of them: virtual ranks; one: a process rank).  On entry every rank's
// first half (NjBuffers::D) holds its rows of the tip-order matrix (block-cyclic by kRowBlock rows) and NjBuffers::U all N
// row sums (replicated); the peers' buffers are mapped (NjRowShard::peer_half).
int njr_build(std::vector<NjBuffers*>& ranks, hipStream_t s)
{
    NjBuffers& b0 = *ranks[0];
    const int64_t N = b0.N;
    std::vector<double> hU((size_t)N);
    DPR_HIP(hipMemcpyAsync(hU.data(), b0.U, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, s));
    DPR_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> perm((size_t)N);
    std::iota(perm.begin(), perm.end(), 0);
    njp_rs_sort_by_row_sum(perm, hU);
    for (NjBuffers* pb : ranks) {
        NjBuffers& b = *pb;
        if (int rc = njr_ensure_buffers(b, s)) return rc;
        b.pr.scan_grid = njp_scan_grid_default();
        if (const char* e = std::getenv("DPR_NJ_GRAPH_ITERS")) { const int v = std::atoi(e); if (v >= 1 && v <= 4096) b.pr.graph_iters = v; }
        DPR_HIP(hipMemsetAsync(b.rs.half[1], 0, b.half_bytes, s));
        if (int rc = njp_rs_epoch(b.pr, N, N, b.rs.half[1], 0, b.rs.rank, b.rs.world, nullptr, s)) return rc;
        b.pr.utot0 = b.pr.utot;
        DPR_HIP(hipMemcpyAsync(b.pr.perm, perm.data(), sizeof(int32_t) * (size_t)N, hipMemcpyHostToDevice, s));
    }
    if (int rc = njr_barrier(ranks, s)) return rc;          // every rank's tip-order rows are complete
    for (NjBuffers* pb : ranks) {
        NjBuffers& b = *pb;
        if (int rc = njr_permute(b, 0, b.ld, kRowBlock, N, b.pr.D, b.pr.ld, N, s)) return rc;
        if (int rc = njp_rs_init_vectors(b.pr, b.U, nullptr, N, N, 0, s)) return rc;
    }
    DPR_HIP(hipStreamSynchronize(s));                        // `perm` goes out of scope
    return njr_barrier(ranks, s);                            // nobody overwrites the tip-order rows (an odd epoch's buffer) before all pulls are done
}

static int njr_rebuild_epoch(std::vector<NjBuffers*>& ranks, hipStream_t s, bool* rebuilt)
{
    NjBuffers& b0 = *ranks[0];
    *rebuilt = false;
    NjState st;
    DPR_HIP(hipMemcpy(&st, b0.st, sizeof(NjState), hipMemcpyDeviceToHost));
    const int64_t n = st.n, Pold = b0.pr.P;
    if (st.status != 0 || n < 3) return DPR_OK;
    std::vector<double> hU((size_t)Pold);
    std::vector<int32_t> hslot((size_t)Pold);
    DPR_HIP(hipMemcpy(hU.data(), b0.pr.U + (st.it & 1) * b0.pr.vstride, sizeof(double) * (size_t)Pold, hipMemcpyDeviceToHost));
    DPR_HIP(hipMemcpy(hslot.data(), b0.pr.slot_of_pos, sizeof(int32_t) * (size_t)Pold, hipMemcpyDeviceToHost));
    std::vector<int32_t> perm;
    perm.reserve((size_t)n);
    for (int64_t p = 0; p < Pold; ++p)
        if (hslot[(size_t)p] >= 0) perm.push_back((int32_t)p);
    if ((int64_t)perm.size() != n) { set_error("njr_rebuild_epoch: live positions do not match the active size"); return DPR_ERR_STATE; }
    njp_rs_sort_by_row_sum(perm, hU);
    if (int rc = njr_barrier(ranks, s)) return rc;          // every rank's finish kernel (the last new node's row) is done
    const int e = b0.pr.epoch_index + 1;
    const int src_half = (b0.pr.epoch_index + 1) & 1, dst_half = (e + 1) & 1;      // epoch e lives in half (e + 1) & 1
    for (NjBuffers* pb : ranks) {
        NjBuffers& b = *pb;
        const NjPruned old = b.pr;          // the old epoch's pointers (read by the kernels below; its slab is the other one)
        DPR_HIP(hipMemsetAsync(b.rs.half[dst_half], 0, b.half_bytes, s));
        if (int rc = njp_rs_epoch(b.pr, n, b.N, b.rs.half[dst_half], e, b.rs.rank, b.rs.world, nullptr, s)) return rc;
        b.pr.utot0 = old.utot0; b.pr.scan_grid = old.scan_grid; b.pr.graph_iters = old.graph_iters;
        DPR_HIP(hipMemcpyAsync(b.pr.perm, perm.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
        if (int rc = njr_permute(b, src_half, old.ld, kNjrChunk, Pold, b.pr.D, b.pr.ld, n, s)) return rc;
        if (int rc = njp_rs_init_vectors(b.pr, old.U + (st.it & 1) * old.vstride, old.slot_of_pos, n, n, st.it, s)) return rc;
        NjState st2 = st;
        st2.pnew[0] = -1; st2.pnew[1] = -1;
        for (auto& c : st2.cnt_list) c = 0ull;
        DPR_HIP(hipMemcpyAsync(b.st, &st2, sizeof(NjState), hipMemcpyHostToDevice, s));
        DPR_HIP(hipStreamSynchronize(s));
    }
    if (int rc = njr_barrier(ranks, s)) return rc;          // all pulls out of the old epoch are done before anybody goes on
    *rebuilt = true;
    return DPR_OK;
}

// one iteration on every rank held here
static const char* const kNjrKernelNames[] = { "njp_scan_kernel<rs>", "(other ranks' launches / exchange)", "njr_extract_kernel", "(other ranks' launches / exchange)",
                                               "njp_post_kernel<64,1,rs>", "" };
static int njr_iteration(std::vector<NjBuffers*>& ranks, hipStream_t s, bool sample)
{
    NjBuffers& b0 = *ranks[0];
    NjKernelTiming* kt = b0.kt;
    // timing samples: the launches of the FIRST rank held here are bracketed by events (with virtual ranks the other ranks'
    // launches of a phase and the copies that stand in for the collectives fall into the intervals in between)
    auto mark = [&](size_t r) -> int {
        if (!sample || r != 0) return DPR_OK;
        hipEvent_t e = nullptr;
        DPR_HIP(hipEventCreate(&e));
        kt->ev.push_back(e);
        DPR_HIP(hipEventRecord(e, s));
        return DPR_OK;
    };
    if (sample) { kt->nk = 5; njp_set_kernel_names(kNjrKernelNames); }
    for (size_t r = 0; r < ranks.size(); ++r) {
        if (int rc = mark(r)) return rc;
        if (int rc = njp_rs_launch_scan(njr_args(*ranks[r]), s)) return rc;
        if (int rc = mark(r)) return rc;
    }
    if (b0.rs.plan == kNjrCollective) { if (int rc = b0.rs.gather(b0.rs.cb_ctx, 0, s)) return rc; ++b0.rs.collectives; }
    for (size_t r = 0; r < ranks.size(); ++r) {
        if (int rc = mark(r)) return rc;
        if (int rc = njr_launch_extract(*ranks[r], s)) return rc;
        if (int rc = mark(r)) return rc;
    }
    if (b0.rs.plan == kNjrCollective) { if (int rc = b0.rs.gather(b0.rs.cb_ctx, 1, s)) return rc; ++b0.rs.collectives; }
    for (size_t r = 0; r < ranks.size(); ++r) {
        if (int rc = mark(r)) return rc;
        if (int rc = njp_rs_launch_post(njr_args(*ranks[r]), ranks[r]->N, s)) return rc;
        if (int rc = mark(r)) return rc;
    }
    b0.rs.launches += 3;
    return DPR_OK;
}

static int njr_run_segment(std::vector<NjBuffers*>& ranks, int64_t it0, int64_t todo, hipStream_t s)
{
    const int64_t limit = it0 + todo;
    for (NjBuffers* pb : ranks) DPR_HIP(hipMemcpyAsync(&pb->st->it_limit, &limit, sizeof(int64_t), hipMemcpyHostToDevice, s));
    DPR_HIP(hipStreamSynchronize(s));
    if (todo <= 0) return DPR_OK;
    for (NjBuffers* pb : ranks) {
        if (!pb->pr.fresh) continue;
        if (int rc = njp_rs_launch_list_all(njr_args(*pb), s)) return rc;
        pb->pr.fresh = false;
    }
    NjBuffers& b0 = *ranks[0];
    const bool timing = b0.kt && b0.kt->stride > 0;
    // Mailbox plan on a rank of its own: nothing between the kernels involves the host, so graph_iters iterations are captured
    // once per epoch and replayed (as njp.hip does); virtual ranks and the collective plan launch eagerly.
    const int gi = b0.pr.graph_iters;
    const bool use_graph = ranks.size() == 1 && b0.rs.plan == kNjrMailbox && todo >= gi && !timing;
    if (use_graph && !b0.pr.graph) {
        hipGraph_t g = nullptr;
        DPR_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = DPR_OK;
        for (int k = 0; k < gi && rc == DPR_OK; ++k) rc = njr_iteration(ranks, s, false);
        hipError_t e = hipStreamEndCapture(s, &g);
        if (rc != DPR_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
        e = hipGraphInstantiate(&b0.pr.graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { b0.pr.graph = nullptr; return hip_fail(e, "hipGraphInstantiate"); }
        b0.rs.launches -= 3 * gi;      // (captured, not launched)
    }
    int64_t done = 0;
    if (use_graph)
        for (; done + gi <= todo; done += gi) { DPR_HIP(hipGraphLaunch(b0.pr.graph, s)); b0.rs.launches += 3 * gi; }
    for (; done < todo; ++done)
        if (int rc = njr_iteration(ranks, s, timing && (it0 + done) % b0.kt->stride == 0)) return rc;
    for (NjBuffers* pb : ranks)
        if (int rc = njp_rs_launch_finish(njr_args(*pb), s)) return rc;
    return DPR_OK;
}

// enqueue `todo` iterations starting at it0, in epochs (njp_run's rule: rebuild once the active size is down to pct % of the
// epoch's positions).  The adaptive hand-over to the streaming loop is a single-GPU plan; here the pruned loop runs throughout.
int njr_run(std::vector<NjBuffers*>& ranks, int64_t it0, int64_t todo, hipStream_t s)
{
    const char* e_min = std::getenv("DPR_NJ_EPOCH_MIN");
    const int64_t epoch_min = e_min ? std::atoll(e_min) : 2048;
    const int64_t pct = 80;
    NjBuffers& b0 = *ranks[0];
    int64_t it = it0, left = todo;
    if (left <= 0) return njr_run_segment(ranks, it0, 0, s);
    while (left > 0) {
        const int64_t n = b0.N - it, P = b0.pr.P;
        int64_t seg = left;
        if (epoch_min > 0 && P >= epoch_min) {
            const int64_t target = P * pct / 100;
            if (n <= target && n >= 3) {
                DPR_HIP(hipStreamSynchronize(s));
                bool rebuilt = false;
                if (int rc = njr_rebuild_epoch(ranks, s, &rebuilt)) return rc;
                if (!rebuilt) return DPR_OK;        // no candidate left: every queued kernel is a no-op, dpr_nj_run reports it
                continue;
            }
            if (n - target < seg) seg = n - target;
        }
        if (int rc = njr_run_segment(ranks, it, seg, s)) return rc;
        it += seg; left -= seg;
    }
    // every rank's finish kernel (it stores the last new node's row) is done before anybody reads another rank's rows -- the
    // final pair's distance in dpr_nj_run, the test hooks -- whichever way the ranks were joined (advisor, round 5: process ranks
    // without RCCL had no barrier here)
    return njr_barrier(ranks, s);
}

}  // namespace dpr
