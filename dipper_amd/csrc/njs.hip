// Row-sharded streaming NJ with ONE exchange and TWO launches per iteration (several GPUs; the layout north_star
// names: the N x N matrix as row blocks over the ranks, a full Q-argmin scan of the own rows every iteration).
// Replaces the loop src/neighborJoining.cu:211-243 of the (single-GPU) reference; same arithmetic, keys and merge log as
// nj.hip's single-GPU streaming loop, bit for bit.
//
// Round 2's sharded loop (nj.hip: scan, select, all-gather of the records, commit + extract, all-gather of three column
// slices, update = 4 launches + 2 collectives) paid ~60 us per iteration on top of 594 / G us of scan at 30 000 tips.
// Here an iteration is
//     SCAN(it)   own rows of the strict lower triangle; the LAST block to finish (ticket) reduces the block records to the
//                rank's record and publishes it: into the local record array (an RCCL all-gather follows: plan PEER), or
//                straight into every rank's mailbox over xGMI (plan MAILBOX: no collective launch at all);
//     POST(it)   every rank reduces the G records to the winner (x, y, d), PULLS rows x and y (and the owner of y: row n-1)
//                from their owners' memory -- the only O(n) data an iteration moves, 2 x 8n bytes per rank --, and does the
//                whole update: replicated row sums and keys, the columns x / y of its own rows, the new rows x / y.
// Why rows can be pulled without a second synchronisation: the owners do NOT overwrite rows x and y in place.  The new
// rows go to row buffers R[it & 1] in the owner's peer-visible window; everybody reads a slot's row through a ROW VIEW
// (the buffer of the previous merge if the slot was rewritten by it, else the matrix row), and the owner flushes the
// buffers of merge it - 1 into the matrix during POST(it) -- behind exchange(it), which every rank passes only after it
// has finished POST(it - 1), i.e. its pulls of iteration it - 1.  Rows x, y and n - 1 of the CURRENT iteration receive
// no in-place write at all during POST(it), so concurrent pulls of them are safe.
// Peer memory: every rank maps the other ranks' matrix and window (hipIpc handles: across processes; plain pointers for
// the virtual ranks of the single-GPU validation mode).  Pulls use system-scope loads; the mailbox words are
// fine-grained memory written and polled with system-scope release / acquire, every poll bounded by a wall-clock limit
// (a rank that never answers makes the run end with DPR_ERR_COMM, not hang).
#include "nj_dev.hpp"

#include <cstdio>
#include <cstdlib>

namespace dpr {

__device__ __forceinline__ double ld_sys_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// Flag words of the windows.  No acquire / release FENCES (an acquire is an L2 invalidate after every poll, a release an L2
// write-back): everything that travels behind a flag is written with system-scope write-through stores and read with
// system-scope loads, which bypass the non-coherent caches on both sides; the writer drains its stores (vmcnt) before the
// flag store, the reader issues its data loads after the poll returned (same thread, address dependency through control).
__device__ __forceinline__ unsigned long long ld_acq_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_rel_u64(unsigned long long* p, unsigned long long v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// stores that other DEVICES read after the next exchange (matrix rows / columns, row buffers): write-through at system scope,
// so nothing of them is left dirty in this device's (per-XCD) L2 when the kernel ends -- a release fence per block instead
// wrote the whole L2 back every iteration
__device__ __forceinline__ void st_sys_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// arguments shared by the kernels of the loop
struct NjsArgs {
    double* D; int64_t ld; NjState* st;
    double* U; double* Ur; uint64_t* KA; double* xpart;
    NjRecord* partials; NjsRec* recs;            // block records of the scan; gathered rank records (plan PEER)
    unsigned int* ticket;                        // [1 + kNjsTicketGroups] counters, 128 bytes apart: the top word, then one per group
    char* const* win;                            // [world] every rank's window (own one included), valid in this process
    double* const* peerD;                        // [world] every rank's matrix rows
    NjsLayout lay;
    int64_t n, it;                               // active size and iteration index of THIS launch
    int has_pending;                             // the rows of merge it - 1 still live in the row buffers R[(it - 1) & 1]
    int rank, world, plan;
    int nparts;                                  // scan grid
    unsigned long long poll_ticks;               // bound of a mailbox poll (100 MHz wall clock)
    unsigned long long seq_base;                 // run id << 32: sequence numbers are unique across the runs of a context
    int64_t fault_it; int fault_rank;            // test hook (dpr_ctx_set_debug_fault): that rank corrupts one pulled element there
    int32_t* log_x; int32_t* log_y; double* log_bx; double* log_by;
};

__device__ __forceinline__ double* win_row(char* w, const NjsLayout& lay, int which, int parity)
{
    return reinterpret_cast<double*>(w + lay.off_rows) + (int64_t)(which * 2 + parity) * lay.ldv;      // which: 0 = x row, 1 = y row
}
__device__ __forceinline__ NjsRec* win_mail(char* w, int parity, int r) { return reinterpret_cast<NjsRec*>(w) + parity * kNjsMaxWorld + r; }

// row of slot a as it stands BEFORE merge `it` (see the header): pointer valid in this process, owner's memory
__device__ __forceinline__ const double* row_view(const NjsArgs& a, int64_t slot, int64_t xp, int64_t yp)
{
    const int o = shard_owner(slot, a.world);
    if (slot == xp) return win_row(a.win[o], a.lay, 0, (int)((a.it - 1) & 1));
    if (slot == yp) return win_row(a.win[o], a.lay, 1, (int)((a.it - 1) & 1));
    return a.peerD[o] + shard_local_row(slot, a.world) * a.ld;
}

// ------------------------------------------------------------------------------------------------
// SCAN(it): nj_scan_kernel's unit walk over the own rows (rows rewritten by the previous merge come from the row
// buffers), then the last block to finish reduces the block records and publishes the rank's record.
// ------------------------------------------------------------------------------------------------
// (The streamed arrays are DIRECT kernel parameters, const __restrict__: only then does the compiler keep the rows' parameters
//  -- Ur[a], KA[a], uniform per row -- in scalar loads.  With every pointer inside the by-value struct the same loop ran 2.25 x
//  slower: 189 instead of 84 us for an eighth of the 30 000-tip triangle.  a.Ur aliases Ur: block 0 stores the single element
//  Ur[xprev] through it, and no block ever USES Ur[xprev] read through the const pointer -- every use substitutes urx.)
template <int RG, bool NT>
__global__ __launch_bounds__(kThreads) void njs_scan_kernel(const double* __restrict__ D, const double* __restrict__ Ur,
                                                            const uint64_t* __restrict__ KA, const double* __restrict__ xrow_p,
                                                            const double* __restrict__ yrow_p, NjsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* pref = reinterpret_cast<int32_t*>(smem);
    __shared__ double sd[kThreads];
    __shared__ double sq[kThreads / 64], sdd[kThreads / 64];
    __shared__ uint64_t sk[kThreads / 64];
    __shared__ unsigned int s_last;

    const int tid = threadIdx.x;
    const int64_t n = a.n, it = a.it;
    const int rank = a.rank, world = a.world;
    double bq = 10000.0;
    uint64_t bk = ~0ull;
    // one round trip: the state words and this thread's first chunk partial of the row sum (xpart holds >= 512 entries)
    const int st_status = a.st->status;
    const int32_t st_x = a.st->x, st_y = a.st->y;
    const double xp0 = a.xpart[tid];
    const bool dead = st_status != 0;

    int64_t xprev = -1;
    double urx = 0.0, ux = 0.0;
    RowView rv;
    if (it > 0 && !dead) {
        xprev = st_x;
        ux = finish_ux_bcast_pre(xp0, a.xpart, n + 1, sd);
        urx = ux / (double)(n - 2);
        if (blockIdx.x == 0 && tid == 0) { a.U[xprev] = ux; a.Ur[xprev] = urx; }
        if (a.has_pending) {
            rv.xp = xprev; rv.yp = st_y;
            rv.xrow = xrow_p;       // this rank's row buffers of merge it - 1 (host: window base + row offset)
            rv.yrow = yrow_p;
        }
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
                kb0 = nj_key_b(b0); kb1 = nj_key_b(b0 + 1);
                fresh = false;
            }
            const int64_t l0 = lstart + (int64_t)g * RG;
            const int nrows = (int)min((int64_t)RG, nloc - l0);
            const int64_t a0 = shard_global_row(l0, rank, world);
            if (a0 < c0 + kTileCols)
                scan_rows<true, NT, false, true>(D, a.ld, Ur, KA, nullptr, a0, l0, nrows, c0, xprev, urx, ub0, ub1, ka0, ka1, kb0, kb1, bq, bk, rv);
            else
                scan_rows<false, NT, false, true>(D, a.ld, Ur, KA, nullptr, a0, l0, nrows, c0, xprev, urx, ub0, ub1, ka0, ka1, kb0, kb1, bq, bk, rv);
            ++g;
        }
    }

    block_best(bq, bk, sq, sk);
    if (tid == 0) {
        NjRecord rec;
        rec.q = bq; rec.key = bk; rec.d = 0.0; rec.pad = 0;
        if (bk != ~0ull) {
            // d = D[max][min]: the block only visited own rows a > b, so the row (or its buffer) is local
            const int64_t i = (int64_t)(bk & 0xFFFFFFull), j = (int64_t)((bk >> 24) & 0xFFFFFFull);
            const int64_t x = i < j ? i : j, y = i < j ? j : i;
            const double* row = (y == rv.xp) ? rv.xrow : (y == rv.yp) ? rv.yrow : D + shard_local_row(y, world) * a.ld;
            rec.d = row[x];
        }
        // the record must be visible to the block that reduces (possibly on another XCD): agent-scope stores, then the ticket
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.partials + blockIdx.x);
        __hip_atomic_store(dst + 0, (unsigned long long)__double_as_longlong(rec.q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, (unsigned long long)rec.key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 2, (unsigned long long)__double_as_longlong(rec.d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (NOT a release fence: on this chip a device-scope release is an L2 write-back per block, which made this kernel 2.2 x
        //  slower -- 183 instead of 84 us for an eighth of the 30 000-tip triangle.  The three stores above are write-through
        //  (sc1); draining them before a RELAXED ticket increment orders them for the reader, whose loads are sc1 too.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Two-level ticket.  The units are split evenly, so the blocks of a launch finish within a microsecond of each other, and
        // an atomic with return on ONE word costs 3 us with 256 blocks on it and 11.7 us with 1 024 (tools/lat_probe): a launch
        // over 4 MB spent 23 us, mostly here.  Block b takes a ticket of group b % kNjsTicketGroups (one 128-byte line each);
        // the last of a group takes one of the top word; the last of those reduces.  Every increment is issued after the
        // block's own stores are drained and after the increments it has seen, so the chain orders all records for the reader.
        const unsigned int G = gridDim.x, grp = blockIdx.x % kNjsTicketGroups;
        const unsigned int in_grp = (G - grp + kNjsTicketGroups - 1) / kNjsTicketGroups, groups = G < kNjsTicketGroups ? G : kNjsTicketGroups;
        unsigned int last = 0u;
        const unsigned int t = __hip_atomic_fetch_add(a.ticket + 32 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == in_grp - 1) {
            const unsigned int tt = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (tt == groups - 1) ? 1u : 0u;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // ---- last block of the launch: the rank's record
    double wq = 10000.0, wd = 0.0;
    uint64_t wk = ~0ull;
    for (int i = tid; i < (int)gridDim.x; i += kThreads) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.partials + i);
        const double q = __longlong_as_double((long long)__hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const uint64_t k = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double d = __longlong_as_double((long long)__hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if ((q < wq) | ((q == wq) & (k < wk))) { wq = q; wk = k; wd = d; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oq = __shfl_down(wq, off, 64);
        const uint64_t ok = __shfl_down((unsigned long long)wk, off, 64);
        const double od = __shfl_down(wd, off, 64);
        if ((oq < wq) | ((oq == wq) & (ok < wk))) { wq = oq; wk = ok; wd = od; }
    }
    __syncthreads();          // (sq / sk of block_best are reused)
    if ((tid & 63) == 0) { sq[tid >> 6] = wq; sk[tid >> 6] = wk; sdd[tid >> 6] = wd; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kThreads / 64; ++w)
            if ((sq[w] < wq) | ((sq[w] == wq) & (sk[w] < wk))) { wq = sq[w]; wk = sk[w]; wd = sdd[w]; }
        if (dead) { wq = 10000.0; wk = ~0ull; wd = 0.0; }
    }
    if (tid <= kNjsTicketGroups) a.ticket[32 * tid] = 0u;      // next launch (stream order)
    // the record also carries this rank's view of the replicated state (NjsRec: the row sum of the node of merge it - 1, bit
    // for bit, and the rank's status), which POST compares across the ranks
    const unsigned long long uxb = (unsigned long long)__double_as_longlong(ux);
    const unsigned long long stw = (unsigned long long)(unsigned int)st_status;      // (nobody changes the status during a scan launch)
    if (a.plan == kNjsMailbox) {
        // thread r sends the record to rank r's mailbox (own one included): data words, then the sequence word with
        // release semantics; the reader acquires on the sequence word.
        // A rank that has failed announces it ONCE, with the record of the first scan after the failure.  Its post kernels
        // return at once, so nothing paces it any more: were it to keep sending, its record of iteration it + 2 would
        // overwrite the one of iteration it (same mailbox line) that a slower rank may not have read yet -- that rank
        // then ended with "record did not arrive" instead of the failure the others report.
        const bool announced = dead && a.st->pad != 0;
        if (tid == 0) { sq[0] = wq; sk[0] = wk; sdd[0] = wd; }
        __syncthreads();
        if (dead && tid == 0) a.st->pad = 1;
        if (tid < world && !announced) {
            NjsRec* m = win_mail(a.win[tid], (int)(it & 1), rank);
            unsigned long long* w = reinterpret_cast<unsigned long long*>(m);
            st_sys_u64(w + 0, (unsigned long long)__double_as_longlong(sq[0]));
            st_sys_u64(w + 1, (unsigned long long)sk[0]);
            st_sys_u64(w + 2, (unsigned long long)__double_as_longlong(sdd[0]));
            st_sys_u64(w + 4, uxb);
            st_sys_u64(w + 5, stw);
            st_rel_u64(w + 3, a.seq_base + (unsigned long long)(it + 1));
        }
    } else if (tid == 0) {
        NjsRec rec;
        rec.q = wq; rec.key = wk; rec.d = wd; rec.seq = a.seq_base + (uint64_t)(it + 1);
        rec.ux = uxb; rec.status = stw; rec.pad[0] = 0; rec.pad[1] = 0;
        a.recs[rank] = rec;
    }
}

// ------------------------------------------------------------------------------------------------
// POST(it): winner of the G records, pulls of rows x / y (/ n-1), the whole update (see the header).  Thread j = slot j.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void njs_post_kernel(NjsArgs a)
{
    __shared__ double s[kThreads];
    __shared__ double s_q[kNjsMaxWorld], s_d[kNjsMaxWorld];
    __shared__ uint64_t s_k[kNjsMaxWorld];
    __shared__ unsigned long long s_ux[kNjsMaxWorld];
    __shared__ int s_fail, s_dead;
    const int tid = threadIdx.x;
    const int64_t n = a.n, it = a.it;
    const int rank = a.rank, world = a.world;
    if ((int64_t)blockIdx.x * kThreads >= n) return;
    // (one read per block: another block of this launch may be storing a failure status right now, and the waves of a
    //  block must agree before the barriers below)
    if (tid == 0) { s_fail = 0; s_dead = a.st->status != 0 ? 1 : 0; }
    __syncthreads();
    if (s_dead) return;
    // ---- the G rank records of this iteration
    if (tid < world) {
        unsigned long long their_status = 0ull;
        if (a.plan == kNjsMailbox) {
            const unsigned long long* w = reinterpret_cast<const unsigned long long*>(win_mail(a.win[rank], (int)(it & 1), tid));
            const unsigned long long t0 = wall_clock64();
            bool ok = true;
            while (ld_acq_u64(w + 3) != a.seq_base + (unsigned long long)(it + 1)) {
                if (wall_clock64() - t0 > a.poll_ticks) { ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) s_fail = 3;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (compiler ordering only: the record words are loaded after the poll)
            s_q[tid] = __longlong_as_double((long long)__hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            s_k[tid] = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            s_d[tid] = __longlong_as_double((long long)__hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
            s_ux[tid] = __hip_atomic_load(w + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            their_status = ok ? __hip_atomic_load(w + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
        } else {
            s_q[tid] = a.recs[tid].q; s_k[tid] = a.recs[tid].key; s_d[tid] = a.recs[tid].d;
            s_ux[tid] = a.recs[tid].ux;
            their_status = a.recs[tid].status;
            if (a.recs[tid].seq != a.seq_base + (uint64_t)(it + 1)) s_fail = 3;       // a record of another iteration: the exchange did not happen
        }
        // a rank that has already failed says so in its record: fail the same way (not with "no candidate")
        if (their_status == 3ull || their_status == 4ull) s_fail = (int)their_status;
    }
    __syncthreads();
    // every rank derived the row sum of the node of merge it - 1 from the rows it pulled: the words must agree bit for bit
    if (tid < world && s_fail == 0 && s_ux[tid] != s_ux[rank]) s_fail = 4;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kThreads + tid;
    const int64_t last = n - 1;
    if (s_fail) {
        // 3: exchange failed (mailbox poll timed out / stale record); 4: the ranks' replicated row sums differ (a pulled
        // row was stale or torn) -- both DPR_ERR_COMM; st->q / st->d keep the two words for the message
        if (i == last) {
            a.st->status = s_fail;
            if (s_fail == 4) {
                int other = 0;
                for (int r = 0; r < world; ++r) if (s_ux[r] != s_ux[rank]) { other = r; break; }
                a.st->q = __longlong_as_double((long long)s_ux[rank]); a.st->d = __longlong_as_double((long long)s_ux[other]);
                a.st->x = rank; a.st->y = other;
            }
        }
        return;
    }
    double bq = 10000.0, d = 0.0;
    uint64_t bk = ~0ull;
    for (int r = 0; r < world; ++r) {
        const double q = s_q[r];
        const uint64_t k = s_k[r];
        if ((q < bq) | ((q == bq) & (k < bk))) { bq = q; bk = k; d = s_d[r]; }
    }
    if (bk == ~0ull || !(bq < 10000.0)) {     // (q == 10000.0 is no candidate: strict `<` of src/neighborJoining.cu:134-141)
        if (i == last) a.st->status = 1;
        return;
    }
    const int64_t ki = (int64_t)(bk & 0xFFFFFFull), kj = (int64_t)((bk >> 24) & 0xFFFFFFull);
    const int64_t x = ki < kj ? ki : kj, y = ki < kj ? kj : ki;
    const int64_t n1 = n - 1;
    const double r1 = (double)(n1 - 2);
    const bool own_x = shard_owner(x, world) == rank, own_y = shard_owner(y, world) == rank;
    // pending rows of merge it - 1 (their slots: the previous log entry)
    int64_t xp = -1, yp = -1;
    if (a.has_pending && it > 0) { xp = a.log_x[it - 1]; yp = a.log_y[it - 1]; }
    const double* __restrict__ rowx = row_view(a, x, xp, yp);
    const double* __restrict__ rowy = row_view(a, y, xp, yp);
    const double* __restrict__ rowl = row_view(a, last, xp, yp);        // read by the owner of y only
    double* __restrict__ RXn = win_row(a.win[rank], a.lay, 0, (int)(it & 1));
    double* __restrict__ RYn = win_row(a.win[rank], a.lay, 1, (int)(it & 1));

    double val = 0.0;
    if (i == last) commit_merge(a.st, a.U, n, it, x, y, d, bq, a.log_x, a.log_y, a.log_bx, a.log_by);   // reads U[y] before the tail rewrites it
    if (i < n && i != x && i != y) {
        double dxi = ld_sys_f64(rowx + i);
        const double dyi = ld_sys_f64(rowy + i);
        if (it == a.fault_it && rank == a.fault_rank && tid == 7 && blockIdx.x == 0) dxi = dxi * 0.5 + 1.0e-3;      // test hook: a "stale" pull
        val = (dxi + dyi - d) * 0.5;
        if (i != last) {
            const double u = a.U[i] + (-dxi - dyi + val);
            a.U[i] = u;
            a.Ur[i] = u / r1;
            if (own_x) st_sys_f64(RXn + i, val);
            if (own_y) st_sys_f64(RYn + i, ld_sys_f64(rowl + i));
            if (shard_owner(i, world) == rank && i != xp && i != yp) {
                // columns x and y of an own row that lives in the matrix (a pending row gets them with its flush below)
                double* row = a.D + shard_local_row(i, world) * a.ld;
                const double far = row[last];
                st_sys_f64(row + x, val);
                st_sys_f64(row + y, far);
            }
        } else {
            // tail of the reference (thread (0,0), src/neighborJoining.cu:184-193)
            const double uy = a.U[last] + (-dxi - dyi + val);
            a.U[y] = uy;
            a.Ur[y] = uy / r1;
            if (own_x) st_sys_f64(RXn + y, val);
            if (own_y) st_sys_f64(RYn + x, val);
        }
    } else if (i == x) {
        if (own_x) st_sys_f64(RXn + x, 0.0);          // diagonal
        if (own_y && y == last) st_sys_f64(RYn + x, 0.0);      // (y == last: no tail; the row dies with the slot, keep it defined)
    } else if (i == y) {
        if (own_y) st_sys_f64(RYn + y, 0.0);
        if (own_x && y == last) st_sys_f64(RXn + y, 0.0);
    }
    // ---- flush the row buffers of merge it - 1 into the matrix (their owner; a row consumed by this merge or dead now is
    // not flushed).  Column x / y of such a row are the values its own thread would have written.
    if (i < n) {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int64_t p = w ? yp : xp;
            if (p < 0 || p == x || p == y || p >= last || shard_owner(p, world) != rank) continue;
            const double* __restrict__ Rp = win_row(a.win[rank], a.lay, w, (int)((it - 1) & 1));
            double v = Rp[i];
            if (i == x) v = (ld_sys_f64(rowx + p) + ld_sys_f64(rowy + p) - d) * 0.5;
            else if (i == y) v = Rp[last];
            st_sys_f64(a.D + shard_local_row(p, world) * a.ld + i, v);
        }
    }
    if (i < n1) a.KA[i] = nj_key_a_dev(i, n1);
    const double cs = block_tree256(val, s);
    if (tid == 0) a.xpart[blockIdx.x] = cs;
}

// after the loop (behind a barrier over the ranks): flush the row buffers of the last merge, materialise U[x]
__global__ __launch_bounds__(kThreads) void njs_finish_kernel(NjsArgs a)
{
    __shared__ double s[kThreads];
    __shared__ int s_dead;
    const int64_t n = a.n, it = a.it;      // state after `it` iterations, n active
    if (threadIdx.x == 0) s_dead = a.st->status != 0 ? 1 : 0;      // (block-uniform, as in njs_post_kernel)
    __syncthreads();
    if (s_dead || it <= 0) return;
    const int64_t xp = a.st->x, yp = a.st->y;
    if (blockIdx.x == 0) {
        const double ux = finish_ux(a.xpart, n + 1, s);
        if (threadIdx.x == 0) { a.U[xp] = ux; a.Ur[xp] = ux / (double)(n - 2); }
    }
    if (!a.has_pending) return;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n + 1) return;               // (columns of the slot that died with the last merge included: harmless)
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const int64_t p = w ? yp : xp;
        if (p < 0 || p >= n || shard_owner(p, a.world) != a.rank) continue;
        const double* __restrict__ Rp = win_row(a.win[a.rank], a.lay, w, (int)((it - 1) & 1));
        st_sys_f64(a.D + shard_local_row(p, a.world) * a.ld + i, Rp[i]);
    }
}

// barrier over the ranks through the windows (plan MAILBOX, no RCCL): epoch numbers only grow
__global__ void njs_barrier_kernel(NjsArgs a, unsigned long long epoch)
{
    const int tid = threadIdx.x;
    if (tid >= a.world) return;
    unsigned long long* theirs = reinterpret_cast<unsigned long long*>(a.win[tid] + a.lay.off_bar) + 8 * a.rank;
    st_rel_u64(theirs, epoch);
    const unsigned long long* mine = reinterpret_cast<const unsigned long long*>(a.win[a.rank] + a.lay.off_bar) + 8 * tid;
    const unsigned long long t0 = wall_clock64();
    while (ld_acq_u64(mine) < epoch) {
        if (wall_clock64() - t0 > a.poll_ticks) { a.st->status = 3; break; }
        __builtin_amdgcn_s_sleep(4);
    }
}

// initial row sums: every rank left the sums of its own rows in its window's slice; U[i] = slice of owner(i)
__global__ __launch_bounds__(kThreads) void njs_unpack_u_kernel(NjsArgs a, int64_t N)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
    const double* sl = reinterpret_cast<const double*>(a.win[shard_owner(i, a.world)] + a.lay.off_slice);
    a.U[i] = ld_sys_f64(sl + shard_local_row(i, a.world));
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
NjsLayout njs_layout(int64_t N, int world, bool with_njr)
{
    NjsLayout l;
    const int64_t nblk = (N + kRowBlock - 1) / kRowBlock;
    l.slice_len = ((nblk + world - 1) / world) * kRowBlock;
    l.ldv = (N + kTileCols + 16 + 15) / 16 * 16;
    l.off_bar = (int64_t)sizeof(NjsRec) * 2 * kNjsMaxWorld;                   // mail: [2][kNjsMaxWorld] records
    l.off_slice = l.off_bar + 64 * kNjsMaxWorld;                              // bar: one 64-byte line per rank
    l.off_rows = (l.off_slice + (int64_t)sizeof(double) * l.slice_len + 255) / 256 * 256;
    l.bytes = l.off_rows + (int64_t)sizeof(double) * 4 * l.ldv;
    if (with_njr) {        // the region of the row-sharded pruned NJ behind the streaming loop's (njr.hip)
        l.off_njr = (l.bytes + 4095) / 4096 * 4096;
        l.bytes = l.off_njr + njr_layout(N, world).bytes;
    }
    return l;
}

static NjsArgs njs_args(NjBuffers& b, int64_t n, int64_t it, bool pending)
{
    NjsArgs a;
    a.D = b.D; a.ld = b.ld; a.st = b.st; a.U = b.U; a.Ur = b.Ur; a.KA = b.KA; a.xpart = b.xpart;
    a.partials = b.partials; a.recs = b.recs64; a.ticket = b.peer.ticket;
    a.win = b.peer.d_win; a.peerD = b.peer.d_D; a.lay = b.peer.lay;
    a.n = n; a.it = it; a.has_pending = pending ? 1 : 0;
    a.rank = b.rank; a.world = b.world; a.plan = b.peer.plan;
    // blocks of the scan: the single-GPU grid (2 048) by default; a rank streams only 1 / world of the triangle, so a
    // smaller grid may ramp and drain faster (profiles/njs_vworld_stats.py)
    // (measured with 8 virtual ranks at 30 000 tips, round 4: 85.1 / 83.6 / 85.2 / 88.5 / 87.7 us per rank and iteration with
    //  512 / 768 / 1 024 / 1 536 / 2 048 blocks)
    a.nparts = b.world >= 4 ? 768 : nj_scan_grid();
    a.poll_ticks = b.peer.poll_ticks;
    a.seq_base = b.peer.run_id << 32;
    a.fault_it = b.peer.fault_it; a.fault_rank = b.peer.fault_rank;
    a.log_x = b.log_x; a.log_y = b.log_y; a.log_bx = b.log_bx; a.log_by = b.log_by;
    return a;
}

int njs_launch_scan(NjBuffers& b, int64_t n, int64_t it, bool pending, hipStream_t s)
{
    const NjsArgs a = njs_args(b, n, it, pending);
    const size_t lds = sizeof(int32_t) * (size_t)((b.N + kTileCols - 1) / kTileCols + 2);
    const int par = (int)((it - 1) & 1);
    const double* rows = reinterpret_cast<const double*>(b.peer.win + b.peer.lay.off_rows);
    const double* xrow = rows + (int64_t)(0 * 2 + par) * b.peer.lay.ldv;
    const double* yrow = rows + (int64_t)(1 * 2 + par) * b.peer.lay.ldv;
    hipLaunchKernelGGL((njs_scan_kernel<16, true>), dim3((unsigned)a.nparts), dim3(kThreads), lds, s, (const double*)b.D, (const double*)b.Ur,
                       (const uint64_t*)b.KA, xrow, yrow, a);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int njs_launch_post(NjBuffers& b, int64_t n, int64_t it, bool pending, hipStream_t s)
{
    const NjsArgs a = njs_args(b, n, it, pending);
    hipLaunchKernelGGL(njs_post_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, a);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int njs_launch_finish(NjBuffers& b, int64_t n, int64_t it, bool pending, hipStream_t s)
{
    const NjsArgs a = njs_args(b, n, it, pending);
    hipLaunchKernelGGL(njs_finish_kernel, dim3((unsigned)((n + 1 + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, a);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int njs_launch_barrier(NjBuffers& b, hipStream_t s)
{
    const NjsArgs a = njs_args(b, 0, 0, false);
    hipLaunchKernelGGL(njs_barrier_kernel, dim3(1), dim3(kNjsMaxWorld), 0, s, a, ++b.peer.bar_epoch);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int njs_launch_unpack_u(NjBuffers& b, hipStream_t s)
{
    const NjsArgs a = njs_args(b, 0, 0, false);
    hipLaunchKernelGGL(njs_unpack_u_kernel, dim3((unsigned)((b.N + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, a, b.N);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// the window of a rank: fine-grained device memory (polled and written by other devices while kernels run)
int njs_alloc_window(NjBuffers& b, hipStream_t s)
{
    NjPeer& p = b.peer;
    const NjsLayout lay = njs_layout(b.N, b.world, b.twin_rows > 0);
    // (p.fault_it / p.fault_rank -- the test hook of the cross-check -- are set by dpr_ctx_set_debug_fault only: no environment
    //  variable can make a production run corrupt a pulled element)
    if (p.win && p.lay.bytes == lay.bytes) {
        // Reuse (same shape again): nothing in the window is cleared -- another rank may already be ahead of this one and
        // writing into it.  Mail sequence numbers carry the run id, barrier epochs only grow, the slice and the row
        // buffers are written before they are read.
        DPR_HIP(hipMemsetAsync(p.ticket, 0, kNjsTicketBytes, s));
        ++p.run_id;
        return DPR_OK;
    }
    njs_free_window(b);
    p.lay = lay;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&p.win), (size_t)lay.bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return hip_fail(e, "hipExtMallocWithFlags(peer window, fine-grained)");
    DPR_HIP(hipMemsetAsync(p.win, 0, (size_t)lay.bytes, s));
    DPR_HIP(hipMalloc(&p.ticket, kNjsTicketBytes));
    DPR_HIP(hipMemsetAsync(p.ticket, 0, kNjsTicketBytes, s));
    DPR_HIP(hipMalloc(&p.d_win, sizeof(char*) * kNjsMaxWorld));
    DPR_HIP(hipMalloc(&p.d_D, sizeof(double*) * kNjsMaxWorld));
    DPR_HIP(hipMemsetAsync(p.d_win, 0, sizeof(char*) * kNjsMaxWorld, s));
    DPR_HIP(hipMemsetAsync(p.d_D, 0, sizeof(double*) * kNjsMaxWorld, s));
    p.bar_epoch = 0;
    p.run_id = 1;
    p.attached = false;
    return DPR_OK;
}

void njs_free_window(NjBuffers& b)
{
    NjPeer& p = b.peer;
    for (void* m : p.opened) (void)hipIpcCloseMemHandle(m);
    p.opened.clear();
    if (p.win) (void)hipFree(p.win);
    if (p.ticket) (void)hipFree(p.ticket);
    if (p.d_win) (void)hipFree(p.d_win);
    if (p.d_D) (void)hipFree(p.d_D);
    const int plan = p.plan;
    const unsigned long long ticks = p.poll_ticks;
    const int64_t fi = p.fault_it; const int fr = p.fault_rank;
    p = NjPeer();
    p.plan = plan; p.poll_ticks = ticks; p.fault_it = fi; p.fault_rank = fr;
}

// pointers of all ranks as seen from this process: wins[r], Ds[r] (host arrays of `world` entries)
int njs_set_peers(NjBuffers& b, char* const* wins, double* const* Ds, hipStream_t s)
{
    NjPeer& p = b.peer;
    DPR_HIP(hipMemcpyAsync(p.d_win, wins, sizeof(char*) * (size_t)b.world, hipMemcpyHostToDevice, s));
    DPR_HIP(hipMemcpyAsync(p.d_D, Ds, sizeof(double*) * (size_t)b.world, hipMemcpyHostToDevice, s));
    DPR_HIP(hipStreamSynchronize(s));      // the host arrays may be temporaries
    p.h_D.assign(Ds, Ds + b.world);
    p.h_win.assign(wins, wins + b.world);
    p.attached = true;
    return DPR_OK;
}

}  // namespace dpr
