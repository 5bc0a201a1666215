// Kernel arguments and small device helpers of the pruned NJ path, shared by njp.hip (single GPU, unit-sharded) and njr.hip
// (row-sharded).  The kernels themselves live in njp.hip.
#pragma once
#include "nj_dev.hpp"

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
inline uint64_t enc_f64_host(double x)
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


// Latency is what matters in the kernels of the loop (a few hundred KB of data per iteration): every kernel issues
// all of its global loads in as few dependent hops as possible.

// arguments shared by the kernels of the loop (by value: one kernarg block)
struct NjpArgs {
    double* D; int64_t ld; NjState* st;
    double* U; double* R; int64_t vstride;      // U, R: [2][vstride]
    double* Ur; uint64_t* KA; uint64_t* KB; int32_t* slot_of_pos; int32_t* pos_of_slot;
    double* xpart; NjRecord* partials; unsigned long long* umin;
    int64_t P;
    const int32_t* blk_cb; const int32_t* blk_g0; int ntest;     // test blocks: (first strip, first group), up to 256 groups each
    int tg, ns;                                                  // ... of tg row groups x up to ns strips
    int nupd;                                                    // update blocks of this post launch
    int32_t* list; unsigned long long* cnt;     // the list of THIS launch's rank and its counters cnt[0..2]
    int ugrid;        // unit-scan blocks per rank
    int urecs;        // unit records in partials (ugrid x ranks); the new-row records follow them
    int nrb;          // new-row blocks = ceil(P / 512)
    int rec_off;      // first unit record of this launch's rank
    int all_defined;  // unit-sharded mode: every unit record is written by every scan
    int sh_rank, sh_world;
    unsigned long long* cnt_all; int cnt_ranks;     // all local counter quadruples (the update role zeroes the next ones)
    int do_update, do_tests, do_rows;
    int32_t* log_x; int32_t* log_y; double* log_bx; double* log_by;
    unsigned long long* dbg; int64_t dbg_it;     // DPR_NJ_PHASES=<iteration>: per-block phase stamps of that iteration (profiles/nj_phases.py)
    // njp_post2_kernel (large shape): what its producer blocks hand to its test blocks
    void* t2_hdr; double* t2_rmax; double* t2_cmax; double* t2_colmin; double* t2_rowmin; double* t2_cmin;
    // row-sharded mode (njr.hip; the kernels' kRS instantiations): D holds this rank's chunks only
    int rs_world, rs_rank;
    unsigned int rs_inv16;         // ceil(65536 / rs_world): k / rs_world == (k * rs_inv16) >> 16 for chunk indices k < 512
    int64_t rs_slice;              // doubles per column slice
    const double* rs_rows;         // [rs_world][2][rs_slice]: columns px / py of every rank's rows as exchanged for this iteration
    int rs_plan;                   // kNjrCollective / kNjrMailbox
    char* const* rs_win;           // [rs_world] the njr regions of all ranks' windows, valid in this process (mailbox plan)
    NjrLayout rs_lay;
    unsigned int* rs_ticket;       // last-block tickets (kNjsTicketBytes)
    unsigned long long rs_seq_base, rs_poll_ticks;
    int64_t rs_fault_it;           // test hook (dpr_ctx_set_debug_fault): this rank's header of that iteration carries a wrong row-sum word (-1: off)
};

// ---- row-sharded mode: where a position's row lives --------------------------------------------------------------
// chunk k = p >> 10 belongs to rank k % world and is that rank's (k / world)-th chunk
template <bool kRS> __device__ __forceinline__ int64_t njp_lrow(const NjpArgs& a, int64_t p)
{
    if (!kRS) return p;
    const unsigned int k = (unsigned int)(p >> 10), kq = (k * a.rs_inv16) >> 16;
    return ((int64_t)kq << 10) | (p & 1023);
}
template <bool kRS> __device__ __forceinline__ bool njp_owns(const NjpArgs& a, int64_t p)
{
    if (!kRS) return true;
    const unsigned int k = (unsigned int)(p >> 10), kq = (k * a.rs_inv16) >> 16;
    return (int)(k - kq * (unsigned int)a.rs_world) == a.rs_rank;
}
// offset of position p in the exchanged column px (add rs_slice for column py)
__device__ __forceinline__ int64_t njp_xoff(const NjpArgs& a, int64_t p)
{
    const unsigned int k = (unsigned int)(p >> 10), kq = (k * a.rs_inv16) >> 16;
    const unsigned int o = k - kq * (unsigned int)a.rs_world;
    return (int64_t)(2u * o) * a.rs_slice + (((int64_t)kq << 10) | (p & 1023));
}

// ---- row-sharded mode, exchange primitives (the conventions of njs.hip: no fences; what travels behind a flag is written with
// system-scope write-through stores and read with system-scope loads, the writer drains its stores before the flag store) ----
__device__ __forceinline__ void njr_st_sys(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned long long njr_ld_sys(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void njr_st_flag(unsigned long long* p, unsigned long long v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// a block record that another block of the same launch reads (agent-scope write-through: no release fence, see njs.hip)
__device__ __forceinline__ void njr_store_record(NjRecord* dst, double q, unsigned long long key, double d, unsigned long long pad)
{
    unsigned long long* w = reinterpret_cast<unsigned long long*>(dst);
    __hip_atomic_store(w + 0, (unsigned long long)__double_as_longlong(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 1, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 2, (unsigned long long)__double_as_longlong(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(w + 3, pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ char* njr_win_recs(char* region, const NjrLayout& lay, int parity, int src_rank)
{
    return region + lay.off_recs + (int64_t)sizeof(NjRecord) * ((int64_t)(parity * kNjsMaxWorld + src_rank) * lay.rec_stride);
}
__device__ __forceinline__ unsigned long long* njr_win_recflag(char* region, const NjrLayout& lay, int parity, int src_rank)
{
    return reinterpret_cast<unsigned long long*>(region + lay.off_recflag) + 8 * (parity * kNjsMaxWorld + src_rank);
}
__device__ __forceinline__ unsigned long long* njr_win_rowflag(char* region, const NjrLayout& lay, int src_rank)
{
    return reinterpret_cast<unsigned long long*>(region + lay.off_rowflag) + 8 * src_rank;
}
// Two-level last-block ticket of a launch (njs.hip's: one 128-byte line per group of blocks, then the top word).  Every
// thread of the block must call it, with the block's own stores to be published issued before; returns true in the block that
// finished last.  The caller resets the words (stream order protects the next launch).
// vmcnt counts per WAVEFRONT: every wavefront drains its own stores and the block meets at a barrier BEFORE thread 0 takes the
// ticket -- otherwise the last block could raise the flag while another wavefront's remote stores (njr_extract_kernel's column
// slices, written by all 256 threads) are still in flight (advisor, round 5).
__device__ __forceinline__ bool njr_last_block(unsigned int* ticket, unsigned int* s_last)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && gridDim.x <= kNjsTicketGroups) {
        // few blocks: one word (a second level would only add a dependent atomic)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int tt = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = (tt == gridDim.x - 1) ? 1u : 0u;
    } else if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int G = gridDim.x, grp = blockIdx.x % kNjsTicketGroups;
        const unsigned int in_grp = (G - grp + kNjsTicketGroups - 1) / kNjsTicketGroups, groups = G < kNjsTicketGroups ? G : kNjsTicketGroups;
        unsigned int last = 0u;
        const unsigned int t = __hip_atomic_fetch_add(ticket + 32 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == in_grp - 1) {
            const unsigned int tt = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (tt == groups - 1) ? 1u : 0u;
        }
        *s_last = last;
    }
    __syncthreads();
    return *s_last != 0u;
}
// SCAN, mailbox plan: the launch's last block sends this rank's header + unit records of iteration `it` to every rank's window
// (own one included) and then the sequence word.  Called by every thread of every block that passed the kernel's first exit.
__device__ __forceinline__ void njr_scan_publish(const NjpArgs& a, int64_t it, unsigned int* s_last)
{
    if (a.rs_plan != kNjrMailbox) return;
    if (!njr_last_block(a.rs_ticket, s_last)) return;
    const int tid = threadIdx.x;
    if (tid <= (int)kNjsTicketGroups) a.rs_ticket[32 * tid] = 0u;
    const int nrec = a.ugrid + 1, par = (int)(it & 1);
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.partials + a.rec_off - 1);
    const int words = 4 * nrec;
    // a thread loads its words ONCE (all loads in flight together), then stores them to every rank
    for (int k0 = tid; k0 < words; k0 += 4 * kThreads) {
        unsigned long long v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (k0 + u * kThreads < words) ? __hip_atomic_load(src + k0 + u * kThreads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        for (int peer = 0; peer < a.rs_world; ++peer) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(njr_win_recs(a.rs_win[peer], a.rs_lay, par, a.rs_rank));
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u * kThreads < words) njr_st_sys(dst + k0 + u * kThreads, v[u]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < a.rs_world) njr_st_flag(njr_win_recflag(a.rs_win[tid], a.rs_lay, par, a.rs_rank), a.rs_seq_base + (unsigned long long)(it + 1));
}

// host launchers of njp.hip's kernels in their row-sharded instantiations (njr.hip builds the arguments)
int njp_rs_launch_list_all(const NjpArgs& a, hipStream_t s);
int njp_rs_launch_scan(const NjpArgs& a, hipStream_t s);
int njp_rs_launch_post(const NjpArgs& a, int64_t N, hipStream_t s);
int njp_rs_launch_finish(const NjpArgs& a, hipStream_t s);
int njp_rs_epoch(NjPruned& q, int64_t P, int64_t N, double* Dbuf, int epoch_index, int rs_rank, int rs_world, const void* hdr_from, hipStream_t s);
int njp_rs_init_vectors(NjPruned& q, const double* U_src, const int32_t* slot_src, int64_t P, int64_t n, int64_t it, hipStream_t s);
void njp_rs_sort_by_row_sum(std::vector<int32_t>& perm, const std::vector<double>& hU);     // ascending, NaN last, stable
int64_t njp_vec_len(int64_t N);        // doubles per per-position vector (256-byte multiple)
int njp_scan_grid_default();           // blocks of the unit scan (DPR_NJP_GRID, default 512)

// phase stamps (debug; 100 MHz wall clock): thread 0 of every block, kernel k (0 scan, 1 post), slot j
#define NJP_STAMP(k, j, drain)                                                                             \
    do {                                                                                                   \
        if (a.dbg != nullptr && it == a.dbg_it && threadIdx.x == 0) {                                      \
            if (drain) __builtin_amdgcn_s_waitcnt(0);                                                      \
            a.dbg[((k) * 2048 + (int)blockIdx.x) * 8 + (j)] = wall_clock64();                              \
        }                                                                                                  \
    } while (0)

// the reference's update arithmetic (src/neighborJoining.cu:171-176), one place for the update role and for the
// test role that needs the same row sums before they are stored
__device__ __forceinline__ double nj_val(double dxi, double dyi, double d) { return (dxi + dyi - d) * 0.5; }
__device__ __forceinline__ double nj_unew(double up, double dxi, double dyi, double val) { return up + (-dxi - dyi + val); }

__device__ __forceinline__ void best_update4(double& bq, uint64_t& bk, uint64_t& bp, double& bd, double q, uint64_t k,
                                             uint64_t pp, double d)
{
    const bool take = (q < bq) | ((q == bq) & (k < bk));
    bq = take ? q : bq; bk = take ? k : bk; bp = take ? pp : bp; bd = take ? d : bd;
}


}  // namespace dpr
