// C ABI, distance matrix and neighbor joining: plan selection (single GPU: pruned / streaming; several ranks: replicated,
// unit-sharded, row-sharded streaming, row-sharded pruned), dpr_dist_matrix, dpr_nj_run, dpr_argmin_once and the NJ getters.
#include "ctx_internal.hpp"

namespace dpr {
// NJ algorithm on a single GPU: 1 = exact pruned scan (njp.hip, default), 0 = full streaming scan
static int g_nj_mode = -1;
static int g_nj_vshards = 1;   // > 1: a single-rank context emulates that many unit-sharded ranks (validation)
// Several ranks, pruned NJ: 0 = auto (unit-sharded scans from kNjShardTips tips on, below that every rank runs the
// single-GPU plan on its own copy: an iteration is then ~20 us of dependent latency and a collective per iteration
// would only add to it; ROW-SHARDED pruned -- njr.hip -- once two copies of the matrix no longer fit one GPU), 1 = always
// unit-sharded, 2 = never, 3 = row-sharded pruned (dpr_set_nj_multi_plan / DPR_NJ_MULTI=auto|shard|solo|rows)
static int g_nj_multi_plan = -1;
constexpr int64_t kNjShardTips = 65536;
static int nj_multi_plan()
{
    if (g_nj_multi_plan < 0) {
        const char* e = std::getenv("DPR_NJ_MULTI");
        g_nj_multi_plan = (e && std::strcmp(e, "shard") == 0) ? 1 : (e && std::strcmp(e, "solo") == 0) ? 2 : (e && std::strcmp(e, "rows") == 0) ? 3 : 0;
    }
    return g_nj_multi_plan;
}

bool want_pruned(const dpr_ctx* c)
{
    if (c->nj_mode >= 0) return c->nj_mode == 1;
    if (g_nj_mode < 0) {
        const char* e = std::getenv("DPR_NJ_MODE");
        g_nj_mode = (e && std::strcmp(e, "stream") == 0) ? 0 : 1;
    }
    return g_nj_mode == 1;
}
static int g_nj_exchange = -1;
int ctx_exchange_plan(const dpr_ctx* c)
{
    if (c->local_comm) return kNjsMailbox;
    if (c->nj_exchange >= 0) return c->nj_exchange;
    if (g_nj_exchange < 0) {
        const char* e = std::getenv("DPR_NJ_EXCHANGE");
        // Default LEGACY (round 4, advisor): the one-exchange plans have only ever run with virtual ranks and process ranks on
        // ONE device, where peer memory is local; until `bench.py --gpus G` has shown `matches_single_gpu` for them on real
        // multi-GPU hardware they are opt-in (DPR_NJ_EXCHANGE=peer|mailbox, dpr_ctx_set_nj_exchange -- bench.py times all three).
        g_nj_exchange = (e && std::strcmp(e, "peer") == 0) ? kNjsPeer : (e && std::strcmp(e, "mailbox") == 0) ? kNjsMailbox : kNjsLegacy;
    }
    return g_nj_exchange;
}
int ctx_multi_plan(const dpr_ctx* c) { return c->nj_multi_plan >= 0 ? c->nj_multi_plan : nj_multi_plan(); }
int ctx_vshards(const dpr_ctx* c) { return c->nj_vshards >= 1 ? c->nj_vshards : g_nj_vshards; }

// Row-sharded exact pruned NJ (njr.hip): asked for (plan 3; the only way for a context of virtual ranks), or -- real ranks,
// plan auto -- when the two epoch buffers of the replicated plans (2 x 8 n^2 bytes) no longer fit this device
bool ctx_njr(const dpr_ctx* c, int64_t n)
{
    if (c->world < 2 || n < 3 || !want_pruned(c)) return false;
    const int plan = ctx_multi_plan(c);
    if (plan == 3) return true;
    if (plan != 0 || c->vworld > 0) return false;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return false; }
    return 2.0 * 8.0 * (double)n * (double)n > 0.85 * (double)total_b;
}
// matrix rows per epoch buffer of a rank under that plan: the same on every rank (the peers compute each other's second half)
int64_t njr_twin_rows(int64_t n, int world)
{
    const int64_t nblk = (n + kRowBlock - 1) / kRowBlock;
    const int64_t tip_rows = ((nblk + world - 1) / world) * kRowBlock + 32, pos_rows = njr_rows_cap(n, world);
    return tip_rows > pos_rows ? tip_rows : pos_rows;
}
static std::vector<NjBuffers*> njr_ranks(dpr_ctx* c)
{
    std::vector<NjBuffers*> v;
    for (auto& b : c->nj) v.push_back(&b);
    return v;
}
// one NJ iteration (active size n, iteration index it) on every rank held by this context
static int nj_iteration(dpr_ctx* c, int64_t n, int64_t it)
{
    if (c->world == 1) {
        NjBuffers& b = c->nj[0];
        if (int rc = nj_launch_scan(b, false, n, it, c->stream)) return rc;
        return nj_launch_post(b, n, it, c->stream);
    }
    if (c->nj_exchange_active != kNjsLegacy) {
        // one exchange, two launches (njs.hip): scan + record, [all-gather of the records | nothing: mailboxes], update
        for (auto& b : c->nj)
            if (int rc = njs_launch_scan(b, n, it, c->njs_pending, c->stream)) return rc;
        if (c->nj_exchange_active == kNjsPeer)
            if (int rc = exchange(c, EX_RECS64)) return rc;
        for (auto& b : c->nj)
            if (int rc = njs_launch_post(b, n, it, c->njs_pending, c->stream)) return rc;
        c->njs_pending = true;
        c->nj_launches += 2;
        return DPR_OK;
    }
    c->nj_launches += 4;
    for (auto& b : c->nj) {
        if (int rc = nj_launch_scan(b, false, n, it, c->stream)) return rc;
        if (int rc = nj_launch_select_local(b, nj_scan_grid(), c->stream)) return rc;
    }
    if (int rc = exchange(c, EX_RECS)) return rc;
    for (auto& b : c->nj)
        if (int rc = nj_launch_commit_extract(b, n, it, c->stream)) return rc;
    if (int rc = exchange(c, EX_SLICES)) return rc;
    for (auto& b : c->nj)
        if (int rc = nj_launch_update_sharded(b, n, c->stream)) return rc;
    return DPR_OK;
}

}  // namespace dpr

using namespace dpr;

__global__ void dpr_warm_kernel(int x);

extern "C" {

// ---- distance matrix ----------------------------------------------------------------------------------
int dpr_dist_matrix(dpr_ctx* c, int source, int dist_type, int k)
{
    if (!c) { set_error("dpr_dist_matrix: null ctx"); return DPR_ERR_ARG; }
    DPR_HIP(hipSetDevice(c->device));
    if (c->world > 1 && c->vworld == 0 && !c->comm && !c->local_comm && !c->shm) { set_error("dpr_dist_matrix: dpr_comm_init was not called"); return DPR_ERR_STATE; }
    int64_t n = 0;
    if (source == DPR_SRC_MSA) {
        if (!c->msa.planes) { set_error("dpr_dist_matrix: call dpr_set_msa first"); return DPR_ERR_STATE; }
        n = c->msa.n;
    } else if (source == DPR_SRC_MATRIX) {
        if (!c->packed_lower) { set_error("dpr_dist_matrix: call dpr_set_matrix_lower first"); return DPR_ERR_STATE; }
        n = c->n_input;
    } else if (source == DPR_SRC_MASH) {
        if (!c->mash.sketches) { set_error("dpr_dist_matrix: call dpr_set_reads and dpr_sketch first"); return DPR_ERR_STATE; }
        if (k != c->mash.k) { set_error("dpr_dist_matrix: k differs from the sketch k"); return DPR_ERR_ARG; }
        n = c->mash.n;
    } else {
        set_error("dpr_dist_matrix: source not available");
        return DPR_ERR_ARG;
    }
    c->have_matrix = 0;
    // Several real ranks + pruned NJ: every rank builds and keeps the WHOLE matrix (7.2 GB at 30 000 tips, 80 GB at
    // 100 000) and the ranks share the per-iteration unit tests and scans (njp.hip, unit-sharded mode).  The
    // streaming algorithm (DPR_NJ_MODE=stream) keeps the row-sharded layout.
    const bool njr = ctx_njr(c, n);
    const bool repl = !njr && c->world > 1 && c->vworld == 0 && want_pruned(c) && n >= 3;
    c->nj_replicated = repl;
    c->nj_row_pruned = false;
    for (size_t r = 0; r < c->nj.size(); ++r)
        if (int rc = nj_alloc(c->nj[r], n, repl ? 0 : (c->vworld > 0 ? (int)r : c->rank), repl ? 1 : c->world, c->stream, njr ? njr_twin_rows(n, c->world) : 0)) return rc;
    const bool row_sharded = c->world > 1 && !repl;
    if (row_sharded) { if (int rc = njs_setup(c, njr)) return rc; }
    else c->nj_exchange_active = kNjsLegacy;
    if (njr && c->nj_exchange_active == kNjsLegacy) {
        set_error("dpr_dist_matrix: the row-sharded pruned NJ needs the peers' buffers mapped on every rank (" + c->nj_exchange_note + "); use DPR_NJ_MODE=stream");
        return DPR_ERR_STATE;
    }
    const bool peer_plan = row_sharded && c->nj_exchange_active != kNjsLegacy;
    DPR_HIP(hipEventRecord(c->ev[0], c->stream));
    for (auto& b : c->nj) {
        if (source == DPR_SRC_MSA) {
            if (int rc = msa_dist_rows(c->msa, b, dist_type, c->stream)) return rc;
        } else if (source == DPR_SRC_MASH) {
            if (b.world > 1) {
                if (int rc = mash_dist_matrix_sharded(c->mash, b.rank, b.world, b.rows_local, b.D, b.ld, c->stream)) return rc;
            } else {
                for (int64_t r0 = 0; r0 < b.rows_local; r0 += 32768) {
                    const int64_t nr = b.rows_local - r0 < 32768 ? b.rows_local - r0 : 32768;
                    if (int rc = mash_dist_rows(c->mash, r0, nr, b.rank, b.world, true, n, b.D + r0 * b.ld, b.ld, c->stream)) return rc;
                }
            }
        } else {
            if (int rc = nj_expand_lower(b, c->packed_lower, c->stream)) return rc;
        }
        // row sums of the own rows: into U (one rank), the slice of the legacy exchange, or the window's slice (peer plans)
        double* sums = !row_sharded ? nullptr : peer_plan ? reinterpret_cast<double*>(b.peer.win + b.peer.lay.off_slice) : b.slice;
        if (int rc = nj_init_sums(b, c->stream, sums)) return rc;
    }
    if (row_sharded && peer_plan) {
        // every rank reads the other ranks' sums straight from their windows, behind one barrier
        if (int rc = njs_barrier(c)) return rc;
        for (auto& b : c->nj)
            if (int rc = njs_launch_unpack_u(b, c->stream)) return rc;
    } else if (row_sharded) {
        if (int rc = exchange(c, EX_U)) return rc;
        for (auto& b : c->nj)
            if (int rc = nj_launch_unpack_u(b, c->stream)) return rc;
    }
    for (auto& b : c->nj)
        if (int rc = nj_prepare(b, c->stream)) return rc;
    if (njr) {
        // exchange plan of the loop: -1 / 0 = default (collective -- all-gathers -- with RCCL and between virtual ranks; mailbox for
        // ranks joined without RCCL), 1 = collective, 2 = mailbox
        int rplan = c->nj_exchange == 2 ? kNjrMailbox : c->nj_exchange == 1 ? kNjrCollective : (c->local_comm ? kNjrMailbox : kNjrCollective);
        if (c->nj_exchange < 0 && !c->local_comm)
            if (const char* e = std::getenv("DPR_NJ_EXCHANGE")) rplan = std::strcmp(e, "mailbox") == 0 ? kNjrMailbox : kNjrCollective;
        if (rplan == kNjrCollective && c->vworld == 0 && !comm_real(c)) { set_error("dpr_dist_matrix: the collective plan of the row-sharded pruned NJ needs a transport between the ranks (RCCL, or dpr_comm_init_shared)"); return DPR_ERR_STATE; }
        for (size_t r = 0; r < c->nj.size(); ++r) {
            NjBuffers& b = c->nj[r];
            b.rs.world = c->world; b.rs.rank = c->vworld > 0 ? (int)r : c->rank; b.rs.plan = rplan;
            b.rs.win_off = b.peer.lay.off_njr;
            b.rs.gather = njr_gather_cb; b.rs.cb_ctx = c;
            // (ranks on the shared region's windows: with the mailbox plan the barrier runs through the njr windows, no callback;
            //  with the collective plan through the region)
            b.rs.barrier = (c->vworld == 0 && (c->comm || (c->shm && rplan == kNjrCollective))) ? njr_barrier_cb : nullptr;
            b.rs.launches = 0; b.rs.collectives = 0;
        }
        std::vector<NjBuffers*> ranks = njr_ranks(c);
        if (int rc = njr_build(ranks, c->stream)) return rc;
        c->nj_row_pruned = true;
        c->nj_exchange_note = std::string("row-sharded pruned NJ (njr.hip), exchange plan ") + (rplan == kNjrMailbox ? "mailbox" : "collective");
    }
    if ((c->world == 1 || repl) && want_pruned(c) && n >= 3) {
        NjPruned& q = c->nj[0].pr;
        const int plan = ctx_multi_plan(c);
        const bool shard = repl && (plan == 1 || (plan == 0 && n >= kNjShardTips));
        c->nj_unit_sharded = shard;
        if (shard) { q.sh_world = c->world; q.sh_rank = c->rank; q.sh_virtual = false; q.gather = njp_gather_cb; q.gather_ctx = c; }
        else if (ctx_vshards(c) > 1) { q.sh_world = ctx_vshards(c); q.sh_rank = 0; q.sh_virtual = true; }
        if (c->nj_adaptive >= 0) q.adaptive = c->nj_adaptive;
        if (int rc = njp_build(c->nj[0], c->stream)) return rc;
        if (c->nj_adaptive >= 0) q.adaptive = c->nj_adaptive;      // (the explicit setting wins over the environment)
    }
    DPR_HIP(hipEventRecord(c->ev[1], c->stream));
    DPR_HIP(hipStreamSynchronize(c->stream));
    float ms = 0;
    DPR_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->dist_ms = ms;
    c->have_matrix = 1;   // (the packed triangle of a MATRIX source stays until dpr_set_matrix_lower / dpr_destroy)
    return DPR_OK;
}

// Allocate the N x N matrix buffers of a following dpr_dist_matrix(n tips) now (single-rank contexts; a no-op
// otherwise): dpr_dist_matrix finds them in place.  The CLI calls it from its device thread as soon as the number of
// input sequences is known, while the host threads are still packing them.
int dpr_reserve_nj(dpr_ctx* c, int64_t n)
{
    if (!c || n < 2 || n >= (1 << 24)) { set_error("dpr_reserve_nj: bad argument"); return DPR_ERR_ARG; }
    if (c->world != 1 || c->vworld > 0) return DPR_OK;
    DPR_HIP(hipSetDevice(c->device));
    c->have_matrix = 0;
    if (int rc = nj_alloc(c->nj[0], n, 0, 1, c->stream)) return rc;
    if (want_pruned(c) && n >= 3) {
        NjPruned& q = c->nj[0].pr;
        if (ctx_vshards(c) > 1) { q.sh_world = ctx_vshards(c); q.sh_rank = 0; q.sh_virtual = true; }
        if (int rc = njp_reserve(q, n, c->stream)) return rc;
    }
    DPR_HIP(hipStreamSynchronize(c->stream));
    return DPR_OK;
}

// The first hipGraph of a process costs ~30 ms to instantiate (the next ones 0.2 ms); the pruned NJ replays graphs, so that
// cost would sit in front of its first 32 iterations with the GPU idle.  The CLI calls this from a helper thread while it
// reads its input (a private stream: nothing of the context's stream is touched).  Safe to call any number of times.
int dpr_warm_graphs(dpr_ctx* c)
{
    if (!c) { set_error("dpr_warm_graphs: null ctx"); return DPR_ERR_ARG; }
    DPR_HIP(hipSetDevice(c->device));
    hipStream_t st = nullptr;
    DPR_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        hipLaunchKernelGGL(dpr_warm_kernel, dim3(1), dim3(64), 0, st, 0);
        if (hipStreamEndCapture(st, &g) == hipSuccess && g) {
            if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess && ge) {
                (void)hipGraphLaunch(ge, st);
                (void)hipStreamSynchronize(st);
                (void)hipGraphExecDestroy(ge);
            }
            (void)hipGraphDestroy(g);
        }
    }
    // ... and of its staged copies: the first device-to-host copy of more than a few KB into pageable memory costs 7.7 ms
    // (staging buffers); without this it is the first epoch rebuild of the NJ run that pays (240 KB of row sums)
    {
        void* d = nullptr;
        constexpr size_t kWarmBytes = 512 << 10;
        if (hipMalloc(&d, kWarmBytes) == hipSuccess) {
            std::vector<char> h(kWarmBytes);
            (void)hipMemsetAsync(d, 0, kWarmBytes, st);
            (void)hipStreamSynchronize(st);
            // (on the private stream: a synchronous hipMemcpy runs on the NULL stream, which serialises with every blocking
            //  stream of the device -- this function may run beside other dpr_* calls of the context; pageable staging is
            //  exercised all the same)
            (void)hipMemcpyAsync(h.data(), d, kWarmBytes, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpyAsync(d, h.data(), kWarmBytes, hipMemcpyHostToDevice, st);
            (void)hipStreamSynchronize(st);
            (void)hipFree(d);
        }
    }
    (void)hipGetLastError();
    (void)hipStreamDestroy(st);
    return DPR_OK;
}

// ---- NJ -------------------------------------------------------------------------------------------------
} // extern "C" (helper)
namespace dpr {
int fetch_state(dpr_ctx* c, NjState* st)
{
    DPR_HIP(hipMemcpyAsync(st, c->nj[0].st, sizeof(NjState), hipMemcpyDeviceToHost, c->stream));
    DPR_HIP(hipStreamSynchronize(c->stream));
    return DPR_OK;
}
}  // namespace dpr
extern "C" {

int64_t dpr_nj_run(dpr_ctx* c, int64_t max_iters, int32_t* merge_x, int32_t* merge_y, double* bl_x,
                   double* bl_y, double* last_d)
{
    if (!c || !c->have_matrix) { set_error("dpr_nj_run: call dpr_dist_matrix first"); return DPR_ERR_STATE; }
    DPR_HIP(hipSetDevice(c->device));
    NjState st;
    if (int rc = fetch_state(c, &st)) return rc;
    int64_t todo = st.n - 2;
    if (todo < 0) todo = 0;
    if (max_iters >= 0 && max_iters < todo) todo = max_iters;
    const int64_t it0 = st.it;
    c->nj[0].kt = &c->nj_kt;
    if (c->nj_kt.stride > 0 && it0 == 0) { c->nj_kt.samples = 0; for (double& v : c->nj_kt.us_sum) v = 0; }
    c->nj_launches = 0; c->nj_collectives = 0;
    DPR_HIP(hipEventRecord(c->ev[2], c->stream));
    if (c->nj_row_pruned) {
        std::vector<NjBuffers*> ranks = njr_ranks(c);
        c->nj[0].rs.launches = 0; c->nj[0].rs.collectives = 0;
        if (int rc = njr_run(ranks, it0, todo, c->stream)) return rc;
        c->nj_launches = c->nj[0].rs.launches; c->nj_collectives = c->nj[0].rs.collectives;
    } else if (c->nj[0].pr.active) {
        if (int rc = njp_run(c->nj[0], it0, todo, c->stream)) return rc;
    } else {
        for (int64_t k = 0; k < todo; ++k)
            if (int rc = nj_iteration(c, st.n - k, it0 + k)) return rc;
    }
    DPR_HIP(hipEventRecord(c->ev[3], c->stream));       // (the loop itself: the barrier + flush below are once per run)
    const bool peer_plan = c->world > 1 && !c->nj_replicated && !c->nj_row_pruned && c->nj_exchange_active != kNjsLegacy;
    if (peer_plan) {
        // every rank must be through its pulls of the last iteration before an owner flushes the last row buffers
        if (int rc = njs_barrier(c)) return rc;
        for (auto& b : c->nj)
            if (int rc = njs_launch_finish(b, st.n - todo, it0 + todo, c->njs_pending, c->stream)) return rc;
        c->njs_pending = false;
        if (int rc = njs_barrier(c)) return rc;      // the flushed rows may be read by other ranks (final distance, hooks)
    } else {
        for (auto& b : c->nj)
            if (!b.pr.active)
                if (int rc = nj_launch_finish(b, st.n - todo, it0 + todo, c->stream)) return rc;
    }
    if (int rc = fetch_state(c, &st)) return rc;
    float ms = 0;
    DPR_HIP(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
    c->nj_ms = ms;
    if (!c->nj_kt.ev.empty()) {       // per-kernel timing samples of this run (stream idle)
        NjKernelTiming& kt = c->nj_kt;
        const size_t grp = (size_t)kt.nk + 1;
        for (size_t g0 = 0; kt.nk > 0 && g0 + grp <= kt.ev.size(); g0 += grp) {
            for (int k = 0; k < kt.nk; ++k) {
                float us = 0;
                if (hipEventElapsedTime(&us, kt.ev[g0 + (size_t)k], kt.ev[g0 + (size_t)k + 1]) == hipSuccess) kt.us_sum[k] += (double)us * 1e3;
            }
            ++kt.samples;
        }
        for (hipEvent_t e : kt.ev) (void)hipEventDestroy(e);
        kt.ev.clear();
    }
    const int64_t done = st.it - it0;
    if (done > 0) {
        if (merge_x) DPR_HIP(hipMemcpy(merge_x, c->nj[0].log_x + it0, sizeof(int32_t) * (size_t)done, hipMemcpyDeviceToHost));
        if (merge_y) DPR_HIP(hipMemcpy(merge_y, c->nj[0].log_y + it0, sizeof(int32_t) * (size_t)done, hipMemcpyDeviceToHost));
        if (bl_x) DPR_HIP(hipMemcpy(bl_x, c->nj[0].log_bx + it0, sizeof(double) * (size_t)done, hipMemcpyDeviceToHost));
        if (bl_y) DPR_HIP(hipMemcpy(bl_y, c->nj[0].log_by + it0, sizeof(double) * (size_t)done, hipMemcpyDeviceToHost));
    }
    if (st.status == 3) {
        set_error("dpr_nj_run: the exchange between the ranks failed (a rank's record did not arrive within the poll limit, or the all-gather delivered a stale one)");
        return DPR_ERR_COMM;
    }
    if (st.status == 4) {
        // (njs_post_kernel left the two differing words in st.q / st.d and the ranks in st.x / st.y)
        char msg[320];
        std::snprintf(msg, sizeof msg, "dpr_nj_run: the ranks' replicated row sums differ after %lld iterations (rank %d: %a, rank %d: %a): a row pulled from its owner "
                      "was stale or torn -- the merge log up to here is not trustworthy; use the legacy exchange (dpr_ctx_set_nj_exchange(ctx, 0))",
                      (long long)st.it, (int)st.x, st.q, (int)st.y, st.d);
        set_error(msg);
        return DPR_ERR_COMM;
    }
    if (st.status == 5) {
        set_error("dpr_nj_run: internal: the test blocks of the post kernel did not see the producer blocks' tag within 2 ms (njp_post2_kernel; DPR_NJP_POST2=0 selects the fused kernel)");
        return DPR_ERR_HIP;
    }
    if (st.status != 0) {
        set_error("dpr_nj_run: no Q candidate below the reference's init value 10000 (undefined in the reference)");
        return DPR_ERR_NOCAND;
    }
    if (last_d && st.n == 2) {
        // D[0][1] of the final pair (src/neighborJoining.cu:245-249); row 1 lives on rank 0
        NjBuffers& b0 = c->nj[0];
        if (c->nj_row_pruned) {
            // the row of slot 1 lives on its position's owner: read through the mapping of that rank's epoch buffer (every
            // rank's finish kernel has run: njr_run ends with a barrier over the ranks behind the finish launches)
            int32_t pos01[2];
            DPR_HIP(hipMemcpy(pos01, b0.pr.pos_of_slot, sizeof(pos01), hipMemcpyDeviceToHost));
            const int half = (b0.pr.epoch_index + 1) & 1, o = njr_owner(pos01[1], c->world);
            DPR_HIP(hipMemcpy(last_d, b0.rs.peer_half[half][(size_t)o] + njr_local_row(pos01[1], c->world) * b0.pr.ld + pos01[0], sizeof(double), hipMemcpyDeviceToHost));
        } else if (b0.pr.in_positions()) {
            int32_t pos01[2];
            DPR_HIP(hipMemcpy(pos01, b0.pr.pos_of_slot, sizeof(pos01), hipMemcpyDeviceToHost));
            DPR_HIP(hipMemcpy(last_d, b0.pr.D + (int64_t)pos01[1] * b0.pr.ld + pos01[0], sizeof(double), hipMemcpyDeviceToHost));
        } else if (c->world == 1 || c->vworld > 0) {
            DPR_HIP(hipMemcpy(last_d, b0.D + 1 * b0.ld + 0, sizeof(double), hipMemcpyDeviceToHost));
        } else if (peer_plan && !b0.peer.h_D.empty()) {
            // rank 0's row 1 through the mapping of its matrix (its flush is behind the barrier above)
            DPR_HIP(hipMemcpy(last_d, b0.peer.h_D[0] + 1 * b0.ld + 0, sizeof(double), hipMemcpyDeviceToHost));
        } else {
            NjRecord rec{ 0.0, 0ull, 0.0, 0ull };
            if (c->rank == 0) DPR_HIP(hipMemcpy(&rec.d, b0.D + 1 * b0.ld + 0, sizeof(double), hipMemcpyDeviceToHost));
            DPR_HIP(hipMemcpy(b0.recs + c->rank, &rec, sizeof(NjRecord), hipMemcpyHostToDevice));
            if (int rc = exchange(c, EX_RECS)) return rc;
            DPR_HIP(hipStreamSynchronize(c->stream));
            DPR_HIP(hipMemcpy(&rec, b0.recs + 0, sizeof(NjRecord), hipMemcpyDeviceToHost));
            *last_d = rec.d;
        }
    }
    return done;
}

int dpr_argmin_once(dpr_ctx* c, int reps, int32_t* out_i, int32_t* out_j, double* out_q, float* out_ms)
{
    if (!c || !c->have_matrix) { set_error("dpr_argmin_once: call dpr_dist_matrix first"); return DPR_ERR_STATE; }
    DPR_HIP(hipSetDevice(c->device));
    if (reps < 1) reps = 1;
    NjState st0;
    if (int rc = fetch_state(c, &st0)) return rc;
    // pruned mode: the streaming kernel runs over the position-space matrix (all P positions, dead
    // ones carry NaN row sums); it = 0 because the bounds kernel already finished U[x]
    auto probe = [&](NjBuffers& b) -> int {
        return b.pr.in_positions() ? nj_launch_scan(b, true, b.pr.P, 0, c->stream) : nj_launch_scan(b, true, st0.n, st0.it, c->stream);
    };
    for (auto& b : c->nj)
        if (int rc = probe(b)) return rc;  // warm
    DPR_HIP(hipEventRecord(c->ev[2], c->stream));
    for (int r = 0; r < reps; ++r)
        for (auto& b : c->nj)
            if (int rc = probe(b)) return rc;
    DPR_HIP(hipEventRecord(c->ev[3], c->stream));
    for (auto& b : c->nj)
        if (int rc = nj_launch_select_local(b, nj_scan_grid(), c->stream)) return rc;
    if (int rc = exchange(c, EX_RECS)) return rc;
    const int ew = c->nj_replicated ? 1 : c->world;      // ranks whose records differ
    std::vector<NjRecord> recs((size_t)ew);
    DPR_HIP(hipMemcpyAsync(recs.data(), c->nj[0].recs, sizeof(NjRecord) * (size_t)ew, hipMemcpyDeviceToHost, c->stream));
    DPR_HIP(hipStreamSynchronize(c->stream));
    float ms = 0;
    DPR_HIP(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
    if (out_ms) *out_ms = ms / (float)reps;
    const int w = dpr_record_reduce(recs.data(), ew);
    if (w < 0 || !(recs[(size_t)w].q < 10000.0)) { set_error("dpr_argmin_once: no Q candidate below 10000"); return DPR_ERR_NOCAND; }
    const NjRecord& rec = recs[(size_t)w];
    if (out_i) *out_i = (int32_t)(rec.key & 0xFFFFFFull);
    if (out_j) *out_j = (int32_t)((rec.key >> 24) & 0xFFFFFFull);
    if (out_q) *out_q = rec.q;
    return DPR_OK;
}

int dpr_njp_unit_owner(int64_t strip, int64_t group, int64_t P, int world) { return njp_unit_owner(strip, group, P, world); }

// validation knob: the next dpr_dist_matrix on a single-rank context sets up `w` emulated unit-sharded ranks
int dpr_set_nj_virtual_shards(int w)
{
    if (w < 1 || w > 64) { set_error("dpr_set_nj_virtual_shards: 1 <= w <= 64"); return DPR_ERR_ARG; }
    g_nj_vshards = w;
    return DPR_OK;
}

int dpr_set_nj_multi_plan(int plan)
{
    if (plan < 0 || plan > 3) { set_error("dpr_set_nj_multi_plan: 0 auto, 1 unit-sharded, 2 single-GPU plan on every rank, 3 row-sharded pruned"); return DPR_ERR_ARG; }
    g_nj_multi_plan = plan;
    return DPR_OK;
}
int dpr_nj_is_unit_sharded(dpr_ctx* c) { return c && c->nj_unit_sharded ? 1 : 0; }
// the multi-rank NJ plan the last dpr_dist_matrix set up, in words (the CLI prints it; tests assert on it)
int dpr_get_nj_multi_info(dpr_ctx* c, char* buf, int cap)
{
    if (!c || !buf || cap <= 0) { set_error("dpr_get_nj_multi_info: bad argument"); return DPR_ERR_ARG; }
    static const char* const ex[] = { "legacy (two all-gathers per iteration)", "peer (one all-gather, rows pulled)", "mailbox (no collective)" };
    std::string s;
    if (c->world <= 1) s = "single rank";
    else if (c->nj_row_pruned) s = c->nj_exchange_note;
    else if (c->nj_replicated) s = c->nj_unit_sharded ? "pruned, matrix replicated, unit tests and scans sharded (one all-gather of block records per iteration)"
                                                       : "pruned, every rank runs the single-GPU plan on its own copy of the matrix (replicas)";
    else s = std::string("streaming, rows sharded block-cyclically, exchange ") + ex[c->nj_exchange_active >= 0 && c->nj_exchange_active <= 2 ? c->nj_exchange_active : 0];
    std::snprintf(buf, (size_t)cap, "%s", s.c_str());
    return DPR_OK;
}

// the same three knobs for ONE context (two contexts in one process may run different plans); value -1 = follow
// the process-wide default again.  Take effect at the context's next dpr_dist_matrix.
int dpr_ctx_set_nj_mode(dpr_ctx* c, int mode)
{
    if (!c || mode < -1 || mode > 1) { set_error("dpr_ctx_set_nj_mode: mode must be -1, 0 or 1"); return DPR_ERR_ARG; }
    c->nj_mode = mode;
    return DPR_OK;
}
int dpr_ctx_set_nj_multi_plan(dpr_ctx* c, int plan)
{
    if (!c || plan < -1 || plan > 3) { set_error("dpr_ctx_set_nj_multi_plan: -1 default, 0 auto, 1 unit-sharded, 2 single-GPU plan on every rank, 3 row-sharded pruned"); return DPR_ERR_ARG; }
    c->nj_multi_plan = plan;
    return DPR_OK;
}
// Per-kernel timing of the pruned NJ loop: stride > 0 makes the following dpr_nj_run calls enqueue their iterations
// eagerly (no hipGraph replay) with HIP events on the library's stream around the launches of every stride-th iteration.
int dpr_ctx_set_nj_kernel_timing(dpr_ctx* c, int stride)
{
    if (!c || stride < 0) { set_error("dpr_ctx_set_nj_kernel_timing: stride >= 0"); return DPR_ERR_ARG; }
    c->nj_kt.stride = stride;
    return DPR_OK;
}
int dpr_get_nj_kernel_timing(dpr_ctx* c, int* kernels, double* us_avg, int64_t* samples)
{
    if (!c) { set_error("dpr_get_nj_kernel_timing: null ctx"); return DPR_ERR_ARG; }
    if (kernels) *kernels = c->nj_kt.nk;
    if (samples) *samples = c->nj_kt.samples;
    if (us_avg) for (int k = 0; k < kNjKernelsMax; ++k) us_avg[k] = c->nj_kt.samples > 0 ? c->nj_kt.us_sum[k] / (double)c->nj_kt.samples : 0.0;
    return DPR_OK;
}
const char* dpr_nj_kernel_name(int idx) { return njp_kernel_name(idx); }
int dpr_get_nj_phase_stamps(uint64_t* out) { return njp_phase_stamps((unsigned long long*)out); }
int dpr_get_njp_list(dpr_ctx* c, int32_t* out, int64_t cap, int64_t* count, int64_t* positions, double* ur, int64_t ur_cap)
{
    if (!c || !out || !count || !positions) { set_error("dpr_get_njp_list: null argument"); return DPR_ERR_ARG; }
    DPR_HIP(hipSetDevice(c->device));
    return njp_debug_list(c->nj[0], out, cap, count, positions, ur, ur_cap);
}

int dpr_ctx_set_nj_virtual_shards(dpr_ctx* c, int w)
{
    if (!c || w < -1 || w == 0 || w > 64) { set_error("dpr_ctx_set_nj_virtual_shards: -1 or 1 <= w <= 64"); return DPR_ERR_ARG; }
    c->nj_vshards = w;
    return DPR_OK;
}

// 0 = full streaming scan every iteration, 1 = exact pruned scan (default)
int dpr_set_nj_mode(int mode)
{
    if (mode != 0 && mode != 1) { set_error("dpr_set_nj_mode: mode must be 0 or 1"); return DPR_ERR_ARG; }
    g_nj_mode = mode;
    return DPR_OK;
}

// Adaptive plan of the single-rank NJ (default on): the exact pruned scan while its bounds prune; hand-over to the streaming
// loop once more than 70 % of an epoch's units are listed per iteration, pruned probes with back-off (see dpr_internal.hpp).  The
// merge log does not depend on it.  on = 0: pruned scans only; -1: DPR_NJ_ADAPTIVE / default.  Takes effect at the next
// dpr_dist_matrix.
int dpr_ctx_set_nj_adaptive(dpr_ctx* c, int on)
{
    if (!c || on < -1 || on > 1) { set_error("dpr_ctx_set_nj_adaptive: -1, 0 or 1"); return DPR_ERR_ARG; }
    c->nj_adaptive = on;
    return DPR_OK;
}
// iterations that ran as streaming scans and epochs that switched, since the matrix was built
int dpr_get_nj_adaptive_stats(dpr_ctx* c, int64_t* stream_iterations, int64_t* stream_epochs)
{
    if (!c || !c->have_matrix || !c->nj[0].pr.active) { set_error("dpr_get_nj_adaptive_stats: pruned path not active"); return DPR_ERR_STATE; }
    if (stream_iterations) *stream_iterations = c->nj[0].pr.stream_iterations;
    if (stream_epochs) *stream_epochs = c->nj[0].pr.stream_epochs;
    return DPR_OK;
}

// units scanned by the pruned path since the matrix was built, and units per full scan
int dpr_get_prune_stats(dpr_ctx* c, uint64_t* units_scanned, uint64_t* units_per_full_scan)
{
    if (!c || !c->have_matrix || !c->nj[0].pr.active) { set_error("dpr_get_prune_stats: pruned path not active"); return DPR_ERR_STATE; }
    NjState st;
    if (int rc = fetch_state(c, &st)) return rc;
    if (units_scanned) *units_scanned = st.units_scanned;
    if (units_per_full_scan) *units_per_full_scan = (uint64_t)c->nj[0].pr.utot0;
    return DPR_OK;
}


int dpr_get_nj_progress(dpr_ctx* c, int64_t* iterations_done, int64_t* active)
{
    if (!c || !c->have_matrix) { set_error("dpr_get_nj_progress: call dpr_dist_matrix first"); return DPR_ERR_STATE; }
    DPR_HIP(hipSetDevice(c->device));
    NjState st;
    if (int rc = fetch_state(c, &st)) return rc;
    if (iterations_done) *iterations_done = st.it;
    if (active) *active = st.n;
    return DPR_OK;
}

int dpr_get_njp_shape(dpr_ctx* c, int64_t* positions, int* row_groups, int* strips, int* post2, int* scan_grid)
{
    if (!c || !c->have_matrix || !c->nj[0].pr.active) { set_error("dpr_get_njp_shape: no pruned NJ state"); return DPR_ERR_STATE; }
    return njp_shape(c->nj[0].pr, positions, row_groups, strips, post2, scan_grid);
}

}  // extern "C"
