// C ABI, communication part: RCCL (resolved at run time), the exchanges of the sharded NJ paths, the peer windows and their
// hipIpc mappings, barriers; dpr_comm_* / dpr_peer_* entry points.
#include "ctx_internal.hpp"

#include <dlfcn.h>

namespace dpr {
Rccl g_rccl;
int rccl_load()
{
    if (g_rccl.lib) return DPR_OK;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* nm : names) {
        g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) { set_error("cannot load librccl.so"); return DPR_ERR_COMM; }
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(g_rccl.lib, "ncclCommCount");
    g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(g_rccl.lib, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
        set_error("librccl.so lacks a required symbol");
        return DPR_ERR_COMM;
    }
    return DPR_OK;
}
// in-place all-gather of the block records of the unit-sharded pruned NJ
int njp_gather_cb(void* ctx, void* buf, size_t bytes_per_rank, hipStream_t s)
{
    dpr_ctx* c = static_cast<dpr_ctx*>(ctx);
    return comm_all_gather(c, static_cast<char*>(buf) + (size_t)c->rank * bytes_per_rank, buf, bytes_per_rank, s);
}
// collective plan: kind 0 = every rank's header + unit records (in place in partials), kind 1 = its column slices (rows_plain)
int njr_gather_cb(void* ctx, int kind, hipStream_t s)
{
    dpr_ctx* c = static_cast<dpr_ctx*>(ctx);
    NjBuffers& b0 = c->nj[0];
    const int world = c->world;
    const int ugrid = b0.pr.scan_grid / world > 0 ? b0.pr.scan_grid / world : 1;
    const size_t seg = kind == 0 ? sizeof(NjRecord) * (size_t)(ugrid + 1) : sizeof(double) * 2 * (size_t)b0.rs.lay.slice;
    if (c->vworld > 0) {
        for (int r = 0; r < c->vworld; ++r)
            for (int t = 0; t < c->vworld; ++t) {
                if (t == r) continue;
                char* src = kind == 0 ? reinterpret_cast<char*>(c->nj[(size_t)r].partials) : reinterpret_cast<char*>(c->nj[(size_t)r].rs.rows_plain);
                char* dst = kind == 0 ? reinterpret_cast<char*>(c->nj[(size_t)t].partials) : reinterpret_cast<char*>(c->nj[(size_t)t].rs.rows_plain);
                DPR_HIP(hipMemcpyAsync(dst + (size_t)r * seg, src + (size_t)r * seg, seg, hipMemcpyDeviceToDevice, s));
            }
        return DPR_OK;
    }
    if (!comm_real(c)) { set_error("njr: the collective plan needs a transport between the ranks (RCCL, or the windows of dpr_comm_init_shared)"); return DPR_ERR_COMM; }
    char* buf = kind == 0 ? reinterpret_cast<char*>(b0.partials) : reinterpret_cast<char*>(b0.rs.rows_plain);
    return comm_all_gather(c, buf + (size_t)c->rank * seg, buf, seg, s);
}

// ---- exchange step of the sharded path: RCCL all-gather, or device copies between virtual ranks --

int exchange(dpr_ctx* c, ExKind kind)
{
    if (c->world == 1 || c->nj_replicated) return DPR_OK;
    if (c->vworld > 0) {
        for (int r = 0; r < c->vworld; ++r) {
            NjBuffers& src = c->nj[(size_t)r];
            for (int t = 0; t < c->vworld; ++t) {
                NjBuffers& dst = c->nj[(size_t)t];
                if (kind == EX_RECS) {
                    if (t == r) continue;
                    DPR_HIP(hipMemcpyAsync(dst.recs + r, src.recs + r, sizeof(NjRecord), hipMemcpyDeviceToDevice, c->stream));
                } else if (kind == EX_RECS64) {
                    if (t == r) continue;
                    DPR_HIP(hipMemcpyAsync(dst.recs64 + r, src.recs64 + r, sizeof(NjsRec), hipMemcpyDeviceToDevice, c->stream));
                } else {
                    const size_t cnt = (size_t)(kind == EX_SLICES ? 3 : 1) * (size_t)src.slice_len;
                    DPR_HIP(hipMemcpyAsync(dst.gath + (size_t)r * cnt, src.slice, cnt * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                }
            }
        }
        return DPR_OK;
    }
    NjBuffers& b = c->nj[0];
    if (!comm_real(c)) { set_error("exchange: no transport between the ranks of this context (dpr_comm_init / dpr_comm_init_shared)"); return DPR_ERR_COMM; }
    ++c->nj_collectives;
    if (kind == EX_RECS) return comm_all_gather(c, b.recs + c->rank, b.recs, sizeof(NjRecord), c->stream);
    if (kind == EX_RECS64) return comm_all_gather(c, b.recs64 + c->rank, b.recs64, sizeof(NjsRec), c->stream);
    return comm_all_gather(c, b.slice, b.gath, sizeof(double) * (size_t)(kind == EX_SLICES ? 3 : 1) * (size_t)b.slice_len, c->stream);
}

// ---- peer windows of the one-exchange sharded loop (njs.hip) -----------------------------------------------------
struct PeerBlob {                  // what a rank tells the others about its buffers (192 bytes)
    uint64_t ok;                   // 1: both handles valid
    uint64_t n_tips;
    hipIpcMemHandle_t d, w;        // matrix rows, window
    uint64_t pad[6];
};
static_assert(sizeof(PeerBlob) == 192, "PeerBlob layout");

// The HIP runtime bundled with PyTorch 2.10+rocm7.0 (7.0.51831: a process that imports torch first runs this library on it)
// does not return from hipIpcOpenMemHandle for an allocation whose size has bit 31 set (2.3, 3.6, 3.9 GB hang; 1.9 GB and
// 5.8 GB map), while the system runtime (/opt/rocm, 7.2) maps 14 GB (profiles/ipc_torch_probe.py, profiles/r3/
// ipc_runtime_probe.txt).  A hang cannot be caught, so matrices of 2 GiB and more per rank are not offered to the peers on a
// runtime older than 7.2 at all: the ranks then agree on the legacy loop (dpr_dist_matrix) or dpr_peer_export fails.
// DPR_IPC_ANY_SIZE=1 lifts the guard.
static bool ipc_size_allowed(size_t bytes)
{
    static int large_ok = -1;
    if (large_ok < 0) {
        int v = 0;
        if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); v = 0; }
        large_ok = (v >= 70200000 || std::getenv("DPR_IPC_ANY_SIZE")) ? 1 : 0;
    }
    return large_ok == 1 || bytes < ((size_t)1 << 31);
}

static int peer_blob_of(dpr_ctx* c, PeerBlob* out)
{
    NjBuffers& b = c->nj[0];
    std::memset(out, 0, sizeof(PeerBlob));
    out->n_tips = (uint64_t)b.N;
    if (!b.D || !b.peer.win) return DPR_OK;
    {
        hipDeviceptr_t base = nullptr;
        size_t bytes = 0;
        if (hipMemGetAddressRange(&base, &bytes, b.D) != hipSuccess) { (void)hipGetLastError(); bytes = ~(size_t)0; }
        out->pad[0] = (uint64_t)bytes;
        if (!ipc_size_allowed(bytes)) { out->pad[1] = 2; return DPR_OK; }      // 2: refused by the runtime guard
    }
    if (hipIpcGetMemHandle(&out->d, b.D) != hipSuccess || hipIpcGetMemHandle(&out->w, b.peer.win) != hipSuccess) { (void)hipGetLastError(); return DPR_OK; }
    out->ok = 1;
    return DPR_OK;
}

// map the other ranks' buffers; all[r] for r = 0 .. world-1.  *ok = 0 when any blob is unusable or a mapping fails.
static int peer_attach_blobs(dpr_ctx* c, const PeerBlob* all, int* ok)
{
    NjBuffers& b = c->nj[0];
    *ok = 1;
    for (int r = 0; r < c->world; ++r)
        if (!all[r].ok || all[r].n_tips != (uint64_t)b.N || (r != c->rank && !ipc_size_allowed((size_t)all[r].pad[0]))) *ok = 0;
    if (!*ok) return DPR_OK;
    std::vector<char*> wins((size_t)c->world, nullptr);
    std::vector<double*> Ds((size_t)c->world, nullptr);
    for (int r = 0; r < c->world && *ok; ++r) {
        if (r == c->rank) { wins[(size_t)r] = b.peer.win; Ds[(size_t)r] = b.D; continue; }
        void *pd = nullptr, *pw = nullptr;
        if (hipIpcOpenMemHandle(&pd, all[r].d, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); *ok = 0; break; }
        b.peer.opened.push_back(pd);
        if (hipIpcOpenMemHandle(&pw, all[r].w, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); *ok = 0; break; }
        b.peer.opened.push_back(pw);
        Ds[(size_t)r] = static_cast<double*>(pd);
        wins[(size_t)r] = static_cast<char*>(pw);
    }
    if (!*ok) {
        for (void* m : b.peer.opened) (void)hipIpcCloseMemHandle(m);
        b.peer.opened.clear();
        return DPR_OK;
    }
    return njs_set_peers(b, wins.data(), Ds.data(), c->stream);
}

// all-gather of `bytes` per rank through the staging buffer b.gath (RCCL); host arrays in / out
int rccl_gather_bytes(dpr_ctx* c, const void* mine, void* all, size_t bytes)
{
    NjBuffers& b = c->nj[0];
    char* stage = reinterpret_cast<char*>(b.gath);
    if (!stage || bytes * (size_t)c->world > sizeof(double) * (size_t)(3 * b.slice_len * c->world)) { set_error("rccl_gather_bytes: staging buffer too small"); return DPR_ERR_STATE; }
    DPR_HIP(hipMemcpyAsync(stage + (size_t)c->rank * bytes, mine, bytes, hipMemcpyHostToDevice, c->stream));
    if (g_rccl.AllGather(stage + (size_t)c->rank * bytes, stage, bytes, kNcclUint8, c->comm, c->stream) != 0) { set_error("ncclAllGather(peer handles) failed"); return DPR_ERR_COMM; }
    DPR_HIP(hipMemcpyAsync(all, stage, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    DPR_HIP(hipStreamSynchronize(c->stream));
    return DPR_OK;
}

// Set up the exchange plan of the row-sharded loop for the buffers nj_alloc just provided.  A plan that cannot be set up
// on EVERY rank (no fine-grained window, IPC handles refused, a mapping fails) falls back to the legacy loop on all
// ranks together -- the decision is taken on gathered flags, so the ranks cannot disagree -- and says why in
// nj_exchange_note.  Ranks joined without RCCL (dpr_comm_init_local) have nothing to fall back to: error.
// force_windows (row-sharded pruned NJ): windows and peer mappings are needed whatever the streaming loop's plan is -- the
// epoch builds pull rows from the peers' buffers, the mailbox plan exchanges through the windows
int njs_setup(dpr_ctx* c, bool force_windows)
{
    int plan = ctx_exchange_plan(c);
    if (force_windows && plan == kNjsLegacy) plan = (c->comm && c->vworld == 0) ? kNjsPeer : kNjsMailbox;
    c->nj_exchange_active = kNjsLegacy;
    c->nj_exchange_note.clear();
    c->njs_pending = false;
    if (plan == kNjsLegacy) return DPR_OK;
    if (c->world > kNjsMaxWorld) { c->nj_exchange_note = "more ranks than mailbox slots"; return DPR_OK; }
    int ok = 1;
    for (auto& b : c->nj) {
        b.peer.plan = plan;
        if (njs_alloc_window(b, c->stream) != DPR_OK) { ok = 0; (void)hipGetLastError(); }
    }
    if (c->vworld > 0) {
        if (!ok) { c->nj_exchange_note = "window allocation failed: " + last_error(); for (auto& b : c->nj) b.peer.plan = kNjsLegacy; return DPR_OK; }
        std::vector<char*> wins((size_t)c->vworld);
        std::vector<double*> Ds((size_t)c->vworld);
        for (int r = 0; r < c->vworld; ++r) { wins[(size_t)r] = c->nj[(size_t)r].peer.win; Ds[(size_t)r] = c->nj[(size_t)r].D; }
        for (auto& b : c->nj)
            if (int rc = njs_set_peers(b, wins.data(), Ds.data(), c->stream)) return rc;
        c->nj_exchange_active = plan;
        return DPR_OK;
    }
    NjBuffers& b = c->nj[0];
    if (c->local_comm && !c->shm) {       // (ranks joined through a shared region exchange their handles themselves, below)
        if (!ok) return DPR_ERR_HIP;
        if (!b.peer.attached) { set_error("dpr_dist_matrix: ranks joined by dpr_comm_init_local need dpr_peer_export / dpr_peer_attach for this tip count first"); return DPR_ERR_STATE; }
        c->nj_exchange_active = kNjsMailbox;
        return DPR_OK;
    }
    // attach or skip: decided on GATHERED flags, never on this rank's own state -- if one rank's buffers were recreated
    // (a context re-made, nj_alloc after a failed call) while the others still hold their mappings, a rank-local test would
    // send some ranks into the all-gathers below and the others past them (advisor, round 3).  Mixed state: everybody
    // drops its mappings and attaches again.
    bool attach = true;
    {
        std::vector<uint64_t> af((size_t)c->world, 0);
        const uint64_t mine_attached = (ok && b.peer.attached) ? 1 : 0;
        if (int rc = comm_gather_host(c, &mine_attached, af.data(), sizeof(uint64_t))) return rc;
        bool all_attached = true;
        for (uint64_t f : af) all_attached = all_attached && f == 1;
        attach = !all_attached;
        if (attach && b.peer.attached) {
            for (void* m : b.peer.opened) (void)hipIpcCloseMemHandle(m);
            b.peer.opened.clear();
            b.peer.attached = false;
        }
    }
    if (attach) {
        PeerBlob mine;
        std::vector<PeerBlob> all((size_t)c->world);
        if (ok) peer_blob_of(c, &mine); else std::memset(&mine, 0, sizeof mine);
        if (int rc = comm_gather_host(c, &mine, all.data(), sizeof(PeerBlob))) return rc;
        int mapped = 0;
        if (int rc = peer_attach_blobs(c, all.data(), &mapped)) return rc;
        // second round: did every rank map every peer?
        std::vector<uint64_t> flags((size_t)c->world, 0);
        const uint64_t mf = mapped ? 1 : 0;
        if (int rc = comm_gather_host(c, &mf, flags.data(), sizeof(uint64_t))) return rc;
        bool all_ok = true;
        for (uint64_t f : flags) all_ok = all_ok && f == 1;
        if (!all_ok) {
            for (void* m : b.peer.opened) (void)hipIpcCloseMemHandle(m);
            b.peer.opened.clear();
            b.peer.attached = false;
            b.peer.plan = kNjsLegacy;
            bool guard = false;
            for (const PeerBlob& pb : all) guard = guard || pb.pad[1] == 2;
            c->nj_exchange_note = guard ? "this HIP runtime (older than 7.2) does not map IPC allocations of 2 GiB and more reliably: legacy two-exchange loop"
                                        : "peer windows could not be mapped on every rank (hipIpc): legacy two-exchange loop";
            return DPR_OK;
        }
    }
    c->nj_exchange_active = plan;
    return DPR_OK;
}

// barrier over the ranks of the sharded loop, enqueued on the context's stream
int njs_barrier(dpr_ctx* c)
{
    if (c->vworld > 0 || c->world == 1) return DPR_OK;       // one stream: already ordered
    if (c->comm) return exchange(c, EX_RECS);                // (the gathered records are dead between iterations)
    if (c->shm && c->nj_exchange_active == kNjsLegacy) return comm_barrier(c, c->stream);      // (no peer windows were set up)
    return njs_launch_barrier(c->nj[0], c->stream);
}

// row-sharded pruned NJ, ranks joined by RCCL: all ranks in step with idle streams (epoch builds)
int njr_barrier_cb(void* ctx)
{
    dpr_ctx* c = static_cast<dpr_ctx*>(ctx);
    return comm_barrier(c, c->stream);
}

NjBuffers* owner_buffers(dpr_ctx* c, int64_t row)
{
    const int o = shard_owner(row, c->world);
    if (c->vworld > 0) return &c->nj[(size_t)o];
    return o == c->rank ? &c->nj[0] : nullptr;
}

}  // namespace dpr

using namespace dpr;

extern "C" {

// ---- multi-GPU ---------------------------------------------------------------------------------------
int dpr_comm_unique_id(void* out128)
{
    if (!out128) { set_error("dpr_comm_unique_id: null"); return DPR_ERR_ARG; }
    if (int rc = rccl_load()) return rc;
    int r = g_rccl.GetUniqueId(out128);
    if (r != 0) { set_error(std::string("ncclGetUniqueId: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?")); return DPR_ERR_COMM; }
    return DPR_OK;
}

int dpr_comm_init(dpr_ctx* c, int rank, int world, const void* id128)
{
    if (!c || world < 1 || rank < 0 || rank >= world) { set_error("dpr_comm_init: bad argument"); return DPR_ERR_ARG; }
    if (c->vworld > 0) { set_error("dpr_comm_init: context holds virtual ranks"); return DPR_ERR_STATE; }
    c->rank = rank; c->world = world;
    if (world == 1) return DPR_OK;
    if (!id128) { set_error("dpr_comm_init: null id"); return DPR_ERR_ARG; }
    if (int rc = rccl_load()) return rc;
    DPR_HIP(hipSetDevice(c->device));
    Id128 id;
    std::memcpy(id.b, id128, 128);
    int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) { set_error(std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?")); return DPR_ERR_COMM; }
    return DPR_OK;
}

// Ranks WITHOUT RCCL (several processes whose GPUs -- or one shared GPU -- can map each other's memory): the row-sharded
// NJ then runs its mailbox plan, and the launcher carries the 192-byte blobs of dpr_peer_export between the processes
// (tests/test_gpu_multiproc.py does it with pipes on ONE GPU, which RCCL refuses: "duplicate GPU").
int dpr_comm_init_local(dpr_ctx* c, int rank, int world)
{
    if (!c || world < 1 || world > kNjsMaxWorld || rank < 0 || rank >= world) { set_error("dpr_comm_init_local: bad argument"); return DPR_ERR_ARG; }
    if (c->vworld > 0 || c->comm || c->shm) { set_error("dpr_comm_init_local: context already holds ranks"); return DPR_ERR_STATE; }
    c->rank = rank; c->world = world;
    c->local_comm = world > 1;
    return DPR_OK;
}

// allocate the NJ buffers and the window for n_tips on this rank and describe them (192 bytes) for the other ranks
int dpr_peer_export(dpr_ctx* c, int64_t n_tips, void* out192)
{
    if (!c || !out192 || n_tips < 2 || n_tips >= (1 << 24)) { set_error("dpr_peer_export: bad argument"); return DPR_ERR_ARG; }
    if (c->world < 2 || c->vworld > 0) { set_error("dpr_peer_export: needs a multi-rank context"); return DPR_ERR_STATE; }
    DPR_HIP(hipSetDevice(c->device));
    c->have_matrix = 0;
    NjBuffers& b = c->nj[0];
    if (int rc = nj_alloc(b, n_tips, c->rank, c->world, c->stream, ctx_njr(c, n_tips) ? njr_twin_rows(n_tips, c->world) : 0)) return rc;
    b.peer.plan = kNjsMailbox;
    if (int rc = njs_alloc_window(b, c->stream)) return rc;
    --b.peer.run_id;          // (dpr_dist_matrix's own njs_alloc_window call counts the run)
    DPR_HIP(hipStreamSynchronize(c->stream));
    PeerBlob blob;
    peer_blob_of(c, &blob);
    if (!blob.ok && blob.pad[1] == 2) {
        set_error("dpr_peer_export: this HIP runtime (older than 7.2) does not map IPC allocations of 2 GiB and more reliably; this rank's rows take " +
                  std::to_string(blob.pad[0]) + " bytes (use the system runtime, more ranks, or DPR_IPC_ANY_SIZE=1)");
        return DPR_ERR_STATE;
    }
    if (!blob.ok) { set_error("dpr_peer_export: hipIpcGetMemHandle failed"); return DPR_ERR_HIP; }
    std::memcpy(out192, &blob, sizeof blob);
    return DPR_OK;
}

// all192: the blobs of all ranks in rank order (this rank's own one is ignored)
int dpr_peer_attach(dpr_ctx* c, const void* all192)
{
    if (!c || !all192) { set_error("dpr_peer_attach: bad argument"); return DPR_ERR_ARG; }
    if (c->world < 2 || c->vworld > 0 || !c->nj[0].peer.win) { set_error("dpr_peer_attach: call dpr_peer_export first"); return DPR_ERR_STATE; }
    DPR_HIP(hipSetDevice(c->device));
    NjBuffers& b = c->nj[0];
    for (void* m : b.peer.opened) (void)hipIpcCloseMemHandle(m);
    b.peer.opened.clear();
    b.peer.attached = false;
    int ok = 0;
    if (int rc = peer_attach_blobs(c, static_cast<const PeerBlob*>(all192), &ok)) return rc;
    if (!ok) { set_error("dpr_peer_attach: a peer's buffers could not be mapped (hipIpcOpenMemHandle) or describe another tip count"); return DPR_ERR_HIP; }
    return DPR_OK;
}

// exchange plan of the row-sharded NJ loop: 0 legacy (4 launches + 2 all-gathers), 1 peer (2 launches + 1 all-gather,
// rows pulled from their owners), 2 mailbox (2 launches, no collective); -1 = DPR_NJ_EXCHANGE / default (peer)
int dpr_ctx_set_nj_exchange(dpr_ctx* c, int plan)
{
    if (!c || plan < -1 || plan > 2) { set_error("dpr_ctx_set_nj_exchange: -1 default, 0 legacy, 1 peer, 2 mailbox"); return DPR_ERR_ARG; }
    c->nj_exchange = plan;
    return DPR_OK;
}
// what the last dpr_dist_matrix set up and what the last dpr_nj_run enqueued on this rank
int dpr_get_nj_exchange_info(dpr_ctx* c, int* active_plan, int64_t* launches, int64_t* collectives, char* note, int cap)
{
    if (!c) { set_error("dpr_get_nj_exchange_info: null ctx"); return DPR_ERR_ARG; }
    if (active_plan) *active_plan = c->nj_exchange_active;
    if (launches) *launches = c->nj_launches;
    if (collectives) *collectives = c->nj_collectives;
    if (note && cap > 0) std::snprintf(note, (size_t)cap, "%s", c->nj_exchange_note.c_str());
    return DPR_OK;
}
// bound of one mailbox poll in milliseconds (default 2000): a rank whose record does not arrive ends the run with DPR_ERR_COMM
// Test hook of the one-exchange loops' cross-check: rank `rank` uses a wrong value for one element of a row it pulled at
// iteration `iteration` (-1, -1 switches it off).  The run must then end with DPR_ERR_COMM on every rank one iteration later.
// A setter of the context, not an environment variable: nothing outside a test can switch it on.
int dpr_ctx_set_debug_fault(dpr_ctx* c, int64_t iteration, int rank)
{
    if (!c) { set_error("dpr_ctx_set_debug_fault: null ctx"); return DPR_ERR_ARG; }
    for (auto& b : c->nj) {
        b.peer.fault_it = iteration; b.peer.fault_rank = rank;
        // (a captured graph of the row-sharded pruned loop holds its arguments by value: capture again with the new setting)
        if (c->nj_row_pruned && b.pr.graph) { (void)hipGraphExecDestroy(b.pr.graph); b.pr.graph = nullptr; }
    }
    return DPR_OK;
}

int dpr_ctx_set_poll_limit_ms(dpr_ctx* c, int ms)
{
    if (!c || ms < 1) { set_error("dpr_ctx_set_poll_limit_ms: ms >= 1"); return DPR_ERR_ARG; }
    for (auto& b : c->nj) {
        b.peer.poll_ticks = (unsigned long long)ms * 100000ull;
        if (c->nj_row_pruned && b.pr.graph) { (void)hipGraphExecDestroy(b.pr.graph); b.pr.graph = nullptr; }
    }
    return DPR_OK;
}

// what the communicator itself says (ncclCommCount / ncclCommUserRank), not what the caller passed to dpr_comm_init:
// bench.py reports these per leg, so that a record claiming G ranks has RCCL's word for it
int dpr_comm_info(dpr_ctx* c, int* rank, int* nranks)
{
    if (!c) { set_error("dpr_comm_info: null ctx"); return DPR_ERR_ARG; }
    if (rank) *rank = 0;
    if (nranks) *nranks = 1;
    if (!c->comm && c->shm) {                    // ranks joined through a shared region: the rank count the region itself has seen
        if (rank) *rank = c->rank;
        if (nranks) *nranks = shm_joined(c);
        return DPR_OK;
    }
    if (!c->comm) return DPR_OK;                 // no communicator: one rank
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) { set_error("librccl.so lacks ncclCommCount / ncclCommUserRank"); return DPR_ERR_COMM; }
    int r = 0, n = 0;
    if (g_rccl.CommCount(c->comm, &n) != 0 || g_rccl.CommUserRank(c->comm, &r) != 0) { set_error("ncclCommCount / ncclCommUserRank failed"); return DPR_ERR_COMM; }
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    return DPR_OK;
}

// RCCL plumbing self-test on ONE GPU: 1-rank communicator + all-gather of one record.  Exercises the
// dlopen'ed entry points, the by-value ncclUniqueId ABI and the datatype constants used by exchange().
int dpr_comm_selftest(dpr_ctx* c)
{
    if (!c) { set_error("dpr_comm_selftest: null ctx"); return DPR_ERR_ARG; }
    if (int rc = rccl_load()) return rc;
    DPR_HIP(hipSetDevice(c->device));
    Id128 id;
    int r = g_rccl.GetUniqueId(&id);
    if (r != 0) { set_error("ncclGetUniqueId failed"); return DPR_ERR_COMM; }
    void* comm = nullptr;
    r = g_rccl.CommInitRank(&comm, 1, id, 0);
    if (r != 0) { set_error(std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?")); return DPR_ERR_COMM; }
    NjRecord h{ -1.5, 42ull, 2.25, 7ull }, back{ 0, 0, 0, 0 };
    NjRecord* d = nullptr;
    double *ds = nullptr, *dg = nullptr;
    DPR_HIP(hipMalloc(&d, sizeof(NjRecord)));
    DPR_HIP(hipMalloc(&ds, sizeof(double) * 192));
    DPR_HIP(hipMalloc(&dg, sizeof(double) * 192));
    std::vector<double> hs(192), hg(192, 0.0);
    for (int i = 0; i < 192; ++i) hs[(size_t)i] = 0.5 * i;
    DPR_HIP(hipMemcpy(d, &h, sizeof(NjRecord), hipMemcpyHostToDevice));
    DPR_HIP(hipMemcpy(ds, hs.data(), sizeof(double) * 192, hipMemcpyHostToDevice));
    r = g_rccl.AllGather(d, d, sizeof(NjRecord), kNcclUint8, comm, c->stream);          // in place
    if (r == 0) r = g_rccl.AllGather(ds, dg, 192, kNcclFloat64, comm, c->stream);
    // the all-reduces of the multi-GPU divide-and-conquer path: in place, uint64 / int32 sums
    if (r == 0 && g_rccl.AllReduce) r = g_rccl.AllReduce(ds, ds, 192, kNcclUint64, kNcclSum, comm, c->stream);
    if (r == 0 && g_rccl.AllReduce) r = g_rccl.AllReduce(ds, ds, 384, kNcclInt32, kNcclSum, comm, c->stream);
    if (r == 0 && !g_rccl.AllReduce) r = -1;
    DPR_HIP(hipStreamSynchronize(c->stream));
    std::vector<double> hr(192, -1.0);
    DPR_HIP(hipMemcpy(hr.data(), ds, sizeof(double) * 192, hipMemcpyDeviceToHost));
    DPR_HIP(hipMemcpy(&back, d, sizeof(NjRecord), hipMemcpyDeviceToHost));
    DPR_HIP(hipMemcpy(hg.data(), dg, sizeof(double) * 192, hipMemcpyDeviceToHost));
    (void)hipFree(d); (void)hipFree(ds); (void)hipFree(dg);
    g_rccl.CommDestroy(comm);
    if (r != 0) { set_error("ncclAllGather / ncclAllReduce failed"); return DPR_ERR_COMM; }
    if (back.q != h.q || back.key != h.key || back.d != h.d || hg != hs || hr != hs) { set_error("dpr_comm_selftest: data mismatch"); return DPR_ERR_COMM; }
    return DPR_OK;
}

}  // extern "C"
