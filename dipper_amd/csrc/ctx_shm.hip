// C ABI, ranks joined through a SHARED HOST REGION: the transport of `dipper --gpus G` (dipper_amd/host/main.cpp forks its ranks
// around one anonymous shared mapping) and of any launcher that can hand G processes of one node the same 64 KiB of shared
// memory.  No reference counterpart: the reference drives one device (src/tree_generation.cu:240-245).
//
// The region carries (a) a sense-reversing barrier with a failure word (a rank that fails -- or the launcher, when a rank
// dies -- sets it and every barrier returns DPR_ERR_COMM instead of waiting for ever), (b) one 512-byte slot per rank for
// host-side all-gathers of small blobs (the RCCL unique id, hipIpc handles, device identities), and, for the `ipc` transport,
// (c) nothing more: the data path is a device window per rank (hipMalloc, mapped by every other rank through hipIpc), through
// which in-place all-gathers and integer all-reduces run in chunks, bracketed by host barriers.
//
// Transports:  rccl = the region only carries the unique id; collectives are RCCL's (ctx_comm.hip) -- the plan for one rank per GPU;
//              ipc  = the window collectives below -- ranks that SHARE a device (RCCL refuses them: the single-GPU rehearsal of every
//                     multi-rank path), or a node without a usable RCCL;  auto = ipc iff two ranks name the same device.
// The dispatchers comm_all_gather / comm_all_reduce_sum / comm_gather_host / comm_barrier are what the algorithms call
// (ctx_place.hip, ctx_nj.hip, ctx_comm.hip): RCCL when the context holds a communicator, the windows when it holds a ShmComm.
#include "ctx_internal.hpp"

#include <sched.h>
#include <time.h>
#include <unistd.h>

namespace dpr {

constexpr uint32_t kShmMagic = 0x44505253u;      // "DPRS"
constexpr int kShmSlotBytes = 512;
struct ShmHeader {
    uint32_t magic, world;
    uint32_t joined;                 // ranks that have called dpr_comm_init_shared
    uint32_t failed;                 // != 0: a rank (or the launcher) gave up -- every wait ends with DPR_ERR_COMM
    uint32_t bar_count, bar_sense;
    uint32_t id_ready, pad0;
    char rccl_id[128];
    char pad1[512 - 32 - 128];
    char slots[kNjsMaxWorld][kShmSlotBytes];
};
static_assert(sizeof(ShmHeader) <= DPR_COMM_SHARED_BYTES, "shared region layout");

struct ShmComm {
    ShmHeader* h = nullptr;
    int rank = 0, world = 1;
    uint32_t sense = 0;
    int64_t timeout_ms = 600000;
    // ipc transport
    char* win = nullptr;
    size_t win_bytes = 0;
    std::vector<char*> peer_win;     // [world]; own entry = win
    int64_t collectives = 0;
};

static inline uint32_t ld_acq(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline void st_rel(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static double now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}
// wait until pred() or the failure word or the time limit; the waiting ranks yield (G ranks may share a few host cores)
template <class Pred> static int shm_wait(ShmHeader* h, int64_t timeout_ms, const char* what, Pred pred)
{
    const double t0 = now_ms();
    for (unsigned spin = 0;; ++spin) {
        if (pred()) return DPR_OK;
        if (ld_acq(&h->failed)) { set_error(std::string(what) + ": another rank failed (or the launcher gave up)"); return DPR_ERR_COMM; }
        if (spin < 2000) sched_yield();
        else {
            usleep(spin < 20000 ? 20 : 200);
            if ((spin & 255) == 0 && now_ms() - t0 > (double)timeout_ms) {
                st_rel(&h->failed, 1u);
                set_error(std::string(what) + ": timed out waiting for the other ranks (DPR_COMM_TIMEOUT_MS)");
                return DPR_ERR_COMM;
            }
        }
    }
}
static int shm_barrier_raw(ShmHeader* h, int world, uint32_t* sense, int64_t timeout_ms)
{
    if (world <= 1) return DPR_OK;
    const uint32_t mine = (*sense ^= 1u);
    if (__atomic_add_fetch(&h->bar_count, 1u, __ATOMIC_ACQ_REL) == (uint32_t)world) {
        __atomic_store_n(&h->bar_count, 0u, __ATOMIC_RELAXED);
        st_rel(&h->bar_sense, mine);
        return DPR_OK;
    }
    return shm_wait(h, timeout_ms, "barrier", [&] { return ld_acq(&h->bar_sense) == mine; });
}
static int shm_gather_raw(ShmHeader* h, int rank, int world, uint32_t* sense, int64_t timeout_ms, const void* mine, void* all, size_t bytes)
{
    if (bytes > (size_t)kShmSlotBytes) { set_error("shared-region gather: more than 512 bytes per rank"); return DPR_ERR_ARG; }
    std::memcpy(h->slots[rank], mine, bytes);
    if (int rc = shm_barrier_raw(h, world, sense, timeout_ms)) return rc;
    for (int r = 0; r < world; ++r) std::memcpy(static_cast<char*>(all) + (size_t)r * bytes, h->slots[r], bytes);
    return shm_barrier_raw(h, world, sense, timeout_ms);      // nobody overwrites its slot before everybody has read it
}

template <class T> __global__ void comm_add_kernel(T* __restrict__ dst, const T* __restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

static int shm_barrier(dpr_ctx* c) { return shm_barrier_raw(c->shm->h, c->shm->world, &c->shm->sense, c->shm->timeout_ms); }

// in-place all-gather through the windows: rank r's segment lives at buf + r * seg
static int shm_all_gather(dpr_ctx* c, char* buf, size_t seg, hipStream_t s)
{
    ShmComm& m = *c->shm;
    ++m.collectives;
    for (size_t off = 0; off < seg; off += m.win_bytes) {
        const size_t nb = seg - off < m.win_bytes ? seg - off : m.win_bytes;
        DPR_HIP(hipMemcpyAsync(m.win, buf + (size_t)m.rank * seg + off, nb, hipMemcpyDeviceToDevice, s));
        DPR_HIP(hipStreamSynchronize(s));
        if (int rc = shm_barrier(c)) return rc;                   // every window holds its rank's chunk
        for (int k = 1; k < m.world; ++k) {
            const int r = (m.rank + k) % m.world;                    // (ranks start at different peers)
            DPR_HIP(hipMemcpyAsync(buf + (size_t)r * seg + off, m.peer_win[(size_t)r], nb, hipMemcpyDeviceToDevice, s));
        }
        DPR_HIP(hipStreamSynchronize(s));
        if (int rc = shm_barrier(c)) return rc;                   // every rank has read: the windows may be overwritten
    }
    return DPR_OK;
}
// in-place sum over the ranks (integer types only: the order of the summands differs between the ranks)
template <class T> static int shm_all_reduce(dpr_ctx* c, T* buf, size_t count, hipStream_t s)
{
    ShmComm& m = *c->shm;
    ++m.collectives;
    const size_t per = m.win_bytes / sizeof(T);
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        DPR_HIP(hipMemcpyAsync(m.win, buf + off, n * sizeof(T), hipMemcpyDeviceToDevice, s));
        DPR_HIP(hipStreamSynchronize(s));
        if (int rc = shm_barrier(c)) return rc;
        size_t grid = (n + 255) / 256;
        if (grid > 4096) grid = 4096;
        for (int k = 1; k < m.world; ++k) {
            const int r = (m.rank + k) % m.world;
            hipLaunchKernelGGL(comm_add_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, buf + off, reinterpret_cast<const T*>(m.peer_win[(size_t)r]), n);
        }
        DPR_HIP(hipGetLastError());
        DPR_HIP(hipStreamSynchronize(s));
        if (int rc = shm_barrier(c)) return rc;
    }
    return DPR_OK;
}

void shm_comm_free(dpr_ctx* c)
{
    if (!c->shm) return;
    ShmComm& m = *c->shm;
    for (int r = 0; r < (int)m.peer_win.size(); ++r)
        if (r != m.rank && m.peer_win[(size_t)r]) (void)hipIpcCloseMemHandle(m.peer_win[(size_t)r]);
    if (m.win) (void)hipFree(m.win);
    delete c->shm;
    c->shm = nullptr;
}

int shm_joined(const dpr_ctx* c) { return c->shm ? (int)ld_acq(&c->shm->h->joined) : 1; }

// ---- dispatchers: what the algorithms call ------------------------------------------------------------------------------------
bool comm_real(const dpr_ctx* c) { return c->world > 1 && c->vworld == 0 && (c->comm != nullptr || (c->shm != nullptr && c->shm->win != nullptr)); }

int comm_gather_host(dpr_ctx* c, const void* mine, void* all, size_t bytes)
{
    if (c->shm) return shm_gather_raw(c->shm->h, c->shm->rank, c->shm->world, &c->shm->sense, c->shm->timeout_ms, mine, all, bytes);
    return rccl_gather_bytes(c, mine, all, bytes);
}

int comm_barrier(dpr_ctx* c, hipStream_t s)
{
    if (c->comm) {
        DPR_HIP(hipStreamSynchronize(s));
        if (int rc = exchange(c, EX_RECS)) return rc;
        DPR_HIP(hipStreamSynchronize(c->stream));
        return DPR_OK;
    }
    if (c->shm) { DPR_HIP(hipStreamSynchronize(s)); return shm_barrier(c); }
    set_error("comm_barrier: this context has no transport between its ranks");
    return DPR_ERR_COMM;
}

// all-gather of seg_bytes per rank into recv (rank r's segment at recv + r * seg_bytes); send may be this rank's own segment of
// recv (in place) or another buffer
int comm_all_gather(dpr_ctx* c, const void* send, void* recv, size_t seg_bytes, hipStream_t s)
{
    ++c->comm_collectives;
    if (c->comm) {
        if (g_rccl.AllGather(send, recv, seg_bytes, kNcclUint8, c->comm, s) != 0) { set_error("ncclAllGather failed"); return DPR_ERR_COMM; }
        return DPR_OK;
    }
    if (c->shm && c->shm->win) {
        char* own = static_cast<char*>(recv) + (size_t)c->rank * seg_bytes;
        if (send != own) DPR_HIP(hipMemcpyAsync(own, send, seg_bytes, hipMemcpyDeviceToDevice, s));
        return shm_all_gather(c, static_cast<char*>(recv), seg_bytes, s);
    }
    set_error("all-gather: this context has no transport between its ranks (dpr_comm_init / dpr_comm_init_shared)");
    return DPR_ERR_COMM;
}

int comm_all_reduce_sum(dpr_ctx* c, void* buf, size_t count, int nccl_type, hipStream_t s)
{
    if (nccl_type != kNcclInt32 && nccl_type != kNcclUint64) { set_error("all-reduce: int32 / uint64 sums only"); return DPR_ERR_ARG; }
    ++c->comm_collectives;
    if (c->comm) {
        if (!g_rccl.AllReduce) { set_error("librccl.so lacks ncclAllReduce"); return DPR_ERR_COMM; }
        if (g_rccl.AllReduce(buf, buf, count, nccl_type, kNcclSum, c->comm, s) != 0) { set_error("ncclAllReduce failed"); return DPR_ERR_COMM; }
        return DPR_OK;
    }
    if (c->shm && c->shm->win)
        return nccl_type == kNcclInt32 ? shm_all_reduce(c, static_cast<int32_t*>(buf), count, s) : shm_all_reduce(c, static_cast<unsigned long long*>(buf), count, s);
    set_error("all-reduce: this context has no transport between its ranks (dpr_comm_init / dpr_comm_init_shared)");
    return DPR_ERR_COMM;
}

}  // namespace dpr

using namespace dpr;

extern "C" {

// host-only helpers over a shared region (CPU tests of the protocol; the launcher's failure path)
int dpr_shared_barrier(void* shared, int world, uint32_t* sense, int timeout_ms)
{
    if (!shared || !sense || world < 1 || world > kNjsMaxWorld) { set_error("dpr_shared_barrier: bad argument"); return DPR_ERR_ARG; }
    return shm_barrier_raw(static_cast<ShmHeader*>(shared), world, sense, timeout_ms);
}
int dpr_shared_gather(void* shared, int rank, int world, uint32_t* sense, int timeout_ms, const void* mine, void* all, int bytes)
{
    if (!shared || !sense || !mine || !all || world < 1 || world > kNjsMaxWorld || rank < 0 || rank >= world || bytes < 0) { set_error("dpr_shared_gather: bad argument"); return DPR_ERR_ARG; }
    return shm_gather_raw(static_cast<ShmHeader*>(shared), rank, world, sense, timeout_ms, mine, all, (size_t)bytes);
}
int dpr_shared_abort(void* shared)
{
    if (!shared) { set_error("dpr_shared_abort: null"); return DPR_ERR_ARG; }
    st_rel(&static_cast<ShmHeader*>(shared)->failed, 1u);
    return DPR_OK;
}
int dpr_shared_failed(const void* shared) { return shared && ld_acq(&static_cast<const ShmHeader*>(shared)->failed) ? 1 : 0; }

// transport of this context's ranks (0 none: one rank or virtual ranks, 1 RCCL, 2 device windows over hipIpc, 3 launcher-attached
// peers without a collective transport: dpr_comm_init_local) and the device collectives it has taken part in so far
int dpr_comm_stats(dpr_ctx* c, int* transport, int64_t* collectives)
{
    if (!c) { set_error("dpr_comm_stats: null ctx"); return DPR_ERR_ARG; }
    if (transport) *transport = c->comm ? 1 : (c->shm && c->shm->win) ? 2 : c->local_comm ? 3 : 0;
    if (collectives) *collectives = c->comm_collectives;
    return DPR_OK;
}

int dpr_comm_init_shared(dpr_ctx* c, int rank, int world, void* shared, uint64_t bytes, int transport)
{
    if (!c || !shared || world < 1 || world > kNjsMaxWorld || rank < 0 || rank >= world || bytes < DPR_COMM_SHARED_BYTES || transport < 0 || transport > 2) {
        set_error("dpr_comm_init_shared: bad argument (1 <= world <= 64, DPR_COMM_SHARED_BYTES of zero-initialised shared memory, transport 0 auto / 1 rccl / 2 ipc)");
        return DPR_ERR_ARG;
    }
    if (c->vworld > 0 || c->comm || c->shm || c->local_comm) { set_error("dpr_comm_init_shared: context already holds ranks"); return DPR_ERR_STATE; }
    c->rank = rank; c->world = world;
    if (world == 1) return DPR_OK;
    DPR_HIP(hipSetDevice(c->device));
    ShmHeader* h = static_cast<ShmHeader*>(shared);
    {   // the first rank to arrive stamps the region; everybody checks the rank count
        uint32_t zero = 0;
        if (__atomic_compare_exchange_n(&h->magic, &zero, kShmMagic, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) __atomic_store_n(&h->world, (uint32_t)world, __ATOMIC_RELEASE);
        else if (zero != kShmMagic) { set_error("dpr_comm_init_shared: the shared region is not zero-initialised"); return DPR_ERR_ARG; }
    }
    ShmComm* m = new ShmComm;
    m->h = h; m->rank = rank; m->world = world;
    if (const char* e = std::getenv("DPR_COMM_TIMEOUT_MS")) { const long long v = std::atoll(e); if (v >= 100) m->timeout_ms = v; }
    c->shm = m;
    __atomic_add_fetch(&h->joined, 1u, __ATOMIC_ACQ_REL);
    auto fail = [&](int rc) { st_rel(&h->failed, 1u); return rc; };
    // device identities: which ranks share a device decides `auto`, and every rank must name the same world
    struct Ident { uint32_t world; int32_t device; char bus[64]; } mine, all[kNjsMaxWorld];
    std::memset(&mine, 0, sizeof mine);
    mine.world = (uint32_t)world; mine.device = c->device;
    if (hipDeviceGetPCIBusId(mine.bus, (int)sizeof(mine.bus) - 1, c->device) != hipSuccess) { (void)hipGetLastError(); std::snprintf(mine.bus, sizeof mine.bus, "device%d", c->device); }
    if (int rc = comm_gather_host(c, &mine, all, sizeof(Ident))) return rc;
    bool shared_device = false;
    for (int a = 0; a < world; ++a) {
        if (all[a].world != (uint32_t)world) { set_error("dpr_comm_init_shared: the ranks disagree about the number of ranks"); return fail(DPR_ERR_ARG); }
        for (int b = a + 1; b < world; ++b) shared_device = shared_device || std::strcmp(all[a].bus, all[b].bus) == 0;
    }
    // (DPR_TEST_COMM_TRY_RCCL=1, tests: `auto` tries RCCL although ranks share a device -- RCCL refuses them, which exercises the
    //  joint fall-back to the device windows on a one-GPU box)
    if (transport == 0 && shared_device && std::getenv("DPR_TEST_COMM_TRY_RCCL")) shared_device = false;
    const int tr = transport != 0 ? transport : (shared_device ? 2 : 1);
    if (tr == 1) {
        if (shared_device) { set_error("dpr_comm_init_shared: RCCL refuses two ranks on one device; use the ipc transport"); return fail(DPR_ERR_ARG); }
        // rank 0's unique id travels through the region (id_ready: 1 = there, 2 = rank 0 could not make one)
        if (rank == 0) {
            char id[128];
            const int rc0 = dpr_comm_unique_id(id);
            if (rc0 == DPR_OK) std::memcpy(h->rccl_id, id, 128);
            st_rel(&h->id_ready, rc0 == DPR_OK ? 1u : 2u);
        } else if (int rc = shm_wait(h, m->timeout_ms, "dpr_comm_init_shared (RCCL id)", [&] { return ld_acq(&h->id_ready) != 0; })) return rc;
        // (the ShmComm stays: host gathers, barriers and the failure word go through the region; the data path is RCCL's)
        uint32_t ok = 0, oks[kNjsMaxWorld];
        if (ld_acq(&h->id_ready) == 1u) {
            char id[128];
            std::memcpy(id, h->rccl_id, 128);
            ok = dpr_comm_init(c, rank, world, id) == DPR_OK ? 1u : 0u;
        }
        if (int rc = comm_gather_host(c, &ok, oks, sizeof(uint32_t))) return rc;
        bool all_ok = true;
        for (int r = 0; r < world; ++r) all_ok = all_ok && oks[r] != 0;
        if (all_ok) return shm_barrier(c);
        // RCCL did not come up on every rank (no librccl, a bootstrap interface that does not answer, ...): with `auto` every rank
        // falls back TOGETHER -- the decision is taken on gathered flags -- to the device windows, which need nothing but hipIpc between
        // the node's GPUs; an explicit `rccl` is an error
        const std::string why = last_error();
        if (c->comm && g_rccl.CommDestroy) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
        if (transport == 1) { set_error("dpr_comm_init_shared: RCCL did not come up on every rank (" + why + ")"); return fail(DPR_ERR_COMM); }
        if (rank == 0) std::fprintf(stderr, "[dipper] RCCL did not come up on every rank (%s): the ranks exchange through device windows over hipIpc instead\n", why.c_str());
    }
    // ipc transport: one device window per rank, mapped by every other rank
    size_t win_mb = 64;
    if (const char* e = std::getenv("DPR_COMM_WINDOW_MB")) { const long long v = std::atoll(e); if (v >= 1 && v <= 1024) win_mb = (size_t)v; }
    m->win_bytes = win_mb << 20;
    struct Handle { uint32_t ok, pad; hipIpcMemHandle_t hdl; } hm, ha[kNjsMaxWorld];
    std::memset(&hm, 0, sizeof hm);
    if (hipMalloc(&m->win, m->win_bytes) == hipSuccess && hipIpcGetMemHandle(&hm.hdl, m->win) == hipSuccess) hm.ok = 1;
    else (void)hipGetLastError();
    if (int rc = comm_gather_host(c, &hm, ha, sizeof(Handle))) return rc;
    m->peer_win.assign((size_t)world, nullptr);
    m->peer_win[(size_t)rank] = m->win;
    uint32_t ok = hm.ok;
    for (int r = 0; r < world && ok; ++r) {
        if (!ha[r].ok) { ok = 0; break; }
        if (r == rank) continue;
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, ha[r].hdl, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = 0; break; }
        m->peer_win[(size_t)r] = static_cast<char*>(p);
    }
    uint32_t oks[kNjsMaxWorld];
    if (int rc = comm_gather_host(c, &ok, oks, sizeof(uint32_t))) return rc;
    for (int r = 0; r < world; ++r)
        if (!oks[r]) { set_error("dpr_comm_init_shared: rank " + std::to_string(r) + " could not allocate or map the device windows (hipIpc)"); return fail(DPR_ERR_HIP); }
    c->local_comm = true;            // (the row-sharded NJ plans: mailbox exchange, peer buffers attached by the library itself)
    return DPR_OK;
}

}  // extern "C"
