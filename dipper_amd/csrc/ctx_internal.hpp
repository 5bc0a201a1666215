// Shared by the translation units that implement the C ABI (ctx.hip: context, inputs, hooks; ctx_comm.hip: RCCL, peer
// windows, exchanges; ctx_nj.hip: distance matrix + NJ plans; ctx_place.hip: placement, exact mode, divide-and-conquer).
#pragma once
#include "dpr_internal.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Id128 { char b[128]; };  // ncclUniqueId is 128 opaque bytes passed by value

namespace dpr {
// ---- RCCL, resolved at run time so that the single-GPU path has no link dependency ----------------
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
extern Rccl g_rccl;
int rccl_load();
const std::string& last_error();
struct ShmComm;                     // ranks joined through a shared host region (ctx_shm.hip)
}  // namespace dpr

struct dpr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // distance rows of the next placement batch (created on first use)
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    int rank = 0, world = 1;  // RCCL rank/world, or world = number of virtual ranks
    int vworld = 0;           // > 0: all ranks live in this context on one device (validation mode)
    void* comm = nullptr;
    dpr::ShmComm* shm = nullptr;     // dpr_comm_init_shared: barrier / host gathers through the region; + device windows (ipc transport)
    int64_t comm_collectives = 0;    // device collectives (all-gathers, all-reduces) this context has taken part in, any transport
    std::vector<dpr::NjBuffers> nj = std::vector<dpr::NjBuffers>(1);  // one per rank held here
    dpr::MsaBuffers msa;
    dpr::MashBuffers mash;
    dpr::PlaceBuffers place;
    dpr::ExactBuffers exact;
    double* place_trace = nullptr;   // [3N] (eid, frac, add) per placed tip
    double* packed_lower = nullptr;  // MATRIX source, device
    int64_t n_input = 0;
    int have_matrix = 0;
    bool nj_replicated = false;      // several ranks, each holding the whole matrix (pruned NJ)
    bool nj_unit_sharded = false;    // ... and sharing the unit tests / scans of an iteration (else: every rank runs the single-GPU plan)
    bool nj_row_pruned = false;      // several ranks, rows sharded, exact pruned NJ (njr.hip)
    double dist_ms = 0, nj_ms = 0;
    double place_dist_ms = 0;        // distance rows of the last placement run (the rest of nj_ms is tree work)
    std::vector<hipEvent_t> place_ev;   // event pairs whose sum is the reported distance part of the current placement run
    std::vector<hipEvent_t> place_ev_busy;   // overlap mode: event pairs around the distance batches on the second stream
    double place_dist_busy_ms = 0;      // overlap mode: time the distance batches were in flight beside the tree kernels
    bool place_overlapped = false;      // some batch of the last placement run was produced beside the tree kernels
    std::vector<hipEvent_t> place_ev_tree;   // per-batch event pairs around the tree kernels (the overlap policy's probes)
    int64_t place_batches = 0, place_batches_overlapped = 0;      // of the last placement run
    dpr::DcStats dc_stats;
    double dc_ms[3] = { 0, 0, 0 };   // backbone, cluster assignment, cluster trees
    // plan knobs of THIS context (dpr_ctx_set_*); -1 = follow the process-wide default (dpr_set_* / environment)
    int nj_mode = -1, nj_vshards = -1, nj_multi_plan = -1;
    int nj_adaptive = -1;            // adaptive pruned / streaming plan of the single-rank NJ (-1 = DPR_NJ_ADAPTIVE, default on)
    // row-sharded streaming NJ: exchange plan of the loop (-1 = DPR_NJ_EXCHANGE, default peer; see njs.hip) and what the
    // last dpr_dist_matrix actually set up (a failed peer set-up falls back to the legacy loop and says why)
    int nj_exchange = -1;
    int nj_exchange_active = dpr::kNjsLegacy;
    std::string nj_exchange_note;
    bool local_comm = false;         // ranks joined by dpr_comm_init_local: no RCCL, windows attached by the launcher
    bool njs_pending = false;        // the rows of the last merge still live in the row buffers
    int64_t nj_launches = 0, nj_collectives = 0;     // of the last dpr_nj_run (per rank)
    dpr::NjKernelTiming nj_kt;
};

namespace dpr {
constexpr int kNcclUint8 = 1, kNcclFloat64 = 8, kNcclInt32 = 2, kNcclUint64 = 5, kNcclSum = 0;
enum ExKind { EX_RECS, EX_SLICES, EX_U, EX_RECS64 /* rank records of the one-exchange loop (NjsRec) */ };
// ctx_nj.hip: plan selection
bool want_pruned(const dpr_ctx* c);
int ctx_exchange_plan(const dpr_ctx* c);
int ctx_multi_plan(const dpr_ctx* c);
int ctx_vshards(const dpr_ctx* c);
bool ctx_njr(const dpr_ctx* c, int64_t n);
int64_t njr_twin_rows(int64_t n, int world);
int fetch_state(dpr_ctx* c, NjState* st);
// ctx_comm.hip: exchanges, peer windows, barriers
int exchange(dpr_ctx* c, ExKind kind);
int njs_setup(dpr_ctx* c, bool force_windows = false);
int njs_barrier(dpr_ctx* c);
int njr_barrier_cb(void* ctx);
int njr_gather_cb(void* ctx, int kind, hipStream_t s);
int njp_gather_cb(void* ctx, void* buf, size_t bytes_per_rank, hipStream_t s);
NjBuffers* owner_buffers(dpr_ctx* c, int64_t row);
int rccl_gather_bytes(dpr_ctx* c, const void* mine, void* all, size_t bytes);
// ctx_shm.hip: the transport-independent collectives the algorithms call (RCCL communicator, or the device windows of ranks
// joined through a shared host region)
bool comm_real(const dpr_ctx* c);      // several real ranks AND a data transport between them
int comm_gather_host(dpr_ctx* c, const void* mine, void* all, size_t bytes);                         // host blobs, <= 512 bytes per rank
int comm_all_gather(dpr_ctx* c, const void* send, void* recv, size_t seg_bytes, hipStream_t s);      // rank r's segment at recv + r * seg_bytes
int comm_all_reduce_sum(dpr_ctx* c, void* buf, size_t count, int nccl_type, hipStream_t s);           // in place; kNcclInt32 / kNcclUint64
int comm_barrier(dpr_ctx* c, hipStream_t s);
void shm_comm_free(dpr_ctx* c);
int shm_joined(const dpr_ctx* c);      // ranks that have joined the context's shared region
}  // namespace dpr
