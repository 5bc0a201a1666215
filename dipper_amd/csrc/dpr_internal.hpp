// Internal declarations shared by the HIP translation units of libdipper_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/dipper_hip.h"

namespace dpr {

constexpr int kRowBlock = DPR_ROW_BLOCK;  // rows per ownership block == rows per scan tile
constexpr int kTileCols = 512;            // columns per scan tile (256 lanes x 2 doubles)
constexpr int kScanBlocks = 8192;         // capacity of the partials array (scan grid <= this)
constexpr int kThreads = 256;

void set_error(const std::string& msg);
// stderr logging of the library, ONE variable: DPR_LOG = comma list of categories, optionally with a level: epoch[=1|2|3]
// (pruned NJ: epoch rebuilds and graph captures; 2: + units listed per watched window; 3: + laps of a rebuild), mash (kernel
// choice, index sizes), import (backbone import rounds), cli (start-up and phase laps of the `dipper` command).  0 = off.
int log_level(const char* category);
int hip_fail(hipError_t e, const char* what);
#define DPR_HIP(call)                                            \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return dpr::hip_fail(e__, #call); \
    } while (0)

// ---- sharding (block-cyclic by kRowBlock rows) -------------------------------------------------
__host__ __device__ inline int shard_owner(int64_t row, int world)
{
    return (int)((row / kRowBlock) % world);
}
__host__ __device__ inline int64_t shard_local_row(int64_t row, int world)
{
    return (row / kRowBlock / world) * kRowBlock + row % kRowBlock;
}
__host__ __device__ inline int64_t shard_global_row(int64_t local, int rank, int world)
{
    return ((local / kRowBlock) * world + rank) * kRowBlock + local % kRowBlock;
}
__host__ __device__ inline int64_t shard_rows(int64_t n, int rank, int world)
{
    // number of slots < n owned by rank
    int64_t nblk = n / kRowBlock, rem = n % kRowBlock;
    int64_t full = nblk / world + ((nblk % world) > rank ? 1 : 0);
    int64_t cnt = full * kRowBlock;
    if (rem && (nblk % world) == rank) cnt += rem;
    return cnt;
}

// ---- tie-break key of the reference's findMinDist + thrust::min_element -------------------------
// (src/neighborJoining.cu:124-146,214): preference (band(i), j mod 256, j, i).
__host__ __device__ inline uint64_t nj_band(int64_t i, int64_t n)
{
    int64_t sz0 = n / 256, rem = n % 256, thr = (sz0 + 1) * rem;
    return (uint64_t)(i < thr ? i / (sz0 + 1) : rem + (i - thr) / sz0);
}
__host__ __device__ inline uint64_t nj_key_a(int64_t i, int64_t n)  // part depending on "i"
{
    return (nj_band(i, n) << 56) | (uint64_t)i;
}
__host__ __device__ inline uint64_t nj_key_b(int64_t j)  // part depending on "j"
{
    return ((uint64_t)(j & 255) << 48) | ((uint64_t)j << 24);
}

struct alignas(32) NjRecord {
    double q;
    uint64_t key;
    double d;
    uint64_t pad;
};

struct NjState {
    int64_t n;       // active size
    int64_t it;      // iterations done
    int32_t x, y;    // slots merged in the current iteration (x<y)
    double d;        // D[x][y]
    double q;        // winning Q value
    int32_t status;  // 0 ok, 1 = no candidate (DPR_ERR_NOCAND)
    int32_t pad;     // njs.hip: 1 once a failed rank has announced its status to the other ranks (it does so once)
    // pruned path (graph-replayed kernels read their iteration from memory): `it` is advanced by the post kernel while
    // it runs, so the post kernel reads `itb`, which the scan kernel (that reads `it`) publishes for it
    int64_t itb;
    int64_t it_limit;  // iterations >= it_limit are no-ops
    int64_t N;         // tips
    unsigned long long cnt_list[4];   // units listed for the scan of iteration it: cnt_list[it % 3]
    unsigned long long units_scanned; // statistics
    // pnew[it & 1]: position of the node created by merge it-1 (-1: none).  During iteration it that node is in
    // QUARANTINE: its row sum is not materialised yet (Ur = NaN, so the unit scans skip it), its pairs are evaluated
    // by the scan kernel's new-row blocks from the row buffer R[(it - 1) & 1]
    int32_t pnew[2];
};

// position-space state of the pruned path (njp.hip)
struct NjPruned {
    bool active = false;
    int64_t P = 0, ld = 0;      // positions of this epoch, row stride
    double* D = nullptr;        // [P][ld] position space
    double* U = nullptr;        // [2][vstride] by position, double-buffered: iteration it reads U + (it & 1) * vstride
    double* Ur = nullptr;       // by position; NaN for dead positions and for the node in quarantine
    double* R = nullptr;        // [2][vstride] row of the node created by merge it: R + (it & 1) * vstride
    int64_t vstride = 0;
    uint64_t *KA = nullptr, *KB = nullptr;  // key parts from the reference slot of each position
    // njp_post2_kernel (large shape): header + arrays its producer blocks hand to its test blocks (in the epoch's slab)
    bool range_known = false;         // t2_hdr holds the range of every entry of this run (njp_range_kernel + the M parts since)
    char* t2_hdr = nullptr; double *t2_rmax = nullptr, *t2_cmax = nullptr, *t2_colmin = nullptr, *t2_rowmin = nullptr, *t2_cmin = nullptr;
    int32_t *slot_of_pos = nullptr, *pos_of_slot = nullptr, *perm = nullptr;   // slot_of_pos < 0: dead position
    uint64_t* umin = nullptr;   // [strips][groups][4] order-encoded lower bound of D per sub-unit
    int64_t nunits_alloc = 0, utot = 0;
    int64_t utot0 = 0;          // units of one full scan at the first epoch (statistics)
    bool fresh = false;         // epoch just built: the first scan is a full one (njp_list_all_kernel), nothing in quarantine
    // unit-sharded mode (several GPUs, each holding the whole position-space matrix): unit ownership by
    // (strip * G16 + group) % sh_world; sh_virtual: all ranks are emulated in this process (validation)
    int sh_world = 1, sh_rank = 0;
    bool sh_virtual = false;
    int (*gather)(void* ctx, void* buf, size_t bytes_per_rank, hipStream_t s) = nullptr;   // in-place all-gather of the block records
    void* gather_ctx = nullptr;
    unsigned long long* cnt_all = nullptr;   // [local ranks][4] list counters (index it % 3)
    int64_t list_stride = 0;
    // Arena, kept until nj_free (a context that builds a matrix of the same size again allocates nothing):
    // one matrix buffer of the pruned path's own -- the epochs alternate between it and NjBuffers::D, whose tip-order
    // contents are dead once epoch 0 is built -- and two slabs for the per-epoch vectors (an epoch rebuild reads the
    // old epoch's vectors while it writes the new one's).
    double* arena_D = nullptr;
    char* arena_slab[2] = { nullptr, nullptr };
    size_t arena_slab_bytes = 0;
    int64_t arena_N = 0;
    int arena_ranks = 0;
    int epoch_index = 0;             // epoch e uses slab e & 1; its matrix lives in arena_D for even e, in NjBuffers::D for odd e
    hipGraphExec_t graph = nullptr;   // graph_iters iterations of (scan, post)
    // plan of THIS context (set by njp_build; two contexts of one process may differ and run on different host threads)
    int scan_grid = 512;              // blocks of the unit scan (DPR_NJP_GRID)
    int graph_iters = 32;             // iterations per captured hipGraph (DPR_NJ_GRAPH_ITERS)
    // Adaptive plan (single rank): the exact pruned scan only pays while its bounds prune.  The host looks at the units
    // listed per iteration after the first graph of an epoch and then every 2 048 iterations; once more than stream_frac of
    // the epoch's units are listed on average, the run is HANDED OVER to the streaming loop of nj.hip: the live positions
    // are permuted back into a dense slot-space matrix (one n^2 copy) and the reference's own algorithm -- one full Q scan
    // per iteration, last slot moved into y -- continues from the same state (same keys, same update arithmetic, so the
    // same merge log).  Every so often (after the active size has shrunk by the epoch factor once, twice, four times, ...
    // while the probes keep failing) a fresh pruned epoch is built from the slot-space matrix and probed again.
    int adaptive = 1;                 // DPR_NJ_ADAPTIVE=0 / dpr_ctx_set_nj_adaptive
    double stream_frac = 0.7;         // DPR_NJ_STREAM_FRAC (tests force the hand-over with 0); a listed unit costs ~1.6 x a streamed one, part of its sub-units are skipped
    bool slots_mode = false;          // the run currently lives in slot space (b.D, b.U, b.Ur, b.KA: nj.hip's loop)
    int64_t slots_probe_n = 0;        // slots mode: build and probe a pruned epoch again once the active size is <= this
    int probe_fail_streak = 0;
    int64_t stream_iterations = 0, stream_epochs = 0;     // iterations run by the streaming loop / hand-overs, since the matrix was built
    bool in_positions() const { return active && !slots_mode; }     // the matrix and the vectors of the moment are the position-space ones
    unsigned long long* dbg = nullptr;   // DPR_NJ_PHASES: [2 kernels][2048 blocks][8 stamps], then 32768 words of accumulate-mode statistics
    int64_t dbg_it = -1;
    int32_t* list = nullptr;         // units selected by the tests (sub-unit mask << 28 | strip << 18 | group)
    int32_t *blk_cb = nullptr, *blk_g0 = nullptr;   // test block -> (strip, first group)
    int nprep = 0;
};

// optional per-kernel timing of the pruned NJ loop (dpr_ctx_set_nj_kernel_timing): the run is enqueued eagerly and every
// stride-th iteration's launches are bracketed by HIP events on the library's stream
constexpr int kNjKernelsMax = 6;
struct NjKernelTiming {
    int stride = 0;                    // 0 = off
    int nk = 0;                        // kernels per iteration of the last timed run
    std::vector<hipEvent_t> ev;        // (nk + 1) events per sampled iteration
    double us_sum[kNjKernelsMax] = { 0, 0, 0, 0, 0, 0 };
    int64_t samples = 0;
};

// ---- one-exchange row-sharded streaming NJ (njs.hip) --------------------------------------------------------------
constexpr int kNjsMaxWorld = 64;
constexpr unsigned int kNjsTicketGroups = 32;                                   // two-level last-block ticket of njs_scan_kernel
constexpr size_t kNjsTicketBytes = 128 * (1 + (size_t)kNjsTicketGroups);       // one 128-byte line per counter
constexpr int kNjsLegacy = 0;    // round 2's loop: 4 launches + 2 all-gathers per iteration (nj.hip)
constexpr int kNjsPeer = 1;      // 2 launches + ONE RCCL all-gather (rank records); rows x / y pulled from their owners' memory
constexpr int kNjsMailbox = 2;   // 2 launches, no collective: the records go straight into every rank's mailbox
// A rank's record of one iteration as the one-exchange loop publishes it (one 64-byte line).  Besides the rank's best
// candidate it carries what lets every rank CHECK the others each iteration (round 4; the loop has no fences and had
// never met a second GPU):
//   seq     run id << 32 | it + 1 -- written last; a record of another iteration means the exchange did not happen;
//   ux      bits of the row sum U[x] of the node created by merge it - 1 as THIS rank computed it: a sum over values
//           derived from every element of the rows x and y the rank pulled from their owners.  The row sums are
//           replicated state, bit-identical on every rank by construction, so one differing word proves that some rank
//           pulled stale or torn data -- the run ends with DPR_ERR_COMM instead of a silently different merge log;
//   status  the publishing rank's state (3 / 4: it has already failed) -- the others fail the same way, not with NOCAND.
struct alignas(64) NjsRec {
    double q;
    uint64_t key;
    double d;
    uint64_t seq;
    uint64_t ux;
    uint64_t status;
    uint64_t pad[2];
};
// layout of a rank's peer-visible window (byte offsets): mail[2][kNjsMaxWorld] records (NjsRec) | barrier lines | slice | 4 row buffers
struct NjsLayout {
    int64_t off_bar = 0, off_slice = 0, off_rows = 0, bytes = 0;
    int64_t off_njr = 0;     // region of the row-sharded pruned NJ (njr.hip), 0 = none
    int64_t slice_len = 0;   // doubles (initial row sums of the own rows)
    int64_t ldv = 0;         // doubles per row buffer
};
struct NjPeer {
    int plan = kNjsLegacy;
    char* win = nullptr;             // this rank's window (fine-grained device memory)
    NjsLayout lay;
    unsigned int* ticket = nullptr;  // last-block tickets of the scan (kNjsTicketBytes)
    char** d_win = nullptr;          // device arrays [kNjsMaxWorld]: every rank's window / matrix as mapped into THIS process
    double** d_D = nullptr;
    std::vector<double*> h_D;        // host copy of d_D
    std::vector<char*> h_win;        // host copy of d_win
    std::vector<void*> opened;       // hipIpc mappings to close
    bool attached = false;
    unsigned long long bar_epoch = 0;
    unsigned long long run_id = 0;   // matrix builds on this window so far (part of every mail sequence number)
    unsigned long long poll_ticks = 200000000ull;    // 2 s of the 100 MHz wall clock
    int64_t fault_it = -1; int fault_rank = -1;      // test hook (dpr_ctx_set_debug_fault)
};

// ---- row-sharded pruned NJ (njr.hip): the position-space matrix of njp.hip with its rows dealt to the ranks ------------
// Chunk k of kNjrChunk positions (64 row groups: the rows of one test block of njp_post_kernel<64, 1>) belongs to rank
// k % world; a rank stores its chunks one behind the other at full width.  Units, their bounds, lists and scans belong to
// the owner of their rows.
constexpr int kNjrChunk = 1024;
constexpr int kNjrCollective = 1;        // block records and column slices travel by all-gathers (RCCL; device copies between virtual ranks)
constexpr int kNjrMailbox = 2;           // ... by stores straight into every rank's window: no collective
__host__ __device__ inline int njr_owner(int64_t p, int world) { return (int)((p / kNjrChunk) % world); }
__host__ __device__ inline int64_t njr_local_row(int64_t p, int world) { return (p / kNjrChunk / world) * kNjrChunk + p % kNjrChunk; }
__host__ __device__ inline int64_t njr_global_pos(int64_t l, int rank, int world) { return ((l / kNjrChunk) * world + rank) * kNjrChunk + l % kNjrChunk; }
// matrix rows a rank must be able to hold for P positions (whole chunks)
__host__ __device__ inline int64_t njr_rows_cap(int64_t P, int world)
{
    const int64_t chunks = (P + kNjrChunk - 1) / kNjrChunk;
    return ((chunks + world - 1) / world) * kNjrChunk;
}
// doubles per exchanged column slice: local rows of positions < N + 2 chunks (the kernels read vectors up to 511 positions
// behind P and an even index behind the last position)
__host__ __device__ inline int64_t njr_slice_len(int64_t N, int world) { return ((N / kNjrChunk + 2 + world - 1) / world) * kNjrChunk; }
// the njr region of a rank's peer-visible window (offsets relative to the region's start)
struct NjrLayout {
    int64_t off_recflag = 0, off_recs = 0, off_rowflag = 0, off_rows = 0, bytes = 0;
    int64_t rec_stride = 0;     // records per (parity, source rank): unit-scan blocks per rank + 1 header record
    int64_t slice = 0;          // doubles per column slice
};
struct NjRowShard {
    int world = 1, rank = 0, plan = 0;     // world > 1: active
    NjrLayout lay;
    int64_t win_off = 0;                   // start of the njr region inside the window (NjPeer::win)
    double* rows_plain = nullptr;          // collective plan: [world][2][slice] all-gather buffer (mailbox plan: the window's region)
    double* stage = nullptr;               // epoch builds: source rows pulled from their owners, [stage_rows][stage_ld]
    int64_t stage_rows = 0, stage_ld = 0;
    char** d_region = nullptr;             // device array [world]: the njr regions of all ranks' windows
    double** d_src = nullptr;              // device array [world]: source buffers of the epoch build in progress
    unsigned int* ticket = nullptr;        // last-block tickets of SCAN and EXTRACT (2 x kNjsTicketBytes)
    double* half[2] = { nullptr, nullptr };   // the two epoch buffers of this rank (halves of NjBuffers::D)
    std::vector<double*> peer_half[2];     // ... of every rank, as mapped into this process (epoch builds pull from them)
    int (*gather)(void* ctx, int kind, hipStream_t s) = nullptr;   // collective plan: kind 0 = block records (in place, b.partials), 1 = column slices (in place, rows_plain)
    int (*barrier)(void* ctx) = nullptr;   // all ranks' streams idle and in step (epoch builds)
    void* cb_ctx = nullptr;
    int64_t launches = 0, collectives = 0;
};

struct NjBuffers {
    NjPeer peer;
    NjKernelTiming* kt = nullptr;   // owned by the context
    double* D = nullptr;       // [rows_local_max][ld] (+ tail pad)
    // row-sharded pruned NJ: D is ONE allocation of two halves of twin_rows rows each (the epochs of njr.hip alternate between
    // them; one hipIpc handle covers both); D points at the first half, which also receives the tip-order rows
    int64_t twin_rows = 0;
    size_t half_bytes = 0;
    int64_t ld = 0;
    int64_t N = 0;             // total tips
    int64_t rows_local = 0;    // local rows at n = N
    double* U = nullptr;       // [N] replicated
    double* Ur = nullptr;      // [N] U/(n-2) for the current n
    uint64_t* KA = nullptr;    // [N] nj_key_a(i, n) for the current n
    NjRecord* partials = nullptr;  // [kScanBlocks]
    NjRecord* recs = nullptr;      // [world]
    NjsRec* recs64 = nullptr;      // [world] rank records of the one-exchange loop (njs.hip)
    double* xpart = nullptr;   // [ceil(N/256)]
    double* gath = nullptr;    // [3][world][slice] gathered column slices (world > 1)
    double* slice = nullptr;   // [3][slice] local column slices (world > 1)
    int64_t slice_len = 0;
    NjState* st = nullptr;
    int32_t* log_x = nullptr;  // [N]
    int32_t* log_y = nullptr;
    double* log_bx = nullptr;
    double* log_by = nullptr;
    int rank = 0, world = 1;
    NjPruned pr;
    NjRowShard rs;
};

// nj.hip
int nj_alloc(NjBuffers& b, int64_t N, int rank, int world, hipStream_t s, int64_t twin_rows = 0);   // fills ordered on s; same shape again: buffers kept
int nj_fill_pads(double* D, int64_t ld, int64_t nrows, int64_t ncols, int64_t rows_alloc, int64_t tail, bool diag, hipStream_t s);
void nj_free(NjBuffers& b);
int nj_expand_lower(NjBuffers& b, const double* d_packed_lower, hipStream_t s);
int nj_init_sums(NjBuffers& b, hipStream_t s, double* local_sums = nullptr);   // U (one rank) or the own rows' sums, state
int nj_prepare(NjBuffers& b, hipStream_t s);            // Ur, KA for n = st->n
int nj_launch_scan(NjBuffers& b, bool probe, int64_t n, int64_t it, hipStream_t s);
void nj_scan_config(int rg, int nt, int grid);  // tuning knobs (dpr_scan_tune)
int nj_scan_grid();
int nj_bw_probe(const double* buf, int64_t cap_bytes, double* sink, int64_t bytes, int nt, int grid, int reps, hipStream_t s, hipEvent_t e0,
                hipEvent_t e1, float* ms);
int nj_launch_post(NjBuffers& b, int64_t n, int64_t it, hipStream_t s);            // world == 1
int nj_launch_select_local(NjBuffers& b, int nparts, hipStream_t s);                            // -> b.recs[b.rank]
int nj_launch_commit_extract(NjBuffers& b, int64_t n, int64_t it, hipStream_t s);   // world > 1
int nj_launch_update_sharded(NjBuffers& b, int64_t n, hipStream_t s);               // world > 1
int nj_launch_unpack_u(NjBuffers& b, hipStream_t s);                                // world > 1: gathered row sums -> U
int nj_launch_finish(NjBuffers& b, int64_t n, int64_t it, hipStream_t s);           // materialise U[x] after the loop

// njs.hip: one-exchange row-sharded loop (world > 1)
NjsLayout njs_layout(int64_t N, int world, bool with_njr = false);
int njs_alloc_window(NjBuffers& b, hipStream_t s);
void njs_free_window(NjBuffers& b);
int njs_set_peers(NjBuffers& b, char* const* wins, double* const* Ds, hipStream_t s);
int njs_launch_scan(NjBuffers& b, int64_t n, int64_t it, bool pending, hipStream_t s);
int njs_launch_post(NjBuffers& b, int64_t n, int64_t it, bool pending, hipStream_t s);
int njs_launch_finish(NjBuffers& b, int64_t n, int64_t it, bool pending, hipStream_t s);
int njs_launch_barrier(NjBuffers& b, hipStream_t s);
int njs_launch_unpack_u(NjBuffers& b, hipStream_t s);

// njr.hip: row-sharded exact pruned NJ (world > 1; NjRowShard)
NjrLayout njr_layout(int64_t N, int world);
int njr_build(std::vector<NjBuffers*>& ranks, hipStream_t s);      // the ranks held by this process (all of them: virtual ranks; one: a process rank)
int njr_run(std::vector<NjBuffers*>& ranks, int64_t it0, int64_t todo, hipStream_t s);
void njr_free(NjBuffers& b);

// njp.hip: exact pruned NJ (world == 1)
int njp_build(NjBuffers& b, hipStream_t s);   // permute the tip-order matrix by ascending row sum into the pruned path's own buffer
void njp_free(NjPruned& q);                   // everything, the arena included
int njp_reserve(NjPruned& q, int64_t N, hipStream_t s);   // allocate the arena for N tips ahead of njp_build
void njp_reset(NjPruned& q);                  // epoch state only (graph, pointers); the arena stays for the next build
int njp_unit_owner(int64_t strip, int64_t group, int64_t P, int world);
int njp_run(NjBuffers& b, int64_t it0, int64_t todo, hipStream_t s);   // enqueue `todo` iterations (hipGraph replays)
const char* njp_kernel_name(int idx);   // intervals of one timed iteration of the LAST timed run, in launch order
void njp_set_kernel_names(const char* const* names);     // (up to kNjKernelsMax, "" terminated; names in () are not kernels)
int njp_phase_stamps(unsigned long long* out);   // debug (DPR_NJ_PHASES)
int njp_debug_list(NjBuffers& b, int32_t* out, int64_t cap, int64_t* count, int64_t* P, double* ur, int64_t urcap);   // debug
const double* njp_current_u(const NjPruned& q, int64_t it);
int njp_shape(const NjPruned& q, int64_t* positions, int* row_groups, int* strips, int* post2, int* scan_grid);   // launch shape of the current epoch   // row sums by position after `it` iterations

// Divide-and-conquer cluster distances (dc.hip builds the jobs; msa.hip / mash.hip run them).
// Cluster ci has cl_m[ci] members (tip ids members[cl_moff[ci] + t], ascending) and a leaf list
// cols[cl_coff[ci] + u], u < kDcLeaves + m: the 2 x 5 closest leaves of the cluster edge and of its
// reverse (-1 = empty list entry), then the members.  Element (t, u), u < kDcLeaves + t, of its
// distance block is out[cl_out[ci] + t * cl_ld[ci] + u] = distance(row tip members[t], column tip cols[u]).
constexpr int kDcLeaves = 10;
struct PairJobs {
    const int4* jobs;          // (cluster, first row t0, first column u0, -)
    const int32_t* members;
    const int32_t* cols;
    const int64_t* cl_moff;
    const int32_t* cl_m;
    const int64_t* cl_coff;
    const int64_t* cl_out;
    const int32_t* cl_ld;
    double* out;
};

// msa.hip
struct MsaBuffers {
    uint32_t* planes = nullptr;  // [4][n][W32]: X (not a base), LO, HI, LX = LO | X bit planes, 32 bases per word
    int64_t n = 0, L = 0, W32 = 0;
    // alignments of at most kMsaTabSites sites: the distance of every (useful, match) pair of counts, types 1 and 2 -- the same
    // function of the same two integers as the epilogue computes, read instead of recomputed (a division and a log per pair are
    // half of the pair kernel's instructions at 400 sites)
    double* jc_tab = nullptr;    // [2][(L + 1) * (L + 1)]; longer alignments: the band [2][kMsaBand + 1][L + 1] (useful = L - g)
    // per sequence: bit j = its 16-word stage j holds a not-a-base position (gap, N, padding); stages from 63 on share bit 63
    unsigned long long* xstage = nullptr;
};
constexpr int64_t kMsaTabSites = 1024;
constexpr int kMsaBand = 15;
int msa_upload(MsaBuffers& m, const uint64_t* packed4, int64_t n, int64_t L, hipStream_t s);
void msa_free(MsaBuffers& m);
int msa_dist_rows(const MsaBuffers& m, NjBuffers& b, int dist_type, hipStream_t s);
int msa_dist_tile_edge(int dist_type);   // rows/columns per job tile of msa_dist_jobs
int msa_counts_row(const MsaBuffers& m, int64_t row, int32_t* d_useful, int32_t* d_match, hipStream_t s);

// mash.hip
// inverted index over the sketches (mash_index.hip): per chunk of 512 tips the (value -> tips, positions) postings
struct MashIndex {
    uint32_t* post = nullptr;    // [n*S] per chunk sorted by value: 2 * (tip mod 512) << 16 | position; tip 511 + position 65535 = not a first copy
    uint64_t* uniq = nullptr;    // [nu] distinct values, chunk after chunk
    uint32_t* off = nullptr;     // [nu + 1] first posting of each distinct value
    uint32_t* bkt = nullptr;     // [chunks][65537] directory on the leading bits: first distinct value of each bucket
    uint32_t* ubase = nullptr;   // [chunks + 1] first distinct value of each chunk
    int32_t* shift = nullptr;    // [chunks] bucket = value >> shift
    uint64_t* vmax = nullptr;    // [chunks] largest value
    uint16_t* mult = nullptr;    // [n*S] copies of the value at its first position in a sketch, 0 elsewhere
    double* dtab = nullptr;      // [S + 1] distance of every count
    int32_t* dblk = nullptr;     // [nu] dense block of the value or -1
    uint16_t* dense = nullptr;   // [ndense][512] position of the value in every tip of the chunk, 65535 = absent
    int64_t chunks = 0;
    uint32_t nu = 0, ndense = 0;
};

struct MashBuffers {
    uint64_t* packed2 = nullptr;   // flat 2-bit packed reads
    uint64_t* word_off = nullptr;  // [n]
    uint64_t* len = nullptr;       // [n] bases
    uint64_t* sketches = nullptr;  // [n][S] ascending
    // run encoding of every sketch against ONE reference list (mash_encode, see mash.hip): tokens [n][S] of 16 bytes,
    // tok_cnt[n]; ref = the reference's distinct values (ref_n of them)
    uint4* tokens = nullptr;
    int32_t* tok_cnt = nullptr;
    uint64_t* ref = nullptr;
    int ref_n = 0;
    double tok_mean = 0.0;         // tokens per sketch (sampled): the token kernel is used while this stays small
    MashIndex index;               // built when the sketches do not resemble each other (or DPR_MASH_INDEX=1)
    bool share_chip = false;       // the next distance launches run beside tree kernels of another stream: leave wave slots free
    uint64_t total_words = 0;
    int64_t n = 0;
    int S = 0, k = 0;
};
int mash_upload(MashBuffers& m, const uint64_t* packed2, const uint64_t* word_off, const uint64_t* len,
                int64_t n, hipStream_t s);
void mash_free(MashBuffers& m);
int mash_sketch(MashBuffers& m, int k, int S, hipStream_t s);
// rows r0..r0+nr (world > 0: owned local rows of (rank,world); world == 0: plain tip ids) x columns
// [0,ncols) -> out[t*ld + j]; full = also j >= i
int mash_dist_rows(const MashBuffers& m, int64_t r0, int64_t nr, int rank, int world, bool full,
                   int64_t ncols, double* out, int64_t ld, hipStream_t s, bool transposed = false);
int mash_dist_jobs(const MashBuffers& m, const PairJobs& J, int njobs, hipStream_t s);
// mash_index.hip
int mash_index_build(MashBuffers& m, hipStream_t s);
void mash_index_free(MashIndex& ix);
int mash_dist_index(const MashBuffers& m, int64_t r0, int64_t nr, int64_t ncols, double* out, int64_t ld, double* mir,
                    bool transposed, hipStream_t s, int rank, int world);
// the whole Mash matrix of a rank whose rows are sharded by row blocks (world > 1): D_local holds its rows at full width
int mash_dist_matrix_sharded(const MashBuffers& m, int rank, int world, int64_t rows_local, double* D_local, int64_t ld, hipStream_t s);
int mash_jobs_rows();   // members per job
int mash_jobs_cols();   // leaf-list positions per job
int mash_hash_positions(const MashBuffers& m, int64_t seq, int k, uint64_t* d_out, uint64_t len, uint64_t word_off,
                        hipStream_t s);

// msa.hip: same row-provider shape as mash_dist_rows
// transposed: out[j * ld + t] instead of out[t * ld + j]
int msa_dist_block_rows(const MsaBuffers& m, int64_t r0, int64_t nr, int rank, int world, int64_t ncols,
                        int dist_type, double* out, int64_t ld, hipStream_t s, bool transposed = false);
int msa_dist_jobs(const MsaBuffers& m, int dist_type, const PairJobs& J, int njobs, hipStream_t s);

// place.hip
struct PlaceBuffers {
    int64_t N = 0;   // tips the arrays are sized for; internal node ids start at N
    int64_t M = 0;   // tips of the edge scan's slot range (4M-4 slots): N, or the backbone size in DC mode
    int32_t *head = nullptr, *e = nullptr, *nxt = nullptr, *belong = nullptr, *rev = nullptr, *cid = nullptr;
    int32_t* cont = nullptr;     // [16N] per slot u->v: the (up to two) slots leaving v other than v->u (-1 none; -2: v has degree > 3, walk its list)
    double *len = nullptr, *cdis = nullptr;
    int32_t *q_id = nullptr, *q_from = nullptr;
    double* q_dis = nullptr;
    void* partials = nullptr;
    int nparts_max = 0;
    void* partials_multi = nullptr;   // block minima of a scan launch that serves several tips (place_tips)
    int64_t nparts_multi = 0;
    int dbg = 0;   // 4: place_update_kernel writes its phase clocks into the trace (DPR_PLACE_CLOCKS, profiling only)
    // Edge records (round 5): a read-optimised MIRROR of what the per-tip scan (calculateBranchLength, src/placement_close_k.cu:
    // 309-358) needs, one entry per UNDIRECTED edge = the slot pair s / rev[s]; side 0 is the slot with belong >= e (the one the
    // reference evaluates), side 1 its reverse.  Struct of arrays (every load of the scan coalesced); the slot-indexed arrays
    // above stay the authoritative state -- the split and the closest-list BFS write both.
    double* er_d = nullptr;      // [11][ecap]: cdis of side 0 (fields 0..4), of side 1 (5..9), len (10)
    int32_t* er_i = nullptr;     // [12][ecap]: cid of side 0 (0..4), of side 1 (5..9), slot of side 0 (10), slot of side 1 (11)
    int32_t* eidx = nullptr;     // [8N] slot -> 2 * edge + side
    int64_t ecap = 0;
    int32_t* bfs_cnt = nullptr;  // [N] per placed tip: slots its closest-list walk reached beyond the two rounds the split applies (negative: -(n + 1), the walk of a node of degree > 3)
    int32_t* misc = nullptr;     // [0]: smallest live slot with belong < e (its tuple (0, 0, 2) competes for the first minimum); [1]: edge counter of an import
};
int place_alloc(PlaceBuffers& p, int64_t N, int64_t M = 0);   // M = 0: M = N
void place_free(PlaceBuffers& p);
int place_init_fresh(PlaceBuffers& p, hipStream_t s);
int place_initial_tree(PlaceBuffers& p, const double* d_dis_row1, hipStream_t s);
int place_import_backbone(PlaceBuffers& p, int64_t m, hipStream_t s);
int place_tip(PlaceBuffers& p, const double* d_dis, int64_t tip, double* d_trace, hipStream_t s);
int place_tips(PlaceBuffers& p, const double* d_dis0, int64_t ldb, int64_t tip0, int64_t count, double* d_trace, hipStream_t s);
int64_t place_multi_min();        // first tip of the four-tip launch pairs

// exact.hip: exact placement mode (src/placement.cu)
struct XStep {                   // counters of the two node lists, status words (device)
    int nroot[2], ntop[2];       // [parity of the rank buffers the lists belong to]: px_patch_kernel of tip i fills [(i + 1) & 1] and clears [i & 1]
    int poll_fail, poll_node, poll_pass;   // px_top_poll: a value never arrived (node on record, pass 0 bottom-up / 1 top-down): the run fails
    int quirk;                   // the reference's swap in updateTreeStructure was taken: its depths are no tree depths any more (see exact.hip)
};
struct ExactBuffers {
    double* lim = nullptr;       // [8N] per directed slot
    int32_t* dep = nullptr;      // [2N] depth below node N
    int32_t* rk[2] = { nullptr, nullptr };   // [2N] DFS pre-order rank, double-buffered: placing tip i reads rk[i & 1] and writes rk[(i + 1) & 1]
    int32_t* sz[2] = { nullptr, nullptr };   // [2N] nodes in the subtree, same buffering
    int32_t* nar = nullptr;      // [2N] node at rank
    int32_t* tix = nullptr;      // [2N] top node -> index in the level-sorted top list (-1: not a top node)
    int32_t* roots = nullptr;    // [2N] x int4: roots of the small subtrees (<= sm nodes, parent's subtree larger): node, rank, size, depth
    int32_t* tops = nullptr;     // nodes whose subtree has more than 64 nodes
    int32_t* order = nullptr;    // [2N] top nodes grouped by depth
    int32_t* lvoff = nullptr;    // [2N+2] first index of every level in order[]
    int32_t* hist = nullptr;     // [2N+2]
    int32_t* nd = nullptr;       // [2N][12] node records: slot[3], reverse slot[3], target node[3], pad
    XStep* st = nullptr;
    void* tpack = nullptr;       // [kTopClimb] structural records of the top nodes, packed beside the small-subtree pass (px_top_climb)
    void* partials = nullptr;
    int32_t* dfsrk = nullptr;    // literal schedule: the reference's single rank array (alias of rk[0])
    bool literal = false;        // run the literal one-workgroup schedule (fallback / DPR_EXACT_LITERAL=1)
    unsigned long long* clk = nullptr;   // DPR_EXACT_CLOCKS=1 (profiling): phase clocks of px_top_kernel, summed over the tips (100 MHz ticks)
    int sm = 64;                 // nodes of a small subtree for the next patch launch (64 / 256 / 1 024: exact_adapt)
    int sm_forced = 0;           // tests (DPR_EXACT_SM=64|256|1024): no adaptation
    int pass_sm = 64;            // ... of the passes that are enqueued last (their workgroups' partials are what the next patch reads)
    bool top_poll = false;       // tests / A-B (DPR_EXACT_TOP_POLL=1): the polling schedule of px_top_poll even where the climbing one applies
    bool top_levels = false;     // tests / A-B (DPR_EXACT_TOP_LEVELS=1): the top-tree pass level by level with a workgroup barrier per level (rounds 3-5)
    bool top_in_memory = false;  // tests (DPR_EXACT_TOP_MEM=1): the top-tree pass keeps its values in memory even when they fit LDS
};
int exact_alloc(ExactBuffers& x, int64_t N);
void exact_free(ExactBuffers& x);
int exact_init(PlaceBuffers& p, ExactBuffers& x, const double* d_dis_row1, const double* d_dis_row2, bool has_tip2,
               hipStream_t s);
int exact_tip(PlaceBuffers& p, ExactBuffers& x, int64_t tip, const double* d_dis_next, bool has_next, double* d_trace,
              hipStream_t s);
int exact_adapt(ExactBuffers& x, hipStream_t s);                 // between batches of tips: larger small subtrees when the top tree approaches its LDS capacity
int exact_quirk(ExactBuffers& x, hipStream_t s, bool* quirk);    // did the fast schedule meet the case only the literal one reproduces?

// dc.hip: divide-and-conquer mode (cluster assignment + concurrent cluster trees)
struct DcTable {
    int nv = 0;                    // eligible backbone slots (belong >= e), ascending
    int32_t* vslots = nullptr;
    int32_t* et_cid = nullptr;     // [nv][10] closest ids: own list, reverse list
    double* et_cdis = nullptr;     // [nv][10]
    double* et_len = nullptr;      // [nv]
    // chunks of consecutive table entries (tree order) whose closest lists name at most kDcRows distinct backbone tips: the assignment
    // scan stages those tips' distance rows in LDS once per (chunk, 64 queries)
    int nch = 0;
    int32_t* ch_e0 = nullptr;      // [nch + 1] first table entry of a chunk
    int32_t* ch_l0 = nullptr;      // [nch + 1] first entry of a chunk in ch_leaf
    int32_t* ch_leaf = nullptr;    // the chunks' distinct backbone tips
    uint4* et_rec = nullptr;       // [nv][11] x 16 B per entry: for each of the 10 list entries {LDS offset (in doubles) of its staged row (absent: the
                                   // -inf row), 0, path length}; then {slot, 0, edge length} -- copied to LDS per chunk, read there at a wave-uniform address
    double* part_add = nullptr;    // [chunks][ldq] per-chunk first minima
    int32_t* part_pos = nullptr;
    size_t part_cap = 0;
};
struct DcStats { int64_t clusters = 0, max_cluster = 0, pairs = 0, groups = 0, jobs = 0; };
int dc_table_build(PlaceBuffers& p, int64_t B, DcTable& t, hipStream_t s);
void dc_table_free(DcTable& t);
// dT[c * ldq + q] = distance(query q of the batch, backbone tip c); d_cluster_id[q] = chosen slot
int dc_assign(DcTable& t, const double* dT, int64_t ldq, int Q, int32_t* d_cluster_id, hipStream_t s);
int dc_cluster_phase(PlaceBuffers& p, const int32_t* h_cluster_id, int64_t N, int64_t B, int source, int dist_type,
                     const MsaBuffers* msa, const MashBuffers* mash, double* d_trace, size_t budget_bytes,
                     DcStats* stats, int rank, int world, hipStream_t s);
// multi-rank merge of the cluster phase: every array element is changed by at most one rank, so
// new = old + sum over ranks of (new - old) in wrap-around integer arithmetic (doubles as their bit patterns)
void dc_deal_clusters(const int64_t* sizes_desc, int64_t count, int world, int32_t* owner);
void dc_query_share(int64_t n, int64_t B, int rank, int world, int64_t* q0, int64_t* q1);
int dc_delta_sub(void* cur, const void* old, int64_t words64, hipStream_t s);   // cur -= old (64-bit words)
int dc_delta_add(void* cur, const void* old, int64_t words64, hipStream_t s);   // cur += old

}  // namespace dpr
