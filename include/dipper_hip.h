/*
 * dipper_hip.h -- C ABI of the MI355X (gfx950) hot path of the `dipper` phylogeny engine.
 *
 * The reference (TurakhiaLab/DIPPER) has no FFI layer: its hot path is reached through the
 * methods of the structs declared in src/mash_placement.cuh.  Each entry point below names the
 * reference interface it replaces (file:line under /root/reference).  The reference computes one
 * matrix row per kernel launch and crosses the host/device boundary every row and every NJ
 * iteration; this ABI is deliberately coarser (whole matrix / whole run per call).
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success, <0 on error; the message of the
 *     last error of the calling thread is returned by dpr_last_error().
 *   - the caller owns every host buffer; the library owns all device memory of a dpr_ctx.
 *   - one dpr_ctx drives ONE GPU from ONE host thread at a time.  Multi-GPU = one process (and one
 *     ctx) per GPU, joined by dpr_comm_init(); the N x N matrix is then sharded by rows
 *     (block-cyclic, DPR_ROW_BLOCK rows).  The row-sharded NJ loop has three exchange plans
 *     (dpr_ctx_set_nj_exchange): legacy = two RCCL all-gathers per iteration (one record, three column
 *     slices; the default until the others have been validated on a multi-GPU node), peer = ONE
 *     all-gather of the rank records, rows x / y pulled from their owners' memory, mailbox = no
 *     collective at all (records stored straight into every rank's mailbox over xGMI).
 *   - nothing here falls back to the CPU: without a usable gfx950 device dpr_create() fails.
 */
#ifndef DIPPER_HIP_H
#define DIPPER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define DPR_ABI_VERSION 1
#define DPR_ROW_BLOCK 64 /* rows per ownership block of the sharded matrix */

/* distance sources for dpr_dist_matrix / dpr_place_run */
#define DPR_SRC_MSA 1    /* 4-bit packed aligned sequences, MSADeviceArrays  */
#define DPR_SRC_MASH 2   /* bottom-S sketches,              MashDeviceArrays */
#define DPR_SRC_MATRIX 3 /* PHYLIP lower triangle,          MatrixReader     */

/* distance types of `-d` (src/MSA.cu:81-86) */
#define DPR_DIST_UNCORRECTED 1
#define DPR_DIST_JC 2
#define DPR_DIST_TAJIMANEI 3
#define DPR_DIST_K2P 4
#define DPR_DIST_TAMURA 5
#define DPR_DIST_JINNEI 6

/* error codes */
#define DPR_OK 0
#define DPR_ERR_ARG -1
#define DPR_ERR_HIP -2     /* "Gpu_ERROR: ..." in the reference (e.g. src/neighborJoining.cu:44-55) */
#define DPR_ERR_STATE -3   /* call order violated */
#define DPR_ERR_NOCAND -4  /* no Q candidate below the reference's init value 10000 */
#define DPR_ERR_COMM -5

typedef struct dpr_ctx dpr_ctx;

const char *dpr_last_error(void);
int dpr_abi_version(void);

/* ---- host-only encoders (no GPU needed) ------------------------------------------------------
 * replace fourBitCompressor (src/fourBitCompressor.cpp:5-41) and twoBitCompressor
 * (src/twoBitCompressor.cpp:5-41): out has ceil(len/16) resp. ceil(len/32) words. */
int dpr_pack4(const char *seq, uint64_t len, uint64_t *out);
int dpr_pack2(const char *seq, uint64_t len, uint64_t *out);

/* ---- host-only sharding helpers (pure functions; used by the N>1 host logic and its CPU tests) */
/* ... and of the ROW-SHARDED EXACT PRUNED NJ (njr.hip; several ranks, dpr_ctx_set_nj_multi_plan(ctx, 3) / DPR_NJ_MULTI=rows, or
 * chosen by itself once two copies of the matrix no longer fit one GPU): north_star's row-block split of the N x N matrix
 * (src/neighborJoining.cu:117-148) under the pruned algorithm.  The position-space matrix (positions = nodes sorted by row sum,
 * per epoch) is dealt in chunks of DPR_NJR_CHUNK positions: chunk k belongs to rank k % world and is that rank's (k / world)-th
 * chunk; the 16 x 512 units, their bounds, tests and scans belong to the owner of their rows. */
#define DPR_NJR_CHUNK 1024
int dpr_njr_owner(int64_t position, int world);              /* rank owning the matrix row of `position`           */
int64_t dpr_njr_local_row(int64_t position, int world);      /* its index in the owner's storage                   */
int64_t dpr_njr_global_pos(int64_t local_row, int rank, int world);   /* inverse                                  */
int64_t dpr_njr_rows_cap(int64_t positions, int world);      /* matrix rows a rank must be able to hold (whole chunks) */
int dpr_shard_owner(int64_t row, int world);                 /* rank owning matrix slot `row`      */
int64_t dpr_shard_local_row(int64_t row, int world);         /* its index in the owner's storage  */
int64_t dpr_shard_rows(int64_t n, int rank, int world);      /* #slots < n owned by `rank`         */
int64_t dpr_shard_global_row(int64_t local, int rank, int world);
/* lexicographic (q, key) minimum over `count` 32-byte records {double q; uint64 key; double d; pad};
 * returns the index of the winner or -1 when no record has key != UINT64_MAX. */
int dpr_record_reduce(const void *records, int count);
/* tie-break key of ordered pair (i,j) at active size n (src/neighborJoining.cu:124-146,214) */
uint64_t dpr_nj_key(int64_t i, int64_t j, int64_t n);

/* ---- context ---------------------------------------------------------------------------------
 * replaces cudaSetDevice(1) (src/tree_generation.cu:240-245; the hard-coded device index is a
 * reference quirk). */
int dpr_create(dpr_ctx **out, int device);
int dpr_destroy(dpr_ctx *ctx);
/* Validation mode of the multi-GPU path: ONE context holds `world` virtual ranks on one device and
 * runs the sharded algorithm (same kernels, same per-rank buffers) with the two per-iteration
 * all-gathers done as device copies instead of RCCL.  Used to prove on a 1-GPU box that the sharded
 * result is bit-identical to the single-rank result. */
int dpr_create_virtual(dpr_ctx **out, int device, int world);
int dpr_device_name(dpr_ctx *ctx, char *buf, int cap);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI).  No reference counterpart (single GPU).
 * rank 0 calls dpr_comm_unique_id, the 128 bytes are broadcast by the launcher (torch.distributed
 * in bench.py), every rank calls dpr_comm_init before any dpr_set_* call. */
int dpr_comm_unique_id(void *out128);
int dpr_comm_init(dpr_ctx *ctx, int rank, int world, const void *id128);
/* 1-rank RCCL round trip on this context's GPU (plumbing check on a single-GPU box) */
int dpr_comm_selftest(dpr_ctx *ctx);
/* Ranks without RCCL: processes whose devices (or ONE shared device -- which RCCL refuses) can map each other's memory.
 * The row-sharded NJ then runs its mailbox plan.  dpr_peer_export allocates this rank's NJ buffers + peer window for
 * n_tips and writes a 192-byte description; the launcher hands every rank all descriptions in rank order
 * (dpr_peer_attach) before dpr_dist_matrix.  With RCCL (dpr_comm_init) the library exchanges them itself. */
int dpr_comm_init_local(dpr_ctx *ctx, int rank, int world);
int dpr_peer_export(dpr_ctx *ctx, int64_t n_tips, void *out192);
int dpr_peer_attach(dpr_ctx *ctx, const void *all192);
/* rank and rank count AS THE COMMUNICATOR REPORTS THEM (ncclCommUserRank / ncclCommCount; for ranks on the window transport of
 * dpr_comm_init_shared: this rank and the number of ranks that have joined the shared region); 0 / 1 without one */
int dpr_comm_info(dpr_ctx *ctx, int *rank, int *nranks);
/* Ranks joined through a SHARED HOST REGION -- what `dipper --gpus G` does (dipper_amd/host/main.cpp forks its ranks around one
 * anonymous shared mapping before any GPU call), and what any launcher can do that hands the G processes of one node the same
 * DPR_COMM_SHARED_BYTES of zero-initialised shared memory (shm_open + mmap).  No reference counterpart: the reference drives one
 * device (src/tree_generation.cu:240-245).  Call after dpr_create, before any dpr_set_* call, on every rank (collective).
 * transport 1 = RCCL: rank 0's ncclUniqueId travels through the region, dpr_comm_init follows -- one rank per GPU;
 * transport 2 = ipc: a device window per rank (DPR_COMM_WINDOW_MB, default 64), mapped by the other ranks through hipIpc; all-gathers
 *   and integer all-reduces run through the windows in chunks between host barriers (synchronous with the host).  The ONLY
 *   transport for ranks that share a device (RCCL refuses them), i.e. for rehearsing every multi-rank path on a single GPU;
 * transport 0 = auto: ipc iff two ranks name the same device (PCI bus id), else RCCL.
 * The region also carries a failure word: a rank that gives up (or the launcher, dpr_shared_abort, when a rank has died) sets it
 * and every wait of every rank ends with DPR_ERR_COMM instead of hanging; DPR_COMM_TIMEOUT_MS (default 600 000) bounds a wait. */
#define DPR_COMM_SHARED_BYTES 65536
int dpr_comm_init_shared(dpr_ctx *ctx, int rank, int world, void *shared, uint64_t bytes, int transport);
/* *transport: 0 = none (one rank / virtual ranks), 1 = RCCL, 2 = device windows over hipIpc, 3 = peers attached by the launcher
 * (dpr_comm_init_local: mailbox NJ only); *collectives: device all-gathers + all-reduces this context has taken part in */
int dpr_comm_stats(dpr_ctx *ctx, int *transport, int64_t *collectives);
/* host-only pieces of that protocol (no GPU needed): the launcher's failure path, and the barrier / 512-byte-per-rank gather the
 * library runs over the region (CPU tests drive them with several processes; *sense is a rank's private word, 0 at the start) */
int dpr_shared_abort(void *shared);
int dpr_shared_failed(const void *shared);
int dpr_shared_barrier(void *shared, int world, uint32_t *sense, int timeout_ms);
int dpr_shared_gather(void *shared, int rank, int world, uint32_t *sense, int timeout_ms, const void *mine, void *all, int bytes);

/* ---- inputs ----------------------------------------------------------------------------------*/
/* MSADeviceArrays::allocateDeviceArrays (src/MSA.cu:14-72): packed4 is [n][ceil(L/16)] words as
 * produced by dpr_pack4; L = length of sequence 0 (src/MSA.cu:19). */
int dpr_set_msa(dpr_ctx *ctx, const uint64_t *packed4, int64_t n, int64_t L);
/* MashDeviceArrays::allocateDeviceArrays (src/mash.cu:14-122): packed2 flat, word_off[i] = first
 * word of sequence i (exclusive scan of ceil(len/32), src/mash.cu:109-119), len[i] in bases. */
int dpr_set_reads(dpr_ctx *ctx, const uint64_t *packed2, const uint64_t *word_off,
                  const uint64_t *len, int64_t n);
/* MatrixReader (src/matrix_reader.cu:15-45): rows concatenated, row i has i entries (i=0..n-1),
 * already parsed with the reference's float rounding by the host reader.  The device copy stays until the next
 * dpr_set_matrix_lower / dpr_destroy, so dpr_dist_matrix(DPR_SRC_MATRIX) and dpr_place_run may be called repeatedly. */
int dpr_set_matrix_lower(dpr_ctx *ctx, const double *rows, int64_t n);

/* ---- Mash sketches: MashDeviceArrays::sketchConstructionOnGpu (src/mash.cu:386-424).
 * S must be 1000 unless the reference quirk of SURVEY 9.8 is lifted; host_sketches (optional)
 * receives [n][S] ascending hashes. */
int dpr_sketch(dpr_ctx *ctx, int k, int S, uint64_t *host_sketches);

/* ---- distance matrix: NJDeviceArrays::getDismatrix (src/neighborJoining.cu:35-85) with the row
 * providers MSADeviceArrays/MashDeviceArrays/MatrixReader::distConstructionOnGpu
 * (src/MSA.cu:271-282, src/mash.cu:457-471, src/matrix_reader.cu:23-45), fillDismatrix (:20-32)
 * and calculateU (:94-115).  Builds the (sharded) symmetric fp64 matrix and the row sums U. */
int dpr_dist_matrix(dpr_ctx *ctx, int source, int dist_type, int k);

/* Pay the one-time costs of the process's first hipGraph instantiation (~30 ms; dpr_nj_run replays graphs) and of its first
 * staged device-to-host copy (~8 ms; the first epoch rebuild of an NJ run reads the row sums back) now, on a private
 * stream -- the CLI calls it from a helper thread while it reads its input.  No reference counterpart.
 * Threading: this is the ONE entry point that may run concurrently with other dpr_* calls on the same context (it
 * touches nothing of the context but its device index; thread-local stream capture, thread-local error string);
 * join the helper thread before dpr_destroy. */
int dpr_warm_graphs(dpr_ctx *ctx);

/* Optional: allocate the matrix buffers of a following dpr_dist_matrix over n tips now (cudaMalloc of the n x n matrix
 * in NJDeviceArrays::getDismatrix, src/neighborJoining.cu:41-56); dpr_dist_matrix then finds them in place.  Lets the
 * CLI overlap the allocation with the packing of the input. */
int dpr_reserve_nj(dpr_ctx *ctx, int64_t n);

/* ---- neighbor joining: NJDeviceArrays::findNeighbourJoiningTree (src/neighborJoining.cu:197-249)
 * without the Newick print.  Outputs (host, N-2 entries each): merged matrix slots x<y and the two
 * branch lengths per iteration; *last_d = D[0][1] of the final pair.  max_iters<0 runs all N-2
 * iterations.  The merge log fully determines the tree (realID bookkeeping :233-237 is host work).
 * Returns the number of iterations done (>=0) or an error code. */
int64_t dpr_nj_run(dpr_ctx *ctx, int64_t max_iters, int32_t *merge_x, int32_t *merge_y,
                   double *bl_x, double *bl_y, double *last_d);

/* ---- single Q-argmin at the CURRENT active size (findMinDist + thrust::min_element,
 * src/neighborJoining.cu:117-148,214).  Returns the reference's winning ordered tuple (i,j,q).
 * reps>1 repeats the scan kernel `reps` times (kernel symbol dpr::nj_scan_kernel<true,16,true,false>:
 * the production streaming scan; the first template flag only tags probe launches in profiles,
 * the code is the same) and reports the average duration measured with HIP events on the library's stream;
 * used for the roofline. */
int dpr_argmin_once(dpr_ctx *ctx, int reps, int32_t *out_i, int32_t *out_j, double *out_q,
                    float *out_ms_per_scan);

/* NJ algorithm on a single GPU: 1 = exact pruned scan (default), 0 = full streaming scan every
 * iteration.  Both produce the same merge log bit for bit (tests run both); takes effect at the next
 * dpr_dist_matrix.  Environment DPR_NJ_MODE=stream selects 0 for the CLI. */
int dpr_set_nj_mode(int mode);
/* Several GPUs + pruned scan: every rank keeps the whole matrix and the ranks share the per-iteration unit tests
 * and scans (a unit belongs to one rank for good; one small all-gather of block records per iteration).
 * dpr_set_nj_virtual_shards(w) makes the next dpr_dist_matrix of a single-rank context emulate w such ranks
 * (validation on one GPU; same merge log bit for bit). */
int dpr_set_nj_virtual_shards(int w);
/* Several ranks (dpr_comm_init), pruned NJ: plan 0 = auto (unit-sharded scans + one all-gather per iteration from
 * 65 536 tips on; below that every rank runs the single-GPU plan on its own copy of the matrix -- an iteration is then
 * ~20 us of dependent latency, which a collective per iteration would only lengthen), 1 = always unit-sharded,
 * 2 = never.  Also DPR_NJ_MULTI=auto|shard|solo.  No counterpart in the reference (single GPU,
 * src/tree_generation.cu:240-245).  dpr_nj_is_unit_sharded: what the last dpr_dist_matrix chose. */
int dpr_set_nj_multi_plan(int plan);
int dpr_nj_is_unit_sharded(dpr_ctx *ctx);
/* the multi-rank NJ plan the last dpr_dist_matrix set up, in words ("single rank", "pruned, every rank runs ... (replicas)",
 * "pruned, ... unit tests and scans sharded ...", "streaming, rows sharded ..., exchange ...", "row-sharded pruned NJ ...") */
int dpr_get_nj_multi_info(dpr_ctx *ctx, char *buf, int cap);
/* The three plan knobs above are process-wide DEFAULTS.  Per context (two contexts of one process may run different
 * plans): same meaning, value -1 = follow the process-wide default again; effective at that context's next
 * dpr_dist_matrix. */
int dpr_ctx_set_nj_mode(dpr_ctx *ctx, int mode);
int dpr_ctx_set_nj_multi_plan(dpr_ctx *ctx, int plan);
/* Adaptive plan of the single-rank NJ (default on; DPR_NJ_ADAPTIVE=0 switches it off): the exact pruned scan while its
 * bounds prune; once more than 70 % of an epoch's units are listed per iteration (tie-heavy / arbitrary `-i d` matrices,
 * src/matrix_reader.cu:23-45 feeds NJ anything) the run is handed over to the streaming loop of src/neighborJoining.cu:
 * 117-148,211-243 (dense slot-space matrix, one full Q scan per iteration); pruned epochs are probed again later with a
 * back-off.  Same merge log either way. */
int dpr_ctx_set_nj_adaptive(dpr_ctx *ctx, int on);
int dpr_get_nj_adaptive_stats(dpr_ctx *ctx, int64_t *stream_iterations, int64_t *stream_epochs);
/* Exchange plan of the ROW-SHARDED streaming NJ loop (several ranks, DPR_NJ_MODE=stream / dpr_ctx_set_nj_mode(ctx, 0);
 * replaces src/neighborJoining.cu:211-243): 0 = legacy (4 launches + 2 all-gathers per iteration; the DEFAULT since round 4:
 * the other two have only run on one device so far), 1 = peer (2 launches + ONE all-gather of the rank records; rows x / y are
 * pulled from their owners' memory), 2 = mailbox (2 launches, no collective: the records go straight into every rank's
 * mailbox); -1 = DPR_NJ_EXCHANGE (legacy | peer | mailbox) / default.
 * A plan that cannot be set up on every rank falls back to 0 on all ranks together (note in dpr_get_nj_exchange_info).
 * Plans 1 and 2 check themselves every iteration: each rank's record carries the bits of the row sum it derived from the rows
 * it pulled (replicated state: identical on every rank by construction) and its status; a differing word, a record of another
 * iteration or a poll that times out ends dpr_nj_run with DPR_ERR_COMM on every rank. */
int dpr_ctx_set_nj_exchange(dpr_ctx *ctx, int plan);
/* test hook of that cross-check (no counterpart in the reference): rank `rank` uses a wrong value for one element of a row it
 * pulled at iteration `iteration`; (-1, -1) = off.  Effective for the context's following dpr_nj_run calls. */
int dpr_ctx_set_debug_fault(dpr_ctx *ctx, int64_t iteration, int rank);
/* what the last dpr_dist_matrix set up (*active_plan, note) and what the last dpr_nj_run enqueued on this rank */
int dpr_get_nj_exchange_info(dpr_ctx *ctx, int *active_plan, int64_t *launches, int64_t *collectives, char *note, int cap);
/* bound of one mailbox / barrier poll in ms (default 2000): a rank whose record does not arrive ends the run with DPR_ERR_COMM */
int dpr_ctx_set_poll_limit_ms(dpr_ctx *ctx, int ms);
int dpr_ctx_set_nj_virtual_shards(dpr_ctx *ctx, int w);
/* Measurement aid of the pruned NJ loop (bench.py's `timed_kernels` record): stride > 0 makes the following dpr_nj_run
 * calls enqueue their iterations eagerly (no hipGraph replay) with HIP events on the library's stream around the launches
 * of every stride-th iteration; dpr_get_nj_kernel_timing returns the kernels per iteration, the average microseconds per
 * kernel (launch order, up to 4 entries) and the number of sampled iterations; dpr_nj_kernel_name(i) names kernel i. */
int dpr_ctx_set_nj_kernel_timing(dpr_ctx *ctx, int stride);
int dpr_get_nj_kernel_timing(dpr_ctx *ctx, int *kernels, double *us_avg4, int64_t *samples);
const char *dpr_nj_kernel_name(int idx);
/* debug, needs DPR_NJ_PHASES=<iteration>: 2 kernels x 2048 blocks x 8 phase stamps (100 MHz ticks, 0 = none) of that
 * iteration of the last pruned NJ run (profiles/nj_phases.py) */
int dpr_get_nj_phase_stamps(uint64_t *out32768);
/* debug (profiles/njp_list_shape.py): after a dpr_nj_run that stopped early on the default single-GPU plan, the units the next
 * scan would walk -- codes (sub-unit mask << 28 | strip << 18 | row group), *count of them (at most cap copied) -- the number
 * of positions of the current epoch and, if ur != NULL, the row sums U / (n - 2) by position (NaN: dead / in quarantine) */
int dpr_get_njp_list(dpr_ctx *ctx, int32_t *out, int64_t cap, int64_t *count, int64_t *positions, double *ur, int64_t ur_cap);
/* host-only: owner of the 16-row x 512-column unit (strip = column block, group = row group) among `world` ranks
 * when the position space has P positions (units are tested in blocks of one strip x 256 consecutive row groups,
 * strip-major; test block t and its units belong to rank t mod world); -1 if the unit holds no pair of the strict
 * lower triangle */
int dpr_njp_unit_owner(int64_t strip, int64_t group, int64_t P, int world);
/* pruned path: 16x512 units scanned since dpr_dist_matrix, and units of one full scan */
int dpr_get_prune_stats(dpr_ctx *ctx, uint64_t *units_scanned, uint64_t *units_per_full_scan);
/* state of the NJ run after the last dpr_nj_run: iterations done since dpr_dist_matrix and active size -- also after a
 * call that ended with DPR_ERR_NOCAND (the reference's undefined (0,0) merge, src/neighborJoining.cu:134-141,214): the merge
 * log up to there has been copied to the caller's arrays, this says how many entries it holds */
int dpr_get_nj_progress(dpr_ctx *ctx, int64_t *iterations_done, int64_t *active);
/* pruned path, current epoch: positions, row groups per test block, strips per test block, 1 if the large-shape post
 * kernel (njp_post2_kernel) serves it, blocks of the unit scan (tests: which launch shape a run really used) */
int dpr_get_njp_shape(dpr_ctx *ctx, int64_t *positions, int *row_groups, int *strips, int *post2, int *scan_grid);

/* microbenchmark: microseconds per launch of a chain of `nlaunch` trivial dependent kernels of `grid`
 * blocks on the context's stream, eager (0) or hipGraph replay of 128-node chains (1) */
int dpr_launch_bench(dpr_ctx *ctx, int nlaunch, int grid, int use_graph, float *us_per_launch);
/* measurement aid: a background load of `blocks` 64-thread workgroups spinning on FMAs on a stream of its own until
 * dpr_spin_stop (or max_ms at the latest) -- to see whether the latency-bound loops run at reduced clocks on an idle chip */
int dpr_spin_start(dpr_ctx *ctx, int blocks, int max_ms);
int dpr_spin_stop(dpr_ctx *ctx);

/* tuning knobs of the streaming Q-argmin scan (process-wide): rows per work unit (16 or 64; +128 selects the
 * filtered candidate update), non-temporal loads (0/1), grid size (0 = default 2048, <= 8192).  Results never
 * depend on them. */
int dpr_scan_tune(int rows_per_unit, int nontemporal, int grid);

/* calibration: plain streaming read (16 B/lane, optional non-temporal) of `bytes` of the matrix
 * buffer the Q-argmin scans (the position-space matrix in pruned mode); average milliseconds per pass.  Gives the
 * read ceiling the scan is compared with. */
int dpr_bw_probe(dpr_ctx *ctx, int64_t bytes, int nontemporal, int grid, int reps, float *out_ms);

/* ---- test hooks ------------------------------------------------------------------------------*/
int64_t dpr_n_active(dpr_ctx *ctx);
int64_t dpr_n_total(dpr_ctx *ctx);
int dpr_get_matrix_row(dpr_ctx *ctx, int64_t i, double *out /* n_total doubles */);
int dpr_get_row_sums(dpr_ctx *ctx, double *out /* n_total doubles */);
int dpr_get_msa_counts(dpr_ctx *ctx, int64_t row, int32_t *useful, int32_t *match /* row entries */);
/* the distance block tips [row0, row0 + nrows) x tips [0, ncols) as the placement batches, --add and the divide-and-conquer
 * assignment compute it (MSADistConstructionRangeDC, src/divide_and_conquer/msa.cu:321-372, a row per launch there); out
 * (optional): [nrows][ncols], or [ncols][nrows] when transposed; *ms_avg: average duration of `reps` launches (HIP events) */
int dpr_msa_dist_block(dpr_ctx *ctx, int64_t row0, int64_t nrows, int64_t ncols, int dist_type, int transposed, double *out,
                       int reps, float *ms_avg);
/* MurmurHash3 value of every k-mer position of read `seq` (len-k+1 values), sketch kernel's hash path */
int dpr_get_kmer_hashes(dpr_ctx *ctx, int64_t seq, int k, const uint64_t *word_off, const uint64_t *len,
                        uint64_t *out);
/* closest lists (40n ints / doubles, 5 per slot) and per-tip trace (eid, frac, add) of the last
 * dpr_place_run; any pointer may be NULL */
int dpr_get_place_state(dpr_ctx *ctx, int32_t *cid, double *cdis, double *trace);
/* phase timings of the last dpr_dist_matrix / dpr_nj_run (or dpr_place_run) in milliseconds (HIP events) */
int dpr_get_timing(dpr_ctx *ctx, double *dist_ms, double *nj_ms);

/* ---- k-closest placement: KPlacementDeviceArrays::{allocateDeviceArrays,findPlacementTree,
 * addQuery} (src/placement_close_k.cu:15-68,646-854,858-990).  Adjacency arrays are in/out host
 * buffers of sizes head[2n], e/nxt/belong[8n], len[8n]; first = 2 builds from scratch, first = m
 * expects the imported backbone (initializeDeviceArrays :126-264) in the arrays. */
int dpr_place_run(dpr_ctx *ctx, int source, int dist_type, int k, int64_t first, int64_t n,
                  int32_t *head, int32_t *e, int32_t *nxt, int32_t *belong, double *len);

/* distance and tree part of the last dpr_place_run in milliseconds (the reference prints them as "Distance Operation
 * Time" / "Tree Operation Time", src/placement_close_k.cu:852-853,985-986).  The two add up to the run.  Without
 * overlap dist_ms is the HIP-event time of the distance batches.  Mash input computes the rows of the next batch on a
 * second stream beside the tree kernels: dist_ms is then the time the tree stream WAITED for distance rows (the
 * non-overlapped remainder), not the batches' own duration -- that one is dpr_get_place_overlap's dist_busy_ms */
int dpr_get_place_timing(dpr_ctx *ctx, double *dist_ms, double *tree_ms);
/* *overlapped = 1 if the last placement run computed its distance rows beside the tree kernels; *dist_busy_ms = time the
 * distance batches were in flight then (concurrent with tree work: not a summand of the wall time); 0 / 0.0 otherwise */
int dpr_get_place_overlap(dpr_ctx *ctx, int *overlapped, double *dist_busy_ms);
/* The overlap decision is taken per batch of distance rows (1 024 tips for Mash input): a batch is produced beside the tree
 * kernels of the previous one only while its distance part is the shorter of the two (src/placement_close_k.cu:756-851 computes
 * one row per tip, serially).  *batches / *overlapped_batches of the last placement run.  DPR_PLACE_NO_OVERLAP=1: never,
 * DPR_PLACE_OVERLAP_ALWAYS=1: round 3's policy (every batch).  Results do not depend on it. */
int dpr_get_place_policy(dpr_ctx *ctx, int64_t *batches, int64_t *overlapped_batches);

/* Per placed tip of the last placement run: slots its closest-list walk reached (updateClosestNodes, src/placement_close_k.cu:
 * 86-124) beyond the two rounds applied with the split; negative = -(reached + 1): the walk of a node of degree > 3 (imported
 * backbones).  The update launch of a tip grows with it (64 queue entries per round trip): the outliers of those launches.
 * stats6: tips whose walk left the 2 048-entry LDS queue, largest walk, sum, tips on the degree > 3 walk, and -- the second kind of
 * outlier, four-tip launches only -- tips for which the launch evaluated EVERY slot itself because the set of slots changed by the
 * launch's earlier tips overflowed / because more than 128 blocks had to be re-scanned.  DPR_LOG=place prints the six numbers. */
int dpr_get_place_walks(dpr_ctx *ctx, int32_t *reached /* n entries or NULL */, int64_t *stats6);

/* ---- exact placement mode: PlacementDeviceArrays::{allocateDeviceArrays,findPlacementTree}
 * (src/placement.cu:17-117,508-789), reached in the reference through `-m 0` with 30000 <= n < 1000000
 * (SURVEY 9.2).  Same inputs/outputs as dpr_place_run with first = 2; per tip the per-slot bounds come
 * from an exact bottom-up / top-down pass over the whole tree instead of the K = 5 closest lists. */
int dpr_place_exact_run(dpr_ctx *ctx, int source, int dist_type, int k, int64_t n, int32_t *head, int32_t *e,
                        int32_t *nxt, int32_t *belong, double *len);
/* test hook: reverse-slot array (8n) and node depths below node n (2n) of the last dpr_place_exact_run */
int dpr_get_exact_state(dpr_ctx *ctx, int32_t *rev, int32_t *dep);

/* ---- divide-and-conquer mode: KPlacementDeviceArraysDC::{findBackboneTreeDC, findClustersDC,
 * findClusterTreeDC} (src/divide_and_conquer/placement_close_k.cu:731-935, 937-1113, 1251-1535) with the
 * DC row providers MashDeviceArraysDC / MSADeviceArraysDC (src/divide_and_conquer/mash.cu:453-755,
 * msa.cu:219-504); dispatch at src/tree_generation.cu:422-449,541-575 (`-m 3` or n >= 1 000 000).
 * Tips [0, backbone) form the backbone (the CLI passes n/20, src/tree_generation.cu:425,545), every
 * other tip is assigned to a backbone edge and placed inside that edge's cluster.  Unlike the
 * reference nothing is staged through host memory: all planes / sketches stay in HBM.
 * Adjacency outputs as dpr_place_run (internal node ids start at n; root-side node n);
 * cluster_id (optional, n entries): -1 for backbone tips, else the backbone slot of the tip's cluster.
 * flags: DPR_DC_EXACT_LAST computes the distance of a query to the LAST backbone tip also for aligned
 * input (the reference's kernel stops one short, src/divide_and_conquer/msa.cu:331, and scans a 0.0;
 * that behaviour is the default so that results match the reference). */
#define DPR_DC_EXACT_LAST 1
/* Multi-GPU: after dpr_comm_init every rank calls dpr_dc_run with the same (replicated) inputs; the
 * backbone is built identically on every rank, the query tips and the clusters are shared out, cluster ids
 * and the state changes are summed over RCCL (two all-reduces per run), and every rank returns the full
 * tree.  DPR_DC_VIRTUAL_RANKS(w) emulates w ranks one after the other on a single GPU (validation of the
 * sharding and of the merge; same result as w = 1, bit for bit). */
#define DPR_DC_VIRTUAL_RANKS(w) (((w) & 0xff) << 8)
int dpr_dc_run(dpr_ctx *ctx, int source, int dist_type, int k, int64_t n, int64_t backbone, int flags,
               int32_t *head, int32_t *e, int32_t *nxt, int32_t *belong, double *len, int32_t *cluster_id);
/* host-only helpers of the multi-GPU split (pure functions; used by dpr_dc_run itself and by the CPU tests):
 * share of the query tips [backbone, n) of `rank` (contiguous, multiples of 256 tips) ... */
int dpr_dc_query_share(int64_t n, int64_t backbone, int rank, int world, int64_t *q0, int64_t *q1);
/* ... and the owner of every cluster: sizes[] in the order dpr_dc_run builds them (descending, ties by
 * ascending slot); largest first onto the least-loaded rank (load = m * (m + 20)), ties to the lowest rank */
int dpr_dc_deal_clusters(const int64_t *sizes_desc, int64_t count, int world, int32_t *owner);
/* counts: clusters, largest cluster, in-cluster pair distances, memory groups, pair jobs;
 * phase_ms: backbone tree, cluster assignment, cluster trees (HIP events) */
int dpr_get_dc_stats(dpr_ctx *ctx, int64_t *counts5, double *phase_ms3);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* DIPPER_HIP_H */
