// C ABI, tree builders that place tips: k-closest placement (batches of distance rows, overlap policy), exact placement mode,
// divide-and-conquer; dpr_place_run, dpr_place_exact_run, dpr_dc_run and their getters.
#include "ctx_internal.hpp"

using namespace dpr;

extern "C" {

// k-closest placement of tips [first, last) into c->place (findPlacementTree / addQuery loop,
// src/placement_close_k.cu:756-851,888-987; findBackboneTreeDC, src/divide_and_conquer/
// placement_close_k.cu:832-925): distance rows in batches of 256 (1024 for Mash input) from the row providers.
// first == 2 starts from the two-tip tree, otherwise the imported backbone is already in the arrays.
static int place_range(dpr_ctx* c, int source, int dist_type, int64_t first, int64_t last)
{
    PlaceBuffers& p = c->place;
    // distance rows per batch: 1024 for Mash input, whose batches run beside the tree kernels (100 000 unaligned tips:
    // 3.81 / 3.60 / 3.93 s with 256 / 1024 / 4096 -- fewer launch tails, but a longer start-up without overlap)
    int64_t R = source == DPR_SRC_MASH ? 1024 : 256;
    if (const char* e = std::getenv("DPR_PLACE_BATCH")) { const int64_t v = std::atoll(e); if (v >= 16 && v <= 65536) R = v; }
    const int64_t ldb = (last + 15) / 16 * 16;
    // Multi-GPU (dpr_comm_init done, inputs replicated): the distance rows of a batch do not depend on the
    // placements, so every rank computes R/world of them and one all-gather per batch completes the block;
    // the tree kernels then run identically on every rank (deterministic), so no tree state is exchanged.
    const bool sharded = comm_real(c) && source != DPR_SRC_MATRIX;
    const int W = sharded ? c->world : 1;
    const int64_t per = (R + W - 1) / W;         // rows per rank and batch
    // The distance rows of the NEXT batch may be produced on a second stream while the tree kernels of the current batch run
    // (they are latency-bound and occupy a few workgroups; the pair kernels fill the rest of the chip): two row buffers, the
    // producer waits for the batch that last read the buffer it overwrites.  Mash input only: with aligned input the distance part
    // is 4 % of the run and the contention costs more than it hides (1.63 -> 1.82 s at 100 000 tips).
    // Round 4: the decision is taken PER BATCH.  Overlap pays while a batch's distance part is the SHORTER one -- it then
    // disappears behind the tree kernels (100 000 unaligned tips from scratch: 3.2 -> 2.5 s).  Where it is the longer one
    // nothing can hide it, and sharing the chip slows both sides: adding 50 000 queries to a 500 000-tip backbone, every batch is
    // 5 x 10^8 pairs (~100 ms alone) against ~50 ms of tree kernels; overlapped, the pair kernel ran at half its rate and the
    // update kernel 5.6 x slower (profiles/r3/kernel_stats_add_mash_500k_plus_50k.csv): 9.2 s where back to back is 7.6 s.
    // So: batch k + 1 is produced beside batch k's tree kernels iff its predicted time alone (pairs / the rate measured on this
    // run's batches that ran alone, 4.5 G pairs/s until there is one) is below the tree time of the latest finished batch
    // (deflated by 1.4 if that batch shared the chip); otherwise it is produced on the main stream right before its own tips, at
    // full chip.  Measured (profiles/r4/place_policy_*.jsonl): --add 500 000 + 50 000 through Mash 8.87 s (every batch beside)
    // -> 6.56 s (none); 100 000 tips from scratch 3.07 s (none) / 2.52 s (every batch) / 2.5x s (policy).  The host waits for batch k - 1 before it decides about batch k + 1 (it never runs more than one batch ahead
    // of the device any more; enqueueing is ~10 x faster than the tree kernels execute, so the device does not starve).
    // Round 5: while the tree kernels are the one-tip launch pairs (a 780-block scan and a one-workgroup update every 15 us) EVERY
    // batch goes beside them, longer than the tree part or not -- 100 000 unaligned tips 2.18 -> 1.72 s (mean branch 2e-5), 2.30 ->
    // 2.14 s (1e-3).  The rule above stays for the four-tip launch pairs (>= 150 000 tips): their scan fills the chip for 54 us of
    // every ~120, so the pair kernel beside it gets half a chip (--add through Mash, every batch beside: 6.8 s against 4.0 s; the
    // 1 024-thread update workgroup needs an empty CU and waits 0.5 ms for one).
    // Results cannot depend on the policy: the rows are the same numbers whichever stream produced them.
    // (ranks on the window transport of dpr_comm_init_shared: its all-gather is synchronous with the host, nothing would overlap)
    const bool overlap_allowed = source == DPR_SRC_MASH && !std::getenv("DPR_PLACE_NO_OVERLAP") && !(sharded && !c->comm);
    // (several ranks: every rank must take the same decisions -- the batches' all-gathers are enqueued on the stream the decision
    //  picks -- and a rank's share of a batch is 1 / G of the pairs, i.e. the short side: every batch beside, as in round 3)
    const bool overlap_always = overlap_allowed && sharded;
    double* rows_buf[2] = { nullptr, nullptr };
    const size_t row_bytes = sizeof(double) * (size_t)(per * W * ldb);
    if (source != DPR_SRC_MATRIX) {
        DPR_HIP(hipMalloc(&rows_buf[0], row_bytes));
        if (overlap_allowed) {
            const hipError_t me = hipMalloc(&rows_buf[1], row_bytes);
            if (me != hipSuccess) { (void)hipFree(rows_buf[0]); return hip_fail(me, "hipMalloc(second row buffer)"); }
        }
    }
    if (overlap_allowed && !c->stream2) {
        // lowest priority: the distance kernels fill the chip, the tree kernels of the current batch (one wavefront or a few
        // blocks each, on the context's stream) must not queue behind them
        int least = 0, greatest = 0;
        DPR_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        DPR_HIP(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, least));
    }
    std::vector<hipEvent_t> sync_ev;                             // fill-done / tree-done events of this run
    auto new_event = [&](hipEvent_t* e) -> int { DPR_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming)); sync_ev.push_back(*e); return DPR_OK; };
    auto row_ptr = [&](int64_t i, int64_t i0, const double* rows) -> const double* {
        return source == DPR_SRC_MATRIX ? c->packed_lower + i * (i - 1) / 2 : rows + (i - i0) * ldb;
    };
    auto fill_some = [&](int64_t i0, int64_t nr, double* out, hipStream_t st) -> int {
        if (nr <= 0) return DPR_OK;
        if (source == DPR_SRC_MSA) return msa_dist_block_rows(c->msa, i0, nr, 0, 0, i0 + nr, dist_type, out, ldb, st);
        if (source == DPR_SRC_MASH) return mash_dist_rows(c->mash, i0, nr, 0, 0, false, i0 + nr, out, ldb, st);
        return DPR_OK;
    };
    auto fill_rows_inner = [&](int64_t i0, int64_t nr, double* rows, hipStream_t ds) -> int {
        if (!sharded) return fill_some(i0, nr, rows, ds);
        const int64_t a = (int64_t)c->rank * per, b = a + per < nr ? a + per : nr;     // this rank's rows of the batch
        if (int rc = fill_some(i0 + a, b - a, rows + a * ldb, ds)) return rc;
        return comm_all_gather(c, rows + a * ldb, rows, sizeof(double) * (size_t)(per * ldb), ds);
    };
    // the reference reports the distance and the tree part of a placement run separately
    // (src/placement_close_k.cu:852-853,985-986).  A batch produced on the main stream: an event pair around it (c->place_ev).
    // A batch produced beside the tree kernels: its own interval overlaps the tree work in wall time (and stretches while it
    // shares the chip) -- kept as `busy` time (c->place_ev_busy); what counts as distance time is the time the tree stream
    // actually WAITED for it (an event pair around the wait, c->place_ev), so distance + tree = the run's wall time again.
    c->place_overlapped = false;
    c->place_batches = 0; c->place_batches_overlapped = 0;
    auto fill_rows = [&](int64_t i0, int64_t nr, double* rows, bool beside, hipEvent_t* t0, hipEvent_t* t1) -> int {
        hipStream_t ds = beside ? c->stream2 : c->stream;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (source != DPR_SRC_MATRIX) {
            DPR_HIP(hipEventCreate(&e0)); DPR_HIP(hipEventCreate(&e1));
            std::vector<hipEvent_t>& dst = beside ? c->place_ev_busy : c->place_ev;
            dst.push_back(e0); dst.push_back(e1);
            DPR_HIP(hipEventRecord(e0, ds));
        }
        c->mash.share_chip = beside;
        const int rc = fill_rows_inner(i0, nr, rows, ds);
        c->mash.share_chip = false;
        if (e1) DPR_HIP(hipEventRecord(e1, ds));
        if (t0) *t0 = e0;
        if (t1) *t1 = e1;
        return rc;
    };
    auto run = [&]() -> int {
        if (first == 2) {
            if (int rc = place_init_fresh(p, c->stream)) return rc;
            if (int rc = fill_some(1, 1, rows_buf[0], c->stream)) return rc;
            if (int rc = place_initial_tree(p, row_ptr(1, 1, rows_buf[0]), c->stream)) return rc;
        } else {
            if (int rc = place_import_backbone(p, first, c->stream)) return rc;
        }
        if (first >= last) return DPR_OK;
        hipEvent_t filled[2] = { nullptr, nullptr }, consumed[2] = { nullptr, nullptr };
        bool ahead = false;                       // the rows of the batch about to be placed were produced beside the previous batch
        // policy state: what a batch that ran alone cost
        double tree_ms_per_tip = -1.0, pairs_per_ms = 4.5e6;
        struct Probe { hipEvent_t d0, d1, t0, t1; double pairs; int64_t nr; bool dist_alone, tree_alone; };
        std::vector<Probe> probes;                // one per batch
        size_t harvested = 0;
        auto batch_pairs = [&](int64_t i0, int64_t nr) { return (double)nr * ((double)i0 + 0.5 * (double)(nr - 1)); };
        auto harvest = [&](size_t upto) -> int {  // read the timings of the batches < upto (host waits for the last of them)
            for (; harvested < upto; ++harvested) {
                Probe& pr = probes[harvested];
                if (!pr.t1) continue;
                DPR_HIP(hipEventSynchronize(pr.t1));
                float ms = 0;
                // (tree kernels that shared the chip with a distance batch ran ~1.3 x slower at 100 000 tips: such a batch's time
                //  is deflated by 1.4 before it stands for "the tree part alone" -- the tree part grows with the tree, so the
                //  latest batch is the better estimate than batch 0's clean one)
                if (hipEventElapsedTime(&ms, pr.t0, pr.t1) == hipSuccess && pr.nr > 0) tree_ms_per_tip = (double)ms / (double)pr.nr / (pr.tree_alone ? 1.0 : 1.4);
                if (pr.dist_alone && pr.d0 && pr.d1 && pr.pairs >= 5.0e7 && hipEventElapsedTime(&ms, pr.d0, pr.d1) == hipSuccess && ms > 0.0f)
                    pairs_per_ms = pr.pairs / (double)ms;
            }
            return DPR_OK;
        };
        int64_t i0 = first;
        int cur = 0;
        for (size_t k = 0; i0 < last; i0 += R, cur ^= 1, ++k) {
            const int64_t nr = last - i0 < R ? last - i0 : R;
            const int64_t j0 = i0 + R;
            double* rows = rows_buf[overlap_allowed ? cur : 0];
            Probe pr{ nullptr, nullptr, nullptr, nullptr, batch_pairs(i0, nr), nr, false, true };
            if (!ahead) {
                // this batch's rows on the main stream, at full chip (a buffer's last reader ran on this stream: ordered)
                if (int rc = fill_rows(i0, nr, rows, false, &pr.d0, &pr.d1)) return rc;
                pr.dist_alone = true;
            } else {
                hipEvent_t w0 = nullptr, w1 = nullptr;
                DPR_HIP(hipEventCreate(&w0)); DPR_HIP(hipEventCreate(&w1));
                c->place_ev.push_back(w0); c->place_ev.push_back(w1);
                DPR_HIP(hipEventRecord(w0, c->stream));
                DPR_HIP(hipStreamWaitEvent(c->stream, filled[cur], 0));
                DPR_HIP(hipEventRecord(w1, c->stream));
            }
            ++c->place_batches;
            // the next batch beside this batch's tree kernels?
            bool next_ahead = false;
            if (overlap_allowed && j0 < last) {
                const int64_t nr2 = last - j0 < R ? last - j0 : R;
                if (overlap_always || i0 + nr <= place_multi_min()) next_ahead = true;      // (one tip per launch pair: always, see above)
                else {
                    if (k >= 1) { if (int rc = harvest(k)) return rc; }      // batches 0 .. k-1 (the host waits for batch k-1 here)
                    const double dist_alone_ms = batch_pairs(j0, nr2) / pairs_per_ms;
                    // (no tree timing yet -- this is batch 0: its successor is produced alone too, unless its distance part is tiny)
                    next_ahead = tree_ms_per_tip > 0.0 ? dist_alone_ms < tree_ms_per_tip * (double)nr : dist_alone_ms < 1.0;
                }
                if (next_ahead) {
                    const int nb = cur ^ 1;
                    if (consumed[nb]) DPR_HIP(hipStreamWaitEvent(c->stream2, consumed[nb], 0));
                    else {
                        // (first use of that buffer by the second stream: everything enqueued so far may still read it)
                        hipEvent_t e;
                        if (int rc = new_event(&e)) return rc;
                        DPR_HIP(hipEventRecord(e, c->stream));
                        DPR_HIP(hipStreamWaitEvent(c->stream2, e, 0));
                    }
                    if (int rc = fill_rows(j0, nr2, rows_buf[nb], true, nullptr, nullptr)) return rc;
                    if (int rc = new_event(&filled[nb])) return rc;
                    DPR_HIP(hipEventRecord(filled[nb], c->stream2));
                    c->place_overlapped = true;
                    ++c->place_batches_overlapped;
                    pr.tree_alone = false;
                }
            }
            if (source != DPR_SRC_MATRIX) {
                DPR_HIP(hipEventCreate(&pr.t0)); DPR_HIP(hipEventCreate(&pr.t1));
                c->place_ev_tree.push_back(pr.t0); c->place_ev_tree.push_back(pr.t1);
                DPR_HIP(hipEventRecord(pr.t0, c->stream));
            }
            if (source == DPR_SRC_MATRIX) {      // packed triangle: rows are not evenly spaced
                for (int64_t i = i0; i < i0 + nr; ++i)
                    if (int rc = place_tip(p, row_ptr(i, i0, rows), i, c->place_trace, c->stream)) return rc;
            } else {
                if (int rc = place_tips(p, rows, ldb, i0, nr, c->place_trace, c->stream)) return rc;
            }
            if (pr.t1) DPR_HIP(hipEventRecord(pr.t1, c->stream));
            if (overlap_allowed) { if (int rc = new_event(&consumed[cur])) return rc; DPR_HIP(hipEventRecord(consumed[cur], c->stream)); }
            probes.push_back(pr);
            ahead = next_ahead;
        }
        return DPR_OK;
    };
    const int rc = run();
    c->mash.share_chip = false;
    if (rows_buf[0] || rows_buf[1]) {
        (void)hipStreamSynchronize(c->stream);
        if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        for (double* q : rows_buf) if (q) (void)hipFree(q);
    }
    for (hipEvent_t e : sync_ev) (void)hipEventDestroy(e);
    return rc;
}

// sum of the distance-batch event pairs of the run that just finished (stream idle); the events are released
static void place_collect_dist_ms(dpr_ctx* c)
{
    auto sum = [](std::vector<hipEvent_t>& evs) {
        double tot = 0;
        for (size_t i = 0; i + 1 < evs.size(); i += 2) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, evs[i], evs[i + 1]) == hipSuccess) tot += ms;
        }
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        evs.clear();
        return tot;
    };
    c->place_dist_ms = sum(c->place_ev);
    c->place_dist_busy_ms = sum(c->place_ev_busy);
    (void)sum(c->place_ev_tree);       // (the policy's probes; released here)
}

// dist_ms: the part of the run the tree kernels could not proceed for want of distance rows (without overlap: the
// distance batches themselves; with overlap: the tree stream's waits for them); tree_ms: the rest of the run.
int dpr_get_place_timing(dpr_ctx* c, double* dist_ms, double* tree_ms)
{
    if (!c) { set_error("dpr_get_place_timing: null ctx"); return DPR_ERR_ARG; }
    if (dist_ms) *dist_ms = c->place_dist_ms;
    if (tree_ms) *tree_ms = c->nj_ms > c->place_dist_ms ? c->nj_ms - c->place_dist_ms : 0.0;
    return DPR_OK;
}

// overlap mode of the last placement run: *overlapped = 1 and *dist_busy_ms = time the distance batches were in flight on
// the second stream (concurrent with the tree kernels, so NOT a summand of the run's wall time); else 0 / 0
int dpr_get_place_overlap(dpr_ctx* c, int* overlapped, double* dist_busy_ms)
{
    if (!c) { set_error("dpr_get_place_overlap: null ctx"); return DPR_ERR_ARG; }
    if (overlapped) *overlapped = c->place_overlapped ? 1 : 0;
    if (dist_busy_ms) *dist_busy_ms = c->place_overlapped ? c->place_dist_busy_ms : 0.0;
    return DPR_OK;
}

// batches of the last placement run and how many of them were produced beside the previous batch's tree kernels (the per-batch
// overlap policy of place_range)
int dpr_get_place_policy(dpr_ctx* c, int64_t* batches, int64_t* overlapped_batches)
{
    if (!c) { set_error("dpr_get_place_policy: null ctx"); return DPR_ERR_ARG; }
    if (batches) *batches = c->place_batches;
    if (overlapped_batches) *overlapped_batches = c->place_batches_overlapped;
    return DPR_OK;
}

int dpr_place_run(dpr_ctx* c, int source, int dist_type, int k, int64_t first, int64_t n, int32_t* head,
                  int32_t* e, int32_t* nxt, int32_t* belong, double* len)
{
    if (!c || !head || !e || !nxt || !belong || !len || n < 3 || first < 2 || first > n) { set_error("dpr_place_run: bad argument"); return DPR_ERR_ARG; }
    DPR_HIP(hipSetDevice(c->device));
    if (source == DPR_SRC_MSA) {
        if (!c->msa.planes || c->msa.n != n) { set_error("dpr_place_run: call dpr_set_msa with n sequences first"); return DPR_ERR_STATE; }
    } else if (source == DPR_SRC_MASH) {
        if (!c->mash.sketches || c->mash.n != n) { set_error("dpr_place_run: call dpr_set_reads and dpr_sketch first"); return DPR_ERR_STATE; }
        if (k != c->mash.k) { set_error("dpr_place_run: k differs from the sketch k"); return DPR_ERR_ARG; }
    } else if (source == DPR_SRC_MATRIX) {
        if (!c->packed_lower || c->n_input != n) { set_error("dpr_place_run: call dpr_set_matrix_lower first"); return DPR_ERR_STATE; }
    } else { set_error("dpr_place_run: unknown source"); return DPR_ERR_ARG; }
    if (first > 2) {
        // An imported backbone must be a rooted binary tree: `first` tips, first - 1 internal nodes, 2 first - 2 edges = the slots
        // [0, 4 first - 4) all in use.  The reference's scan reads head[e[slot]] of every slot below 4 num - 4
        // (src/placement_close_k.cu:325-338): an unused slot (e = belong = -1: a trifurcating root, a polytomy) is an
        // out-of-bounds read there; here it is an error, since the edge records (one per undirected edge, dense: 2 num - 2 after
        // num tips) have no place for a missing edge either (advisor, round 5).
        for (int64_t s = 0; s < 4 * first - 4; ++s)
            if (e[s] < 0 || belong[s] < 0) {
                set_error("dpr_place_run: the backbone is not a rooted binary tree (directed edge slot " + std::to_string(s) + " of " +
                          std::to_string(4 * first - 4) + " is unused: a trifurcating root or a polytomy); resolve it first");
                return DPR_ERR_ARG;
            }
    }
    if (int rc = place_alloc(c->place, n)) return rc;
    PlaceBuffers& p = c->place;
    if (c->place_trace) { (void)hipFree(c->place_trace); c->place_trace = nullptr; }
    DPR_HIP(hipMalloc(&c->place_trace, sizeof(double) * (size_t)(3 * n)));
    DPR_HIP(hipMemsetAsync(c->place_trace, 0, sizeof(double) * (size_t)(3 * n), c->stream));
    DPR_HIP(hipMemsetAsync(p.misc + 2, 0, 2 * sizeof(int32_t), c->stream));      // fallback counters of the four-tip launches (dpr_get_place_walks)
    if (first > 2) {
        DPR_HIP(hipMemcpyAsync(p.head, head, sizeof(int32_t) * (size_t)(2 * n), hipMemcpyHostToDevice, c->stream));
        DPR_HIP(hipMemcpyAsync(p.e, e, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyHostToDevice, c->stream));
        DPR_HIP(hipMemcpyAsync(p.nxt, nxt, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyHostToDevice, c->stream));
        DPR_HIP(hipMemcpyAsync(p.belong, belong, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyHostToDevice, c->stream));
        DPR_HIP(hipMemcpyAsync(p.len, len, sizeof(double) * (size_t)(8 * n), hipMemcpyHostToDevice, c->stream));
    }
    DPR_HIP(hipEventRecord(c->ev[2], c->stream));
    if (int rc = place_range(c, source, dist_type, first, n)) { place_collect_dist_ms(c); return rc; }
    DPR_HIP(hipEventRecord(c->ev[3], c->stream));
    DPR_HIP(hipMemcpyAsync(head, p.head, sizeof(int32_t) * (size_t)(2 * n), hipMemcpyDeviceToHost, c->stream));
    DPR_HIP(hipMemcpyAsync(e, p.e, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
    DPR_HIP(hipMemcpyAsync(nxt, p.nxt, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
    DPR_HIP(hipMemcpyAsync(belong, p.belong, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
    DPR_HIP(hipMemcpyAsync(len, p.len, sizeof(double) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
    DPR_HIP(hipStreamSynchronize(c->stream));
    float ms = 0;
    DPR_HIP(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
    c->nj_ms = ms;
    place_collect_dist_ms(c);
    if (log_level("place") > 0) {
        int64_t st[6] = { 0, 0, 0, 0, 0, 0 };
        if (dpr_get_place_walks(c, nullptr, st) == DPR_OK)
            std::fprintf(stderr, "[place] closest-list walks of %lld tips: %lld slots reached in all, largest walk %lld, %lld beyond the 2 048-entry LDS queue, %lld from a node of degree > 3; "
                         "four-tip launches: %lld tips fell back to evaluating every slot (dirty set full), %lld (too many blocks to re-scan)\n",
                         (long long)(n - first), (long long)st[2], (long long)st[1], (long long)st[0], (long long)st[3], (long long)st[4], (long long)st[5]);
    }
    return DPR_OK;
}

// ---- exact placement mode -----------------------------------------------------------------------------
static int place_exact_attempt(dpr_ctx* c, int source, int dist_type, int k, int64_t n, int32_t* head, int32_t* e,
                               int32_t* nxt, int32_t* belong, double* len)
{
    if (!c || !head || !e || !nxt || !belong || !len || n < 3) { set_error("dpr_place_exact_run: bad argument"); return DPR_ERR_ARG; }
    DPR_HIP(hipSetDevice(c->device));
    if (source == DPR_SRC_MSA) {
        if (!c->msa.planes || c->msa.n != n) { set_error("dpr_place_exact_run: call dpr_set_msa with n sequences first"); return DPR_ERR_STATE; }
    } else if (source == DPR_SRC_MASH) {
        if (!c->mash.sketches || c->mash.n != n) { set_error("dpr_place_exact_run: call dpr_set_reads and dpr_sketch first"); return DPR_ERR_STATE; }
        if (k != c->mash.k) { set_error("dpr_place_exact_run: k differs from the sketch k"); return DPR_ERR_ARG; }
    } else if (source == DPR_SRC_MATRIX) {
        if (!c->packed_lower || c->n_input != n) { set_error("dpr_place_exact_run: call dpr_set_matrix_lower first"); return DPR_ERR_STATE; }
    } else { set_error("dpr_place_exact_run: unknown source"); return DPR_ERR_ARG; }
    if (int rc = place_alloc(c->place, n)) return rc;
    if (int rc = exact_alloc(c->exact, n)) return rc;
    PlaceBuffers& p = c->place;
    ExactBuffers& x = c->exact;
    if (c->place_trace) { (void)hipFree(c->place_trace); c->place_trace = nullptr; }
    DPR_HIP(hipMalloc(&c->place_trace, sizeof(double) * (size_t)(3 * n)));
    DPR_HIP(hipMemsetAsync(c->place_trace, 0, sizeof(double) * (size_t)(3 * n), c->stream));
    // distance rows in batches of R+1: the step of tip i also runs the passes of tip i+1, so a batch
    // shares its last row with the next one
    const int64_t R = 256;
    const int64_t ldb = (n + 15) / 16 * 16;
    double* rows = nullptr;
    if (source != DPR_SRC_MATRIX) DPR_HIP(hipMalloc(&rows, sizeof(double) * (size_t)((R + 1) * ldb)));
    int64_t r0 = 1;
    auto row_ptr = [&](int64_t i) -> const double* {
        return source == DPR_SRC_MATRIX ? c->packed_lower + i * (i - 1) / 2 : rows + (i - r0) * ldb;
    };
    auto fill_rows = [&](int64_t i0, int64_t nr) -> int {
        if (source == DPR_SRC_MSA) return msa_dist_block_rows(c->msa, i0, nr, 0, 0, i0 + nr, dist_type, rows, ldb, c->stream);
        if (source == DPR_SRC_MASH) return mash_dist_rows(c->mash, i0, nr, 0, 0, false, i0 + nr, rows, ldb, c->stream);
        return DPR_OK;
    };
    DPR_HIP(hipEventRecord(c->ev[2], c->stream));
    int rc = DPR_OK;
    while (!rc) {
        const int64_t nr = n - r0 < R + 1 ? n - r0 : R + 1;
        if (r0 > 1) rc = exact_adapt(x, c->stream);
        if (!rc) rc = fill_rows(r0, nr);
        if (!rc && r0 == 1) rc = exact_init(p, x, row_ptr(1), nr > 1 ? row_ptr(2) : nullptr, nr > 1, c->stream);
        for (int64_t i = r0 < 2 ? 2 : r0; !rc && i < r0 + nr - 1; ++i) rc = exact_tip(p, x, i, row_ptr(i + 1), true, c->place_trace, c->stream);
        if (rc) break;
        if (r0 + nr == n) { rc = exact_tip(p, x, n - 1, nullptr, false, c->place_trace, c->stream); break; }
        r0 = r0 + nr - 1;
    }
    if (!rc) {
        DPR_HIP(hipEventRecord(c->ev[3], c->stream));
        DPR_HIP(hipMemcpyAsync(head, p.head, sizeof(int32_t) * (size_t)(2 * n), hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipMemcpyAsync(e, p.e, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipMemcpyAsync(nxt, p.nxt, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipMemcpyAsync(belong, p.belong, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipMemcpyAsync(len, p.len, sizeof(double) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
    }
    const hipError_t se = hipStreamSynchronize(c->stream);
    if (rows) (void)hipFree(rows);
    if (rc) return rc;
    DPR_HIP(se);
    float ms = 0;
    DPR_HIP(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
    c->nj_ms = ms;
    return DPR_OK;
}

int dpr_place_exact_run(dpr_ctx* c, int source, int dist_type, int k, int64_t n, int32_t* head, int32_t* e,
                        int32_t* nxt, int32_t* belong, double* len)
{
    if (!c) { set_error("dpr_place_exact_run: bad argument"); return DPR_ERR_ARG; }
    // The fast schedule (small subtrees on all CUs + top tree in LDS) gives the reference's lim[] whenever the reference's
    // depths are the tree's depths.  They stop being that only if the default tuple (slot 0, pendant length 2) wins an argmin
    // (updateTreeStructure's swap, src/placement.cu:236-239); the run is then repeated with the literal level-by-depth
    // schedule, the only one that reproduces what the reference computes from there on.
    c->exact.literal = std::getenv("DPR_EXACT_LITERAL") != nullptr;
    int rc = place_exact_attempt(c, source, dist_type, k, n, head, e, nxt, belong, len);
    if (rc != DPR_OK || c->exact.literal) return rc;
    bool quirk = false;
    if (int rq = exact_quirk(c->exact, c->stream, &quirk)) return rq;
    if (!quirk) return DPR_OK;
    c->exact.literal = true;
    rc = place_exact_attempt(c, source, dist_type, k, n, head, e, nxt, belong, len);
    return rc;
}

int dpr_get_exact_state(dpr_ctx* c, int32_t* rev, int32_t* dep)
{
    if (!c || !c->exact.dep || !c->place.rev) { set_error("dpr_get_exact_state: no exact placement state"); return DPR_ERR_STATE; }
    DPR_HIP(hipStreamSynchronize(c->stream));   // the plain copies below run on the null stream, which does not wait for c->stream
    const int64_t n = c->place.N;
    if (rev) DPR_HIP(hipMemcpy(rev, c->place.rev, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost));
    if (dep) DPR_HIP(hipMemcpy(dep, c->exact.dep, sizeof(int32_t) * (size_t)(2 * n), hipMemcpyDeviceToHost));
    return DPR_OK;
}

// ---- divide-and-conquer mode ------------------------------------------------------------------------
int dpr_dc_run(dpr_ctx* c, int source, int dist_type, int k, int64_t n, int64_t backbone, int flags, int32_t* head,
               int32_t* e, int32_t* nxt, int32_t* belong, double* len, int32_t* cluster_id)
{
    if (!c || !head || !e || !nxt || !belong || !len || n < 4) { set_error("dpr_dc_run: bad argument"); return DPR_ERR_ARG; }
    if (backbone < 3 || backbone >= n) { set_error("dpr_dc_run: backbone size must be in [3, n)"); return DPR_ERR_ARG; }
    DPR_HIP(hipSetDevice(c->device));
    if (source == DPR_SRC_MSA) {
        if (!c->msa.planes || c->msa.n != n) { set_error("dpr_dc_run: call dpr_set_msa with n sequences first"); return DPR_ERR_STATE; }
    } else if (source == DPR_SRC_MASH) {
        if (!c->mash.sketches || c->mash.n != n) { set_error("dpr_dc_run: call dpr_set_reads and dpr_sketch first"); return DPR_ERR_STATE; }
        if (k != c->mash.k) { set_error("dpr_dc_run: k differs from the sketch k"); return DPR_ERR_ARG; }
    } else {
        // src/divide_and_conquer/placement_close_k.cu:969-972
        set_error("dpr_dc_run: input must be unaligned or aligned sequences for the clustering based approach");
        return DPR_ERR_ARG;
    }
    const int64_t B = backbone;
    if (int rc = place_alloc(c->place, n, B)) return rc;
    PlaceBuffers& p = c->place;
    if (c->place_trace) { (void)hipFree(c->place_trace); c->place_trace = nullptr; }
    DPR_HIP(hipMalloc(&c->place_trace, sizeof(double) * (size_t)(3 * n)));
    DPR_HIP(hipMemsetAsync(c->place_trace, 0, sizeof(double) * (size_t)(3 * n), c->stream));
    hipEvent_t ev[4];
    for (auto& x : ev) DPR_HIP(hipEventCreate(&x));
    int32_t* d_cl = nullptr;
    double* dT = nullptr;
    uint64_t *snap_old = nullptr, *snap_acc = nullptr;
    DcTable tab;
    // ranks: RCCL ranks of dpr_comm_init, or -- validation on one GPU -- DPR_DC_VIRTUAL_RANKS(w) emulated in turn
    const bool real = comm_real(c);
    const int W = real ? c->world : (((flags >> 8) & 0xff) > 1 ? ((flags >> 8) & 0xff) : 1);
    std::vector<int32_t> h_cl((size_t)n, -1);
    auto run = [&]() -> int {
        // ---- backbone tree: tips [0, B) (findBackboneTreeDC)
        DPR_HIP(hipEventRecord(ev[0], c->stream));
        if (int rc = place_range(c, source, dist_type, 2, B)) return rc;
        DPR_HIP(hipEventRecord(ev[1], c->stream));
        // ---- cluster assignment of tips [B, n) (findClustersDC).  Multi-GPU: the backbone above is built
        // identically on every rank (same inputs, deterministic kernels); the queries are independent, so each
        // rank assigns a contiguous share and the ids are summed (zeros elsewhere) over RCCL.
        if (int rc = dc_table_build(p, B, tab, c->stream)) return rc;
        int64_t Q = ((int64_t)1 << 31) / (8 * B) / 256 * 256;
        if (Q < 256) Q = 256;
        if (Q > 8192) Q = 8192;
        const int64_t nq = n - B;
        if (Q > (nq + 255) / 256 * 256) Q = (nq + 255) / 256 * 256;
        DPR_HIP(hipMalloc(&dT, sizeof(double) * (size_t)(B * Q)));
        DPR_HIP(hipMalloc(&d_cl, sizeof(int32_t) * (size_t)(n + 1)));
        DPR_HIP(hipMemsetAsync(d_cl, 0, sizeof(int32_t) * (size_t)(n + 1), c->stream));
        // the reference's aligned-input kernel never writes the distance to backbone tip B-1
        // (src/divide_and_conquer/msa.cu:331 `idx>=ed-st`) and scans the 0.0 of a fresh allocation
        const bool skip_last = source == DPR_SRC_MSA && !(flags & DPR_DC_EXACT_LAST);
        for (int v = 0; v < W; ++v) {
            if (real && v != c->rank) continue;     // virtual ranks: every share is processed here, one after the other
            int64_t q0 = 0, q1 = 0;
            dc_query_share(n, B, v, W, &q0, &q1);
            for (int64_t i0 = q0; i0 < q1; i0 += Q) {
                const int64_t nr = q1 - i0 < Q ? q1 - i0 : Q;
                int rc;
                if (source == DPR_SRC_MSA) rc = msa_dist_block_rows(c->msa, i0, nr, 0, 0, B, dist_type, dT, Q, c->stream, true);
                else rc = mash_dist_rows(c->mash, i0, nr, 0, 0, false, B, dT, Q, c->stream, true);
                if (rc) return rc;
                if (skip_last) DPR_HIP(hipMemsetAsync(dT + (B - 1) * Q, 0, sizeof(double) * (size_t)Q, c->stream));
                if (int rc2 = dc_assign(tab, dT, Q, (int)nr, d_cl + i0, c->stream)) return rc2;
            }
        }
        if (real) {
            if (int rc = comm_all_reduce_sum(c, d_cl, (size_t)n, kNcclInt32, c->stream)) return rc;
        }
        DPR_HIP(hipMemcpyAsync(h_cl.data(), d_cl, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipEventRecord(ev[2], c->stream));
        DPR_HIP(hipStreamSynchronize(c->stream));
        for (int64_t t = 0; t < B; ++t) h_cl[(size_t)t] = -1;
        (void)hipFree(dT); dT = nullptr;
        // ---- cluster trees (findClusterTreeDC).  Multi-GPU: clusters are dealt to the ranks; an array element
        // is changed by at most one rank, so the states are merged as old + sum of (new - old) (dc_delta_*).
        size_t free_b = 0, total_b = 0;
        DPR_HIP(hipMemGetInfo(&free_b, &total_b));
        size_t budget = free_b / 2;
        if (const char* env = std::getenv("DPR_DC_BUDGET_MB")) budget = (size_t)std::atoll(env) << 20;
        if (W == 1) {
            if (int rc = dc_cluster_phase(p, h_cl.data(), n, B, source, dist_type, &c->msa, &c->mash, c->place_trace, budget,
                                          &c->dc_stats, 0, 1, c->stream)) return rc;
        } else {
            struct Arr { void* cur; int64_t words; };
            const Arr arrs[] = { { p.head, n }, { p.e, 4 * n }, { p.nxt, 4 * n }, { p.belong, 4 * n }, { p.rev, 4 * n },
                                 { p.len, 8 * n }, { p.cid, 20 * n }, { p.cdis, 40 * n }, { c->place_trace, 3 * n } };
            int64_t tot = 0;
            for (const Arr& a : arrs) tot += a.words;
            DPR_HIP(hipMalloc(&snap_old, sizeof(uint64_t) * (size_t)tot));
            if (!real) { DPR_HIP(hipMalloc(&snap_acc, sizeof(uint64_t) * (size_t)tot)); DPR_HIP(hipMemsetAsync(snap_acc, 0, sizeof(uint64_t) * (size_t)tot, c->stream)); }
            int64_t off = 0;
            for (const Arr& a : arrs) { DPR_HIP(hipMemcpyAsync(snap_old + off, a.cur, sizeof(uint64_t) * (size_t)a.words, hipMemcpyDeviceToDevice, c->stream)); off += a.words; }
            if (budget > sizeof(uint64_t) * (size_t)tot * 2) budget -= sizeof(uint64_t) * (size_t)tot * 2;
            for (int v = 0; v < W; ++v) {
                if (real && v != c->rank) continue;
                if (!real && v > 0) {                // next virtual rank starts from the backbone state again
                    off = 0;
                    for (const Arr& a : arrs) { DPR_HIP(hipMemcpyAsync(a.cur, snap_old + off, sizeof(uint64_t) * (size_t)a.words, hipMemcpyDeviceToDevice, c->stream)); off += a.words; }
                }
                if (int rc = dc_cluster_phase(p, h_cl.data(), n, B, source, dist_type, &c->msa, &c->mash, c->place_trace, budget,
                                              &c->dc_stats, v, W, c->stream)) return rc;
                off = 0;
                for (const Arr& a : arrs) {
                    if (int rc = dc_delta_sub(a.cur, snap_old + off, a.words, c->stream)) return rc;
                    if (!real) { if (int rc = dc_delta_add(snap_acc + off, a.cur, a.words, c->stream)) return rc; }
                    off += a.words;
                }
            }
            off = 0;
            for (const Arr& a : arrs) {
                if (real) {
                    if (int rc = comm_all_reduce_sum(c, a.cur, (size_t)a.words, kNcclUint64, c->stream)) return rc;
                } else {
                    DPR_HIP(hipMemcpyAsync(a.cur, snap_acc + off, sizeof(uint64_t) * (size_t)a.words, hipMemcpyDeviceToDevice, c->stream));
                }
                if (int rc = dc_delta_add(a.cur, snap_old + off, a.words, c->stream)) return rc;
                off += a.words;
            }
        }
        DPR_HIP(hipEventRecord(ev[3], c->stream));
        DPR_HIP(hipMemcpyAsync(head, p.head, sizeof(int32_t) * (size_t)(2 * n), hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipMemcpyAsync(e, p.e, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipMemcpyAsync(nxt, p.nxt, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipMemcpyAsync(belong, p.belong, sizeof(int32_t) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipMemcpyAsync(len, p.len, sizeof(double) * (size_t)(8 * n), hipMemcpyDeviceToHost, c->stream));
        DPR_HIP(hipStreamSynchronize(c->stream));
        float ms = 0;
        DPR_HIP(hipEventElapsedTime(&ms, ev[0], ev[1])); c->dc_ms[0] = ms;
        DPR_HIP(hipEventElapsedTime(&ms, ev[1], ev[2])); c->dc_ms[1] = ms;
        DPR_HIP(hipEventElapsedTime(&ms, ev[2], ev[3])); c->dc_ms[2] = ms;
        c->nj_ms = c->dc_ms[0] + c->dc_ms[1] + c->dc_ms[2];
        if (cluster_id) std::copy(h_cl.begin(), h_cl.end(), cluster_id);
        return DPR_OK;
    };
    const int rc = run();
    place_collect_dist_ms(c);     // (backbone placement batches; the events must not outlive the run)
    if (dT) (void)hipFree(dT);
    if (d_cl) (void)hipFree(d_cl);
    if (snap_old) (void)hipFree(snap_old);
    if (snap_acc) (void)hipFree(snap_acc);
    dc_table_free(tab);
    for (auto& x : ev) (void)hipEventDestroy(x);
    return rc;
}

int dpr_dc_query_share(int64_t n, int64_t backbone, int rank, int world, int64_t* q0, int64_t* q1)
{
    if (!q0 || !q1 || world < 1 || rank < 0 || rank >= world || backbone < 0 || backbone > n) { set_error("dpr_dc_query_share: bad argument"); return DPR_ERR_ARG; }
    dc_query_share(n, backbone, rank, world, q0, q1);
    return DPR_OK;
}

int dpr_dc_deal_clusters(const int64_t* sizes_desc, int64_t count, int world, int32_t* owner)
{
    if (!sizes_desc || !owner || count < 0 || world < 1) { set_error("dpr_dc_deal_clusters: bad argument"); return DPR_ERR_ARG; }
    dc_deal_clusters(sizes_desc, count, world, owner);
    return DPR_OK;
}

int dpr_get_dc_stats(dpr_ctx* c, int64_t* counts5, double* phase_ms3)
{
    if (!c) { set_error("dpr_get_dc_stats: null ctx"); return DPR_ERR_ARG; }
    if (counts5) {
        counts5[0] = c->dc_stats.clusters; counts5[1] = c->dc_stats.max_cluster; counts5[2] = c->dc_stats.pairs;
        counts5[3] = c->dc_stats.groups; counts5[4] = c->dc_stats.jobs;
    }
    if (phase_ms3) for (int i = 0; i < 3; ++i) phase_ms3[i] = c->dc_ms[i];
    return DPR_OK;
}

// Per placed tip of the last placement run: how many slots its closest-list walk reached (updateClosestNodes,
// src/placement_close_k.cu:86-124, serial there) beyond the two rounds the split applies in registers; negative = -(reached + 1):
// the walk started at the new leaf because a node of degree > 3 lies behind the split edge (imported backbones only).  One
// wavefront takes 64 queue entries per round trip, so a tip's update launch grows with this number: it is what the outliers of
// place_update_kernel is made of.  The four-tip launch (place_update_multi_kernel) has a second kind: its speculative block
// minima stand except where an earlier tip of the same launch changed an input (the "dirty" slots: the walks' slots and their
// reverses); when those outgrow the set (2 048), or more than 128 blocks must be re-scanned, the launch evaluates EVERY slot itself --
// milliseconds on a 500 000-tip tree.  stats6 (optional): tips whose walk left the 2 048-entry LDS queue, largest walk, sum over
// the tips, tips that took the degree > 3 walk, tips of four-tip launches that fell back to the full evaluation because the dirty
// set overflowed, ... because too many blocks had to be re-scanned.
int dpr_get_place_walks(dpr_ctx* c, int32_t* reached /* n, or NULL */, int64_t* stats4)
{
    if (!c || !c->place.bfs_cnt) { set_error("dpr_get_place_walks: no placement state"); return DPR_ERR_STATE; }
    DPR_HIP(hipStreamSynchronize(c->stream));
    const int64_t n = c->place.N;
    std::vector<int32_t> h((size_t)n);
    DPR_HIP(hipMemcpy(h.data(), c->place.bfs_cnt, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    if (reached) std::copy(h.begin(), h.end(), reached);
    if (stats4) {
        stats4[0] = stats4[1] = stats4[2] = stats4[3] = 0;
        int32_t misc[4] = { 0, 0, 0, 0 };
        DPR_HIP(hipMemcpy(misc, c->place.misc, sizeof misc, hipMemcpyDeviceToHost));
        stats4[4] = misc[2]; stats4[5] = misc[3];
        for (int32_t v : h) {
            const int64_t r = v < 0 ? -(int64_t)v - 1 : v;
            if (r > 2048) ++stats4[0];
            if (r > stats4[1]) stats4[1] = r;
            stats4[2] += r;
            if (v < 0) ++stats4[3];
        }
    }
    return DPR_OK;
}

int dpr_get_place_state(dpr_ctx* c, int32_t* cid, double* cdis, double* trace)
{
    if (!c || !c->place.cid) { set_error("dpr_get_place_state: no placement state"); return DPR_ERR_STATE; }
    DPR_HIP(hipStreamSynchronize(c->stream));   // the plain copies below run on the null stream, which does not wait for c->stream
    const int64_t n = c->place.N;
    if (cid) DPR_HIP(hipMemcpy(cid, c->place.cid, sizeof(int32_t) * (size_t)(40 * n), hipMemcpyDeviceToHost));
    if (cdis) DPR_HIP(hipMemcpy(cdis, c->place.cdis, sizeof(double) * (size_t)(40 * n), hipMemcpyDeviceToHost));
    if (trace) DPR_HIP(hipMemcpy(trace, c->place_trace, sizeof(double) * (size_t)(3 * n), hipMemcpyDeviceToHost));
    return DPR_OK;
}

}  // extern "C"
