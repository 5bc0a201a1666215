// k-closest placement on gfx950.  Replaces initialize (src/placement_close_k.cu:266-289),
// buildInitialTree (:530-554), calculateBranchLength (:309-358) + thrust::min_element (:807),
// updateTreeStructure (:446-528), updateClosestNodes (:86-124) and initializeID (:70-84).
//
// The reference runs per tip: one O(N) kernel over all 4N-4 slots, a Thrust reduction with a
// device->host copy, and two single-thread kernels.  Here per tip: one scan kernel over the 4i-4
// live slots (block-level argmin partials) and one single-wave kernel that finishes the argmin,
// splits the edge and runs the closest-list BFS frontier-parallel (in a tree every directed edge is
// reached at most once per BFS, so the insertions are independent of the visiting order).
// Distance rows are produced in batches by the row providers (msa.hip / mash.hip).
//
// State layout mirrors the reference (forward-star adjacency head/e/nxt/belong/len, K=5 sorted
// closest lists per directed edge) plus rev[slot] = reverse slot, which replaces the list walk of
// src/placement_close_k.cu:339-340.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nj_dev.hpp"

namespace dpr {

constexpr int K5 = 5;

struct PlacePartial { double add; int32_t idx; int32_t eid; double frac; int32_t rev; int32_t pad; };   // rev = reverse slot of eid (-1: look it up)

__global__ __launch_bounds__(kThreads) void place_init_kernel(PlaceBuffers p, int64_t lim, int64_t nodes)
{
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx < lim) {
        for (int i = 0; i < K5; ++i) { p.cdis[idx * K5 + i] = 2; p.cid[idx * K5 + i] = -1; }
        p.nxt[idx] = -1; p.e[idx] = -1; p.belong[idx] = -1; p.rev[idx] = -1;
    }
    if (idx < nodes) p.head[idx] = -1;
}

// lists only (backbone import keeps the adjacency): initializeID
__global__ __launch_bounds__(kThreads) void place_init_lists_kernel(PlaceBuffers p, int64_t lim)
{
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx < lim)
        for (int i = 0; i < K5; ++i) { p.cdis[idx * K5 + i] = 2; p.cid[idx * K5 + i] = -1; }
}

// rev[] for an imported adjacency: slots come in (child->parent, parent->child) pairs
__global__ __launch_bounds__(kThreads) void place_pair_rev_kernel(PlaceBuffers p, int64_t nslots)
{
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx < nslots) p.rev[idx] = (int32_t)(idx ^ 1);
}

__device__ __forceinline__ void list_insert(double* cdis, int32_t* cid, int slot, int x, double d, bool& inserted)
{
    inserted = false;
    for (int j = 0; j < K5; ++j) {
        const double nowd = cdis[slot * K5 + j];
        if (nowd > d) {
            for (int k = K5 - 1; k > j; --k) {
                cdis[slot * K5 + k] = cdis[slot * K5 + k - 1];
                cid[slot * K5 + k] = cid[slot * K5 + k - 1];
            }
            cdis[slot * K5 + j] = d;
            cid[slot * K5 + j] = x;
            inserted = true;
            break;
        }
    }
}

// ---- edge records (PlaceBuffers::er_d / er_i / eidx, see dpr_internal.hpp) ----------------------------------------
// the list of one side of an edge record; ex = eidx[slot] = 2 * edge + side
__device__ __forceinline__ void er_write_side(const PlaceBuffers& p, int ex, const double* cd, const int* ci)
{
    const int64_t k = ex >> 1;
    const int o = (ex & 1) * K5;
#pragma unroll
    for (int i = 0; i < K5; ++i) { p.er_d[(int64_t)(o + i) * p.ecap + k] = cd[i]; p.er_i[(int64_t)(o + i) * p.ecap + k] = ci[i]; }
}
// a whole record: edge k = (slot s0 with its list, slot s1 with its list), length
__device__ __forceinline__ void er_write_edge(const PlaceBuffers& p, int k, int s0, const double* cd0, const int* ci0, int s1, const double* cd1, const int* ci1, double len)
{
    er_write_side(p, 2 * k, cd0, ci0);
    er_write_side(p, 2 * k + 1, cd1, ci1);
    p.er_d[(int64_t)10 * p.ecap + k] = len;
    p.er_i[(int64_t)10 * p.ecap + k] = s0; p.er_i[(int64_t)11 * p.ecap + k] = s1;
    p.eidx[s0] = 2 * k; p.eidx[s1] = 2 * k + 1;
}

// (round 6: when a four-tip launch's earlier tips change more slots than the set holds -- or more than kRescanCap blocks must be
//  re-scanned -- the launch evaluates all 2.2 M slots of a 500 000-tip tree itself: milliseconds.  Counted in misc[2] / misc[3]
//  (dpr_get_place_walks); 0 of 50 000 queries on 500 000 tips, whose largest walk reaches 512 slots.)
constexpr int kDirtyHash = 4096, kDirtyCap = 2048, kRescanCap = 128;
struct DirtySet {          // slots whose evaluation inputs changed since the scan launch (LDS)
    int* hash;             // [kDirtyHash] open addressing, -1 = empty
    int* list;             // [kDirtyCap]
    int* count;            // entries in list (may run past kDirtyCap: overflow)
};
__device__ __forceinline__ void dirty_add(const DirtySet& ds, int slot)
{
    if (*ds.count >= kDirtyCap) { atomicAdd(ds.count, 1); return; }      // overflow: the caller falls back to a full scan
    unsigned h = ((unsigned)slot * 2654435761u) >> 20;
    for (int probe = 0; probe < kDirtyHash; ++probe) {
        const int old = atomicCAS(&ds.hash[h], -1, slot);
        if (old == slot) return;
        if (old == -1) {
            const int k = atomicAdd(ds.count, 1);
            if (k < kDirtyCap) ds.list[k] = slot;
            return;
        }
        h = (h + 1) & (kDirtyHash - 1);
    }
}
// two slots at once (a slot and its reverse): the two probes in flight together, one addition to the count
__device__ __forceinline__ void dirty_add2(const DirtySet& ds, int a, int b)
{
    if (*ds.count >= kDirtyCap) { atomicAdd(ds.count, 2); return; }       // overflow: the caller falls back to a full scan
    unsigned ha = ((unsigned)a * 2654435761u) >> 20, hb = ((unsigned)b * 2654435761u) >> 20;
    bool doa = true, dob = true, newa = false, newb = false;
    for (int probe = 0; probe < kDirtyHash && (doa || dob); ++probe) {
        int olda = 0, oldb = 0;
        if (doa) olda = atomicCAS(&ds.hash[ha], -1, a);
        if (dob) oldb = atomicCAS(&ds.hash[hb], -1, b);
        if (doa) { if (olda == a) doa = false; else if (olda == -1) { doa = false; newa = true; } else ha = (ha + 1) & (kDirtyHash - 1); }
        if (dob) { if (oldb == b) dob = false; else if (oldb == -1) { dob = false; newb = true; } else hb = (hb + 1) & (kDirtyHash - 1); }
    }
    const int n = (newa ? 1 : 0) + (newb ? 1 : 0);
    if (n) {
        int k = atomicAdd(ds.count, n);
        if (newa) { if (k < kDirtyCap) ds.list[k] = a; ++k; }
        if (newb && k < kDirtyCap) ds.list[k] = b;
    }
}
__device__ __forceinline__ bool dirty_has(const DirtySet& ds, int slot)
{
    unsigned h = ((unsigned)slot * 2654435761u) >> 20;
    for (int probe = 0; probe < kDirtyHash; ++probe) {
        const int v = ds.hash[h];
        if (v == slot) return true;
        if (v == -1) return false;
        h = (h + 1) & (kDirtyHash - 1);
    }
    return false;
}

// updateClosestNodes, one wave: frontier entries l..r processed 64 at a time.  The frontier holds SLOTS:
// reaching slot i = (u -> v) with the distance d of u inserts the new leaf into list[i]; if it entered, the
// slots leaving v other than the reverse of i (cont[2i], cont[2i+1]; write-once except at a split) follow
// with d + len[i].  That is the reference's node BFS with the adjacency walk folded into the queue entry.
// A round costs ONE global-memory hop (everything about the slot, all loads in flight together): the
// frontier lives in LDS (entries beyond kQueueLds spill to the global queue), its append offsets come from
// two ballots (a slot contributes 0, 1 or 2 entries), and -- every directed edge being reached at most once
// per BFS -- no list is read after it was written, so the rounds need no memory fence beyond the wave's own
// program order.  cont[2i] == -2 marks a target node of degree > 3 (possible only in an imported backbone),
// which falls back to walking that node's list.
constexpr int kQueueLds = 2048;
// (the frontier starts with the `ns` entries (slot, distance) the caller has put into the LDS queue -- one for a BFS from the new
//  leaf's slot; the split applies the first two rounds in registers and leaves the entries of round 2, place_split_wave)
// (kRec, the multi-tip launch: every slot whose list changed and its reverse are recorded in the block's dirty set)
// The first round after the split, loaded AHEAD by the split (place_split_wave): the slots behind x and y that round 2 can reach are known
// as soon as the split's own loads are back, two list insertions before it is known whether the leaf gets that far.  Lane k < 4
// holds candidate k (behind e0: oy0, oy1; behind e1: ox0, ox1) with everything a round reads about a slot; the round trip runs
// beside the split's arithmetic and stores instead of after them (one dependent hop less per tip).
struct BfsPre {
    int sl;               // candidate slot (-1: none)
    bool active;          // the leaf entered the list in front of it: the slot is on the frontier, reached with distance d
    double d;
    double cd[K5], ln;
    int ci[K5], k0, k1, ex, back;
};
__device__ __forceinline__ void bfs_pre_load(const PlaceBuffers& p, int sl, BfsPre& q)
{
    q.sl = sl; q.active = false; q.d = 0.0; q.ln = 0.0; q.k0 = -1; q.k1 = -1; q.ex = 0; q.back = -1;
#pragma unroll
    for (int j = 0; j < K5; ++j) { q.cd[j] = 0.0; q.ci[j] = 0; }
    if (sl >= 0) {
#pragma unroll
        for (int j = 0; j < K5; ++j) { q.cd[j] = p.cdis[sl * K5 + j]; q.ci[j] = p.cid[sl * K5 + j]; }
        q.ln = p.len[sl];
        q.k0 = p.cont[2 * sl]; q.k1 = p.cont[2 * sl + 1];
        q.ex = p.eidx[sl];
        q.back = p.rev[sl];
    }
}

// returns the number of slots the walk reached (queue entries; the two rounds the split applies in registers are not in it)
template <bool kRec>
__device__ __forceinline__ int closest_update_wave_impl(const PlaceBuffers& p, int x, int ns, const DirtySet& ds, int32_t* sq_id, double* sq_dis,
                                                        const BfsPre* pre = nullptr)
{
    const int lane = threadIdx.x & 63;
    int l = 0, r = ns;  // queue [l, r): the first ns entries are in sq_id / sq_dis (stored by a lane of this wavefront)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bool first = pre != nullptr;            // (wave-uniform) the round whose slots and records the split has loaded
    while (first || l < r) {
        const int cnt = first ? 0 : min(64, r - l);
        int sl = -1;
        double d = 0.0;
        if (first) {
            if (pre->active) { sl = pre->sl; d = pre->d; }
        } else if (lane < cnt) {
            const int qi = l + lane;
            if (qi < kQueueLds) { sl = sq_id[qi]; d = sq_dis[qi]; }
            else { sl = p.q_id[qi]; d = p.q_dis[qi]; }
        }
        int c0 = -1, c1 = -1, nnew = 0;
        double dn = 0.0;
        bool walk = false;
        if (sl >= 0) {
            double cd[K5], ln;
            int ci[K5], k0, k1, ex, back = -1;
            if (first) {
#pragma unroll
                for (int j = 0; j < K5; ++j) { cd[j] = pre->cd[j]; ci[j] = pre->ci[j]; }
                ln = pre->ln; k0 = pre->k0; k1 = pre->k1; ex = pre->ex; back = pre->back;
            } else {
#pragma unroll
                for (int j = 0; j < K5; ++j) { cd[j] = p.cdis[sl * K5 + j]; ci[j] = p.cid[sl * K5 + j]; }
                ln = p.len[sl];
                k0 = p.cont[2 * sl]; k1 = p.cont[2 * sl + 1];
                ex = p.eidx[sl];
                if (kRec) back = p.rev[sl];
            }
            int j = K5;
#pragma unroll
            for (int t = K5 - 1; t >= 0; --t)
                if (cd[t] > d) j = t;                  // first entry farther than d
            if (j < K5) {
#pragma unroll
                for (int t = K5 - 1; t > 0; --t)
                    if (t > j) { p.cdis[sl * K5 + t] = cd[t - 1]; p.cid[sl * K5 + t] = ci[t - 1]; }
                p.cdis[sl * K5 + j] = d;
                p.cid[sl * K5 + j] = x;
                {   // the same list into the slot's edge record (the scan of the next tip reads it there)
                    double nd[K5]; int ni[K5];
#pragma unroll
                    for (int t = 0; t < K5; ++t) { nd[t] = t < j ? cd[t] : (t == j ? d : cd[t > 0 ? t - 1 : 0]); ni[t] = t < j ? ci[t] : (t == j ? x : ci[t > 0 ? t - 1 : 0]); }
                    er_write_side(p, ex, nd, ni);
                }
                if (kRec) dirty_add2(ds, sl, back);
                dn = d + ln;
                if (k0 == -2) {                        // high-degree target: count its other slots
                    walk = true;
                    const int back = p.rev[sl];
                    for (int i = p.head[p.e[sl]]; i != -1; i = p.nxt[i]) nnew += (i != back) ? 1 : 0;
                } else {
                    c0 = k0; c1 = k1;
                    nnew = (c0 >= 0 ? 1 : 0) + (c1 >= 0 ? 1 : 0);
                }
            }
        }
        // append in lane order (order is irrelevant for the result, kept deterministic anyway)
        int excl, total;
        if (__builtin_amdgcn_ballot_w64(walk) == 0ull) {       // wave-uniform: every lane adds 0, 1 or 2 entries
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(nnew >= 1), m2 = __builtin_amdgcn_ballot_w64(nnew >= 2);
            const unsigned long long below = (1ull << lane) - 1ull;
            excl = __popcll(m1 & below) + __popcll(m2 & below);
            total = __popcll(m1) + __popcll(m2);
        } else {
            int incl = nnew;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            total = __shfl(incl, 63, 64);
            excl = incl - nnew;
        }
        const bool spill = r + total > kQueueLds;               // wave-uniform
        if (nnew) {
            int w = r + excl;
            auto push = [&](int slot) {
                if (w < kQueueLds) { sq_id[w] = slot; sq_dis[w] = dn; }
                else { p.q_id[w] = slot; p.q_dis[w] = dn; }
                ++w;
            };
            if (!walk) {
                if (c0 >= 0) push(c0);
                if (c1 >= 0) push(c1);
            } else {
                const int back = p.rev[sl];
                for (int i = p.head[p.e[sl]]; i != -1; i = p.nxt[i])
                    if (i != back) push(i);
            }
        }
        l += cnt;
        r += total;
        first = false;
        if (spill) {                                            // entries went to the global queue
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    return r;
}

__device__ __forceinline__ void closest_update_wave(const PlaceBuffers& p, int x, int start_slot)
{
    __shared__ int32_t sq_id[kQueueLds];
    __shared__ double sq_dis[kQueueLds];
    if ((threadIdx.x & 63) == 0) { sq_id[0] = start_slot; sq_dis[0] = 0.0; }
    closest_update_wave_impl<false>(p, x, 1, DirtySet{ nullptr, nullptr, nullptr }, sq_id, sq_dis);
}

// ---- the split, by one wavefront with the five list entries of a slot held one per LANE (lanes 0..4) -----------------
// updateTreeStructure (src/placement_close_k.cu:430-527) was, through round 5's first half, the work of lane 0 alone: ~800
// dependent instructions of list arithmetic and ~250 stores, 2.5 us of the 3.6 us the split took.  Here a list is a (double, int)
// pair per lane: loads, inherits, inserts and stores of a list are one or two instructions for the wave, the four new slots'
// scalars go out one slot per lane, and the merge of the two inherited lists is a rank computation over 15 lanes.
//
// insert (d, leaf) before the first entry strictly farther (list_insert); false if no entry is
__device__ __forceinline__ bool lane_list_insert(double& cd, int& ci, double d, int leaf, int lane)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(lane < K5 && cd > d);
    if (m == 0ull) return false;
    const int j = (int)__builtin_ctzll(m);
    const double pd = __shfl_up(cd, 1, 64);
    const int pi = __shfl_up(ci, 1, 64);
    if (lane == j) { cd = d; ci = leaf; }
    else if (lane > j) { cd = pd; ci = pi; }
    return true;
}
__device__ __forceinline__ void lane_list_store(const PlaceBuffers& p, int slot, double cd, int ci, int lane)
{
    if (lane < K5) { p.cdis[slot * K5 + lane] = cd; p.cid[slot * K5 + lane] = ci; }
}
__device__ __forceinline__ void lane_er_write_side(const PlaceBuffers& p, int ex, double cd, int ci, int lane)
{
    const int64_t k = ex >> 1;
    const int o = (ex & 1) * K5;
    if (lane < K5) { p.er_d[(int64_t)(o + lane) * p.ecap + k] = cd; p.er_i[(int64_t)(o + lane) * p.ecap + k] = ci; }
}

// Splits edge slot `eid` (reverse slot brev, -1: look it up) at fracLen for tip `num` with pendant length addLen; ec0 = 4*num-4 is
// the first of the four new slots e0 = middle->x, e1 = middle->y, e2 = outside->middle, e3 = middle->outside.  i0..i3 = what
// those slots' lists hold now (lane-distributed; slots the reference never touched keep the init values 2 / -1, which it reads
// back here).  Also applies rounds 0 and 1 of the new leaf's closest-list BFS while the lists are in registers (two dependent
// global round trips less per tip): round 0 reaches e2 with distance 0, round 1 -- through cont[e2] = (e0, e1) with distance
// len[e2] = addLen -- e0 and e1; a slot the leaf enters passes it on to its continuation slots (oy0, oy1 behind e0; ox0, ox1
// behind e1) with distance + length: same insertion rule and the same additions in the same order as closest_update_wave.
// Returns the number of frontier slots of round 2 (<= 4; lanes 0..3 of wavefront 0 hold them in `pre`, records included), or -1 when a node of degree > 3 lies behind x or y
// (imported backbone: the BFS then starts at e2 and walks).  Everything returned is wave-uniform.
// TWO wavefronts share the work (role = wavefront index 0 / 1, both called with the same arguments): wavefront 0 takes what the
// closest-list BFS it runs next depends on -- the inherited lists of e0 / e1, rounds 0-1 of the BFS, their stores and record
// sides -- and wavefront 1 the bookkeeping nothing in that BFS reads (the slots' scalars, continuation slots, the merged list of
// e3, the record scalars and the sides of xe / ye / e3): 1.4 us of dependent instructions off the tip's critical path.  The BFS
// only reaches slots beyond x and y, none of which the split writes -- except in the degree > 3 fallback, where it starts at e2
// and reads the new slots: there wavefront 0 does everything itself and wavefront 1 nothing.  Disjoint stores; both end before
// the next kernel / the block barrier of the multi-tip kernel.  Loads against the other wavefront's stores: wavefront 1 overwrites
// what both read here (len / e / cont / eidx of the split edge's slots), so it stores only after wavefront 0 has its loads back
// (*sync == tip, an LDS word the caller initialises to -1 before a block barrier); wavefront 0 overwrites the lists of e0..e2,
// which both read as i0..i2: the caller loads those BEFORE the block barrier that publishes the winner.
__device__ __forceinline__ int place_split_wave(const PlaceBuffers& p, int64_t num, int ec0, int eid, int brev, double fracLen, double addLen,
                                                double i0d, int i0i, double i1d, int i1i, double i2d, int i2i, double i3d, int i3i,
                                                int mininel, int role, BfsPre& pre, int* sync, int& xe_out, int& ye_out)
{
    const int lane = threadIdx.x & 63;
    const int li = lane < K5 ? lane : 0;
    const int placeId = (int)num, N = (int)p.N;
    const int middle = placeId + N - 1, outside = placeId;
    const int xe = eid, ye = brev >= 0 ? brev : p.rev[eid];   // the reference finds them by walking head[x] / head[y]
    // every load of the split in one round trip, before the first store (same addresses for all lanes but the list entries)
    const int x = p.belong[eid], y = p.e[eid];
    const double originalDis = p.len[eid];
    const double cdx = p.cdis[xe * K5 + li], cdy = p.cdis[ye * K5 + li];
    const int cix = p.cid[xe * K5 + li], ciy = p.cid[ye * K5 + li];
    const double lenye = p.len[ye];
    const int ox0 = p.cont[2 * xe], ox1 = p.cont[2 * xe + 1], oy0 = p.cont[2 * ye], oy1 = p.cont[2 * ye + 1];
    const int kold = p.eidx[xe] >> 1;                  // the record of the edge being split: taken over by (middle, x)
    const int e0 = ec0, e1 = ec0 + 1, e2 = ec0 + 2, e3 = ec0 + 3;
    const int nedge = (int)(2 * num - 2);              // records of the two new edges: nedge = (middle, y), nedge + 1 = (middle, outside)
    const double len1 = originalDis - fracLen;
    const bool fallback = ox0 == -2 || oy0 == -2;
    const bool lists = role == 0, book = fallback ? role == 0 : role == 1;
    if (role == 0) {
        // every load above has returned -> wavefront 1 may overwrite what they read.  (Release: the compiler must not sink a load
        // below the store; the waitcnt: the hardware must have the data back, not just the requests out.)
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) __hip_atomic_store(sync, (int)num, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (book) {
        while (__hip_atomic_load(sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (int)num) __builtin_amdgcn_s_sleep(1);
    }
    // round 2 of the BFS ahead of time (wavefront 0, lanes 0..3): in flight during everything below
    bfs_pre_load(p, (lists && !fallback && lane < 4) ? (lane == 0 ? oy0 : lane == 1 ? oy1 : lane == 2 ? ox0 : ox1) : -1, pre);
    // middle -> x inherits the list of y -> x, middle -> y the list of x -> y (entry by entry; an empty entry keeps what the slot held)
    const bool has0 = ciy != -1, has1 = cix != -1;
    double n0d = has0 ? cdy + originalDis - fracLen : i0d, n1d = has1 ? cdx + fracLen : i1d;
    int n0i = has0 ? ciy : i0i, n1i = has1 ? cix : i1i;
    if (book) {
        // the four new slots, one per lane: e0 = middle -> x, e1 = middle -> y, e2 = outside -> middle, e3 = middle -> outside
        if (lane < 4) {
            const int s = ec0 + lane;
            p.e[s] = lane == 0 ? x : lane == 1 ? y : lane == 2 ? middle : outside;
            p.len[s] = lane == 0 ? fracLen : lane == 1 ? len1 : addLen;
            p.nxt[s] = lane == 1 ? e0 : lane == 3 ? e1 : -1;
            p.belong[s] = lane == 2 ? outside : middle;
            p.rev[s] = lane == 0 ? xe : lane == 1 ? ye : lane == 2 ? e3 : e2;
            p.eidx[s] = lane == 0 ? 2 * kold : lane == 1 ? 2 * nedge : lane == 2 ? 2 * (nedge + 1) + 1 : 2 * (nedge + 1);
        }
        // continuation slots: the new slots towards x / y inherit what lay beyond y -> x / x -> y; e2 leads to e0 and e1, e3 ends at
        // the new leaf; slots entering x or y from elsewhere keep theirs (slot ids do not change)
        if (lane < 8)
            p.cont[2 * ec0 + lane] = lane == 0 ? oy0 : lane == 1 ? oy1 : lane == 2 ? ox0 : lane == 3 ? ox1 : lane == 4 ? e0 : lane == 5 ? e1 : -1;
        // the two slots of the split edge now end in `middle`
        if (lane < 2) {
            const int s = lane == 0 ? xe : ye;
            p.e[s] = middle;
            p.len[s] = lane == 0 ? fracLen : lenye - fracLen;
            p.rev[s] = ec0 + lane;
            p.cont[2 * s] = lane == 0 ? e1 : e0; p.cont[2 * s + 1] = e3;
            p.eidx[s] = lane == 0 ? 2 * kold + 1 : 2 * nedge + 1;
        }
        // edge records (scalars): `middle` has the largest node id, so the slots leaving it are the evaluated sides
        if (lane < 3) {
            const int k = lane == 0 ? kold : nedge + lane - 1;
            p.er_d[(int64_t)10 * p.ecap + k] = lane == 0 ? fracLen : lane == 1 ? len1 : addLen;
            p.er_i[(int64_t)10 * p.ecap + k] = lane == 0 ? e0 : lane == 1 ? e1 : e3;
            p.er_i[(int64_t)11 * p.ecap + k] = lane == 0 ? xe : lane == 1 ? ye : e2;
        }
        if (lane == 0) {
            p.head[outside] = e2; p.head[middle] = e3;
            const int lo = xe < ye ? xe : ye;                  // xe, ye (and e2) now have belong < e
            if (lo < mininel) p.misc[0] = lo;
        }
        // middle -> outside: the slot's list, then the entries of e1's list up to its first empty one, then e0's likewise, each inserted
        // before the first entry strictly farther, five kept (src/placement_close_k.cu:506-527).  An entry dropped at one insertion never
        // comes back, so that is the first five of the 5 + a + b candidates in order of (distance, order of insertion): ranks over lanes
        // 0..4 (slot's list), 5..9 (e1's), 10..14 (e0's)
        double md;
        int mi;
        {
            const unsigned long long m1 = __builtin_amdgcn_ballot_w64(lane < K5 && n1i != -1), m0 = __builtin_amdgcn_ballot_w64(lane < K5 && n0i != -1);
            const int a = (int)__builtin_ctzll(~m1), b = (int)__builtin_ctzll(~m0);       // lengths of the non-empty prefixes
            const int src = lane < K5 ? lane : lane < 2 * K5 ? lane - K5 : lane - 2 * K5;
            const double s1d = __shfl(n1d, src, 64), s0d = __shfl(n0d, src, 64);
            const int s1i = __shfl(n1i, src, 64), s0i = __shfl(n0i, src, 64);
            const double cd = lane < K5 ? i3d : lane < 2 * K5 ? s1d : s0d;
            const int ci = lane < K5 ? i3i : lane < 2 * K5 ? s1i : s0i;
            const unsigned vmask = 0x1fu | (((1u << a) - 1u) << K5) | (((1u << b) - 1u) << (2 * K5));
            const bool valid = lane < 3 * K5 && ((vmask >> lane) & 1u);
            int rank = 0;
    #pragma unroll
            for (int k = 0; k < 3 * K5; ++k) {
                const double dk = readlane_f64(cd, k);
                const bool before = dk < cd || (dk == cd && k < lane);
                rank += (((vmask >> k) & 1u) && before) ? 1 : 0;
            }
            md = 0.0; mi = 0;
    #pragma unroll
            for (int r = 0; r < K5; ++r) {
                const unsigned long long mr = __builtin_amdgcn_ballot_w64(valid && rank == r);
                const int from = mr ? (int)__builtin_ctzll(mr) : r;      // (exactly one lane unless a distance is NaN)
                const double vd = readlane_f64(cd, from);
                const int vi = __builtin_amdgcn_readlane(ci, from);
                if (lane == r) { md = vd; mi = vi; }
            }
        }
        lane_list_store(p, e3, md, mi, lane);
        lane_er_write_side(p, 2 * kold + 1, cdx, cix, lane);
        lane_er_write_side(p, 2 * nedge + 1, cdy, ciy, lane);
        lane_er_write_side(p, 2 * (nedge + 1), md, mi, lane);
    }
    // rounds 0 and 1 of the closest-list BFS (not behind a node of degree > 3: the walk form stays in the BFS)
    int ns = -1;
    if (lists) {
        if (!fallback) {
            bool in0 = false, in1 = false;
            double dn0 = 0.0, dn1 = 0.0;
            if (lane_list_insert(i2d, i2i, 0.0, placeId, lane)) {
                const double d1 = 0.0 + addLen;             // (d + len[e2], as the BFS computes it)
                in0 = lane_list_insert(n0d, n0i, d1, placeId, lane);
                dn0 = d1 + fracLen;                         // len[e0]
                in1 = lane_list_insert(n1d, n1i, d1, placeId, lane);
                dn1 = d1 + len1;                            // len[e1]
            }
            pre.active = pre.sl >= 0 && (lane < 2 ? in0 : in1);
            pre.d = lane < 2 ? dn0 : dn1;
            ns = __popcll(__builtin_amdgcn_ballot_w64(pre.active));
            lane_list_store(p, e2, i2d, i2i, lane);
        }
        lane_list_store(p, e0, n0d, n0i, lane);
        lane_list_store(p, e1, n1d, n1i, lane);
        // the same lists into the edge records (the scan of the next tip reads them there)
        lane_er_write_side(p, 2 * kold, n0d, n0i, lane);
        lane_er_write_side(p, 2 * nedge, n1d, n1i, lane);
        lane_er_write_side(p, 2 * (nedge + 1) + 1, i2d, i2i, lane);
    }
    xe_out = xe; ye_out = ye;
    return ns;
}

// buildInitialTree + the two closest updates; dis = distance row of tip 1 (entry 0)
__global__ __launch_bounds__(64) void place_initial_tree_kernel(PlaceBuffers p, const double* __restrict__ dis)
{
    if (threadIdx.x == 0) {
        const int nv = (int)p.N;
        const double d = dis[0];
        int ec = 0;
        p.e[ec] = nv; p.len[ec] = d / 2; p.nxt[ec] = p.head[0]; p.head[0] = ec; p.belong[ec] = 0; p.rev[ec] = 2; ec++;
        p.e[ec] = nv; p.len[ec] = d / 2; p.nxt[ec] = p.head[1]; p.head[1] = ec; p.belong[ec] = 1; p.rev[ec] = 3; ec++;
        p.e[ec] = 0;  p.len[ec] = d / 2; p.nxt[ec] = p.head[nv]; p.head[nv] = ec; p.belong[ec] = nv; p.rev[ec] = 0; ec++;
        p.e[ec] = 1;  p.len[ec] = d / 2; p.nxt[ec] = p.head[nv]; p.head[nv] = ec; p.belong[ec] = nv; p.rev[ec] = 1; ec++;
        // continuation slots: beyond 0 -> nv lies nv -> 1, beyond 1 -> nv lies nv -> 0, leaves end the walk
        p.cont[0] = 3; p.cont[1] = -1; p.cont[2] = 2; p.cont[3] = -1;
        p.cont[4] = -1; p.cont[5] = -1; p.cont[6] = -1; p.cont[7] = -1;
        // edge records: (nv -> 0 | 0 -> nv), (nv -> 1 | 1 -> nv); the lists are still the initial ones (2 / -1)
        double l2[K5]; int lm[K5];
        for (int i = 0; i < K5; ++i) { l2[i] = 2; lm[i] = -1; }
        er_write_edge(p, 0, 2, l2, lm, 0, l2, lm, d / 2);
        er_write_edge(p, 1, 3, l2, lm, 1, l2, lm, d / 2);
        p.misc[0] = 0;                       // slot 0 (0 -> nv) is the smallest slot with belong < e, now and later
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    closest_update_wave(p, 0, 0);
    closest_update_wave(p, 1, 1);
}

// continuation slots of an imported backbone (target node of degree > 3 -> -2: list walk)
__global__ __launch_bounds__(kThreads) void place_build_cont_kernel(PlaceBuffers p, int64_t nslots)
{
    const int64_t sl = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (sl >= nslots) return;
    const int back = p.rev[sl];
    int c[2] = { -1, -1 }, cnt = 0;
    for (int i = p.head[p.e[sl]]; i != -1; i = p.nxt[i]) {
        if (i == back) continue;
        if (cnt < 2) c[cnt] = i;
        ++cnt;
    }
    if (cnt > 2) { c[0] = -2; c[1] = -2; }
    p.cont[2 * sl] = c[0]; p.cont[2 * sl + 1] = c[1];
}

// Closest lists of an imported backbone without m dependent launches, EXACTLY as the reference builds them.
// The reference inserts the leaves one after the other (m single-thread BFS launches, src/placement_close_k.cu:
// 247-260): leaf x, coming from slot w->u with distance d, is offered to list[u->v] (five entries, sorted, start
// value 2, an entry gives way only to a strictly smaller distance) and continues behind v only if it entered.  So the
// leaves ARRIVE at a slot in ascending leaf id, a leaf PASSES a slot iff it entered the list as it stood at its turn,
// and the arrivals of u->v are the passers of the slots w->u (w != v) with len[w->u] added -- a recurrence over the
// directed edges of the tree, which form a DAG (depth = tree diameter).  place_lists_level_kernel evaluates it level
// by level: a slot whose feeding slots are done merges their passer sequences by leaf id, replays the five-entry
// insertion over the merged arrivals, keeps the leaves that entered (its own passer sequence) and writes the final list.  Same lists as the serial order bit for bit, including the ties that floating-point rounding
// of d + len creates downstream of distinct distances (a plain "five smallest by (distance, id)" relaxation -- round 1's
// first version -- gets those wrong: a leaf evicted upstream can tie with its evictors after the addition and, having
// arrived first, stay ahead of them; found by tests/test_gpu_fullsize.py::test_config4_add_50k_onto_500k).
// Passer sequences have very different lengths (a handful near the leaves, a few hundred for slots that have most of the
// tree behind them), so they live in one pool: a slot replays its arrivals twice, first to count, then -- after reserving its range with one
// atomic add -- to store.
__global__ __launch_bounds__(kThreads) void place_lists_level_kernel(PlaceBuffers p, int32_t* __restrict__ level,
                                                                     int32_t* __restrict__ pcnt, long long* __restrict__ poff,
                                                                     int32_t* __restrict__ pid, double* __restrict__ pdis,
                                                                     unsigned long long* __restrict__ pool_top,
                                                                     unsigned long long pool_cap, int round, int64_t nslots,
                                                                     int64_t m, int* __restrict__ flags)
{
    const int64_t s = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (s >= nslots || level[s] >= 0) return;
    const int u = p.belong[s];
    // feeding slots w -> u (all but the reverse of s): done in an EARLIER round?
    constexpr int kMaxIn = 8;
    int in[kMaxIn], nin = 0;
    bool ready = true, many = false;
    for (int o = p.head[u]; o != -1; o = p.nxt[o]) {
        if (o == (int)s) continue;
        const int w = p.rev[o];
        const int lv = level[w];
        if (lv < 0 || lv >= round) ready = false;
        if (nin < kMaxIn) in[nin++] = w; else many = true;
    }
    if (!ready) return;
    if (many) { flags[1] = 1; return; }                 // node of degree > 9: leave it to the serial order
    int cn[kMaxIn];
    long long of[kMaxIn];
    double ln[kMaxIn];
    for (int k = 0; k < nin; ++k) { cn[k] = pcnt[in[k]]; of[k] = poff[in[k]]; ln[k] = p.len[in[k]]; }
    double bd[K5];
    int bi[K5];
    // merge of the feeding passer sequences (each ascending in leaf id) with the leaf u itself (a backbone leaf has no
    // feeding slot; its own arrival has distance 0), replaying list_insert; out != nullptr: the passers are stored
    auto replay = [&](int32_t* out_id, double* out_d) -> int {
#pragma unroll
        for (int j = 0; j < K5; ++j) { bd[j] = 2.0; bi[j] = -1; }
        int npass = 0;
        int hd[kMaxIn];
        for (int k = 0; k < nin; ++k) hd[k] = 0;
        bool self = u < m;
        for (;;) {
            int best = -1, bx = 0x7fffffff;
            for (int k = 0; k < nin; ++k)
                if (hd[k] < cn[k]) {
                    const int x = pid[of[k] + hd[k]];
                    if (x < bx) { bx = x; best = k; }
                }
            int x;
            double d;
            if (self && u < bx) { x = u; d = 0.0; self = false; }
            else if (best < 0) break;
            else { x = bx; d = pdis[of[best] + hd[best]] + ln[best]; ++hd[best]; }
            int pos = K5;
#pragma unroll
            for (int j = K5 - 1; j >= 0; --j)
                if (bd[j] > d) pos = j;                   // first entry strictly farther (list_insert)
            if (pos == K5) continue;
#pragma unroll
            for (int j = K5 - 1; j > 0; --j)
                if (j > pos) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; }
#pragma unroll
            for (int j = 0; j < K5; ++j)
                if (j == pos) { bd[j] = d; bi[j] = x; }
            if (out_id) { out_id[npass] = x; out_d[npass] = d; }
            ++npass;
        }
        return npass;
    };
    const int npass = replay(nullptr, nullptr);
    const unsigned long long off = atomicAdd(pool_top, (unsigned long long)npass);
    if (off + (unsigned long long)npass > pool_cap) { flags[1] = 1; return; }      // pool exhausted: the serial order takes over
    (void)replay(pid + off, pdis + off);
    pcnt[s] = npass;
    poff[s] = (long long)off;
#pragma unroll
    for (int j = 0; j < K5; ++j) { p.cid[s * K5 + j] = bi[j]; p.cdis[s * K5 + j] = bd[j]; }
    level[s] = round;
    flags[0] = 1;                                         // progress
}

// closest lists of an imported backbone: leaves 0..m-1 in order (src/placement_close_k.cu:247-260)
__global__ __launch_bounds__(64) void place_backbone_lists_kernel(PlaceBuffers p, int64_t t0, int64_t t1)
{
    for (int64_t t = t0; t < t1; ++t) closest_update_wave(p, (int)t, p.head[t]);   // a leaf has one slot
}

// Per tip two kernels: place_tip_edges_kernel (calculateBranchLength for the live edges and the
// block-level first minimum) and the one-workgroup place_update_kernel, which finishes the argmin (four
// wavefronts), splits the edge (updateTreeStructure; two wavefronts, place_split_wave) and runs the
// closest-list update (updateClosestNodes; one wavefront, its first round loaded ahead by the split) -- the
// reference's Thrust reduction, device->host copy and two single-thread kernels.  (Fusing the two with a
// last-block-done ticket was measured 1.5-2x SLOWER: every block then needs a device-scope release
// fence, i.e. an L2 write-back, which costs more than the kernel boundary it saves.)
constexpr int kUpdThreads = 256;
// the block's winner (smallest pendant length, then smallest slot) from the per-wavefront winners in LDS: lane w takes wavefront w's,
// one wave reduction -- a loop over the entries is nw dependent LDS round trips (1.6 us for the 16 wavefronts of the multi-tip launch)
__device__ __forceinline__ void place_block_winner(const double* s_add, const int* s_idx, const int* s_eid, const double* s_frac, const int* s_rev, int nw,
                                                   double& badd, int& bidx, int& beid, double& bfrac, int& brev)
{
    const int lane = threadIdx.x & 63;
    double a = __builtin_inf(), fr = 0.0;
    int ix = 0x7fffffff, ei = 0, rv = -1;
    if (lane < nw) { a = s_add[lane]; ix = s_idx[lane]; ei = s_eid[lane]; fr = s_frac[lane]; rv = s_rev[lane]; }
    const double wa = wave_fmin(a);
    const uint64_t wi = wave_umin64(a == wa ? (uint64_t)(uint32_t)ix : ~0ull);
    const unsigned long long own = __builtin_amdgcn_ballot_w64((a == wa) & ((uint64_t)(uint32_t)ix == wi));
    const int src = (int)__builtin_ctzll(own);
    badd = readlane_f64(a, src); bfrac = readlane_f64(fr, src);
    bidx = __builtin_amdgcn_readlane(ix, src); beid = __builtin_amdgcn_readlane(ei, src); brev = __builtin_amdgcn_readlane(rv, src);
}
__device__ __forceinline__ void place_finish_and_update(const PlaceBuffers& p, const PlacePartial* partials, int nparts,
                                                        int64_t num, int64_t edge_count, double* __restrict__ trace)
{
    __shared__ double s_add[kUpdThreads / 64], s_frac[kUpdThreads / 64];
    __shared__ int s_idx[kUpdThreads / 64], s_eid[kUpdThreads / 64], s_rev[kUpdThreads / 64];
    __shared__ int s_sync;                    // place_split_wave: wavefront 0's loads are back
    const int tid = threadIdx.x, lane = tid & 63;
    int ec = (int)edge_count;
    if (tid == 0) s_sync = -1;
    const unsigned long long tk0 = wall_clock64();
    // (a) loads that do not depend on the winner: the (initial) lists of the new slots ec, ec+1, ec+3 -- slots the
    // reference never touched keep the init values 2 / -1, which it reads back at the split; in flight during (b)
    double i0d = 2.0, i1d = 2.0, i2d = 2.0, i3d = 2.0;     // one list entry per lane (lanes 0..4 of wavefront 0)
    int i0i = -1, i1i = -1, i2i = -1, i3i = -1;
    int mininel = 0x7fffffff;
    if (tid < 128 && lane < K5) {          // (wavefronts 0 and 1 split the edge, place_split_wave)
        i0d = p.cdis[ec * K5 + lane]; i0i = p.cid[ec * K5 + lane];
        i1d = p.cdis[(ec + 1) * K5 + lane]; i1i = p.cid[(ec + 1) * K5 + lane];
        i2d = p.cdis[(ec + 2) * K5 + lane]; i2i = p.cid[(ec + 2) * K5 + lane];
        i3d = p.cdis[(ec + 3) * K5 + lane]; i3i = p.cid[(ec + 3) * K5 + lane];
    }
    if (tid < 128 && lane == 0) mininel = p.misc[0];
    // (b) first minimum over the block partials of the scan: four waves, eight loads in flight per thread
    double badd = __builtin_inf(), bfrac = 0;
    int bidx = 0x7fffffff, beid = 0, brev = -1;
    const int nthr = (int)blockDim.x;
    for (int i0 = tid; i0 < nparts; i0 += 8 * nthr) {
        PlacePartial pp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + nthr * u;
            if (i < nparts) pp[u] = partials[i];
            else { pp[u].add = __builtin_inf(); pp[u].idx = 0x7fffffff; pp[u].eid = 0; pp[u].frac = 0; pp[u].rev = -1; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (pp[u].add < badd || (pp[u].add == badd && pp[u].idx < bidx)) { badd = pp[u].add; bidx = pp[u].idx; beid = pp[u].eid; bfrac = pp[u].frac; brev = pp[u].rev; }
    }
    // slots >= 4*num-4 (and < 4M-4) all carry the tuple (0,0,2): the first of them competes -- and so does the first live slot
    // with belong < e, which the edge scan does not visit (same tuple; src/placement_close_k.cu:309-358 writes it for those)
    const int64_t live = 4 * num - 4, lim = 4 * p.M - 4;
    if (tid == 0) {
        if (live < lim && (2.0 < badd || (2.0 == badd && (int)live < bidx))) { badd = 2.0; bidx = (int)live; beid = 0; bfrac = 0; brev = -1; }
        if (mininel < (int)live && (2.0 < badd || (2.0 == badd && mininel < bidx))) { badd = 2.0; bidx = mininel; beid = 0; bfrac = 0; brev = -1; }
    }
    {   // wave winner (smallest add, then smallest idx; add is never NaN here), then the four wave winners through LDS
        const double wa = wave_fmin(badd);
        const uint64_t wi = wave_umin64(badd == wa ? (uint64_t)(uint32_t)bidx : ~0ull);
        const unsigned long long own = __builtin_amdgcn_ballot_w64((badd == wa) & ((uint64_t)(uint32_t)bidx == wi));
        const int src = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(own));
        if (lane == src) { s_add[tid >> 6] = badd; s_idx[tid >> 6] = bidx; s_eid[tid >> 6] = beid; s_frac[tid >> 6] = bfrac; s_rev[tid >> 6] = brev; }
    }
    __syncthreads();
    if (tid >= 128) return;                   // wavefronts 0 and 1 go on: the split is shared, the BFS is wavefront 0's
    const int role = __builtin_amdgcn_readfirstlane(tid >> 6);
    place_block_winner(s_add, s_idx, s_eid, s_frac, s_rev, nthr / 64, badd, bidx, beid, bfrac, brev);
    const int eid = beid;
    const double fracLen = bfrac, addLen = badd;
    const unsigned long long tk1 = wall_clock64();
    const int placeId = (int)num;
    if (tid == 0 && trace) { trace[3 * num] = eid; trace[3 * num + 1] = fracLen; trace[3 * num + 2] = addLen; }
    // (c) the split and rounds 0-1 of the closest-list BFS, lists one entry per lane
    __shared__ int32_t sq_id[kQueueLds];      // the BFS frontier (wavefront 0's)
    __shared__ double sq_dis[kQueueLds];
    int xe_, ye_;
    BfsPre pre;
    int bfs_ns = place_split_wave(p, num, ec, eid, brev, fracLen, addLen, i0d, i0i, i1d, i1i, i2d, i2i, i3d, i3i,
                                  __builtin_amdgcn_readfirstlane(mininel), role, pre, &s_sync, xe_, ye_);
    if (role != 0) return;
    // the wave reads what its lanes just stored: program order within the wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long tk2 = wall_clock64();
    if (bfs_ns < 0) {                       // (degree > 3 behind x or y) from the new leaf's only slot: outside -> middle
        if (lane == 0) { sq_id[0] = (int)edge_count + 2; sq_dis[0] = 0.0; }
        const int reached = closest_update_wave_impl<false>(p, placeId, 1, DirtySet{ nullptr, nullptr, nullptr }, sq_id, sq_dis);
        if (lane == 0) p.bfs_cnt[placeId] = -reached - 1;      // (negative: the walk of a node of degree > 3)
    } else if (bfs_ns > 0) {
        const int reached = closest_update_wave_impl<false>(p, placeId, 0, DirtySet{ nullptr, nullptr, nullptr }, sq_id, sq_dis, &pre);
        if (lane == 0) p.bfs_cnt[placeId] = reached;
    }
    if ((p.dbg & 4) && trace && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long tk3 = wall_clock64();
        trace[3 * num] = (double)(tk1 - tk0); trace[3 * num + 1] = (double)(tk2 - tk1); trace[3 * num + 2] = (double)(tk3 - tk2);
    }
}

constexpr int kTipThreads = 128;      // edges per block of the one-tip scan (100 000 tips: 128: 1 413 ms of tree kernels, 256: 1 449, 512: 1 584)
constexpr int kTipThreadsM = 256;     // ... of the four-tip scan (--add 50 000 onto 500 000: 128: 1 611 ms, 256: 1 554, 512: 1 716)
// The per-tip scan over the EDGE RECORDS (round 5): one thread per undirected edge instead of one per directed slot -- the slot
// scan of rounds 1-4 loaded every list twice (as the slot's own and as its reverse's) and kept half of its threads for slots
// with belong < e, whose tuple is the constant (0, 0, 2): 10.8 -> 6.7 us per launch at 100 000 tips.  Here a thread reads the 2 x 5 list entries, the length and the two slot ids
// of its edge in one round trip of coalesced loads (struct of arrays), then the ten distance gathers: two dependent hops, half
// the bytes, half the blocks.  Same arithmetic per edge, same key (pendant length, slot of the evaluated side); the constant
// tuple of the smallest slot with belong < e joins in place_finish_and_update.
__global__ __launch_bounds__(kTipThreads) void place_tip_edges_kernel(PlaceBuffers p, const double* __restrict__ dis,
                                                                   int64_t num, PlacePartial* __restrict__ partials)
{
    __builtin_amdgcn_s_setprio(3);      // chains of dependent steps: these waves go first where a distance kernel shares the SIMD
    __shared__ double sadd[kTipThreads / 64];
    __shared__ int sidx[kTipThreads / 64];
    const int64_t nedge = 2 * num - 2;
    const int64_t k = (int64_t)blockIdx.x * kTipThreads + threadIdx.x;
    const bool have = k < nedge;
    double add = 2.0, d1 = 0.0;
    int sl0 = 0x7fffffff, sl1 = -1;
    if (have) {
        double cd[2 * K5];
        int ci[2 * K5];
#pragma unroll
        for (int i = 0; i < 2 * K5; ++i) { cd[i] = p.er_d[(int64_t)i * p.ecap + k]; ci[i] = p.er_i[(int64_t)i * p.ecap + k]; }
        const double L = p.er_d[(int64_t)10 * p.ecap + k];
        sl0 = p.er_i[(int64_t)10 * p.ecap + k]; sl1 = p.er_i[(int64_t)11 * p.ecap + k];
        double dv[2 * K5];
#pragma unroll
        for (int i = 0; i < 2 * K5; ++i) dv[i] = ci[i] != -1 ? dis[ci[i]] : 0.0;
        double dis1 = 0, dis2 = 0, val;
#pragma unroll
        for (int i = 0; i < K5; ++i)
            if (ci[i] != -1) { val = dv[i] - cd[i]; if (val > dis1) dis1 = val; }
#pragma unroll
        for (int i = 0; i < K5; ++i)
            if (ci[K5 + i] != -1) { val = dv[K5 + i] - cd[K5 + i]; if (val > dis2) dis2 = val; }
        double a = (dis1 + dis2 - L) / 2;
        if (a < 0) a = 0;
        dis1 -= a; dis2 -= a;
        if (dis1 < 0) dis1 = 0;
        if (dis2 < 0) dis2 = 0;
        if (dis1 > L) { a += dis1 - L; dis1 = L; }
        if (dis2 > L) { a += dis2 - L; dis2 = L; }
        const double rest = L - dis1 - dis2;
        dis1 += rest / 2; dis2 += rest / 2;
        add = a; d1 = dis1;
    }
    // first minimum of add over the slot index (thrust::min_element): key (add, slot); NaN never wins
    double badd = (have && add == add) ? add : __builtin_inf();
    int bidx = have ? sl0 : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oa = __shfl_down(badd, off, 64);
        const int oi = __shfl_down(bidx, off, 64);
        if (oa < badd || (oa == badd && oi < bidx)) { badd = oa; bidx = oi; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sadd[w] = badd; sidx[w] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kTipThreads / 64; ++i)
            if (sadd[i] < badd || (sadd[i] == badd && sidx[i] < bidx)) { badd = sadd[i]; bidx = sidx[i]; }
        sadd[0] = badd; sidx[0] = bidx;
    }
    __syncthreads();
    if (have && sl0 == sidx[0]) {
        PlacePartial pp; pp.add = add; pp.idx = sl0; pp.eid = sl0; pp.frac = d1; pp.rev = sl1; pp.pad = 0;
        partials[blockIdx.x] = pp;
    }
    if (threadIdx.x == 0 && sidx[0] == 0x7fffffff) {
        PlacePartial pp; pp.add = __builtin_inf(); pp.idx = 0x7fffffff; pp.eid = 0; pp.frac = 0; pp.rev = -1; pp.pad = 0;
        partials[blockIdx.x] = pp;
    }
}

// edge records of an adjacency that was not built by splits (imported backbone): one record per slot with belong >= e
__global__ __launch_bounds__(kThreads) void place_pack_edges_kernel(PlaceBuffers p, int64_t nslots)
{
    const int64_t s = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (s >= nslots) return;
    const int bel = p.belong[s], tgt = p.e[s];
    if (tgt < 0) return;                      // (unused slot; dpr_place_run rejects backbones that leave any below 4m - 4)
    if (bel < tgt) { atomicMin(&p.misc[0], (int)s); return; }
    const int r = p.rev[s];
    const int k = atomicAdd(&p.misc[1], 1);
    double c0[K5], c1[K5];
    int i0[K5], i1[K5];
#pragma unroll
    for (int i = 0; i < K5; ++i) { c0[i] = p.cdis[s * K5 + i]; i0[i] = p.cid[s * K5 + i]; c1[i] = p.cdis[(int64_t)r * K5 + i]; i1[i] = p.cid[(int64_t)r * K5 + i]; }
    er_write_edge(p, k, (int)s, c0, i0, r, c1, i1, p.len[s]);
}

__global__ __launch_bounds__(kUpdThreads) void place_update_kernel(PlaceBuffers p, const PlacePartial* partials, int nparts,
                                                          int64_t num, double* __restrict__ trace)
{
    __builtin_amdgcn_s_setprio(3);      // chains of dependent steps: these waves go first where a distance kernel shares the SIMD
    place_finish_and_update(p, partials, nparts, num, 4 * num - 4, trace);
}

// ------------------------------------------------------------------------------------------------
// Several tips per launch (round 2).  The per-tip pair of launches costs ~7 us of scan + ~12 us of update, most of it
// launch ramp, kernel boundary and dependent round trips.  Here ONE scan launch evaluates every live slot for kMultiB
// consecutive tips against the state BEFORE the first of them (the slot's lists are loaded once), and ONE update launch
// places the tips one after the other.  For the 2nd..Bth tip the speculative block minima stay valid except where an
// earlier placement of the same launch changed an input of the evaluation: the lists of the slots the new leaf entered
// (and, since a slot's evaluation reads its reverse's list, their reverse slots), the split edge's two slots and the four
// new slots.  Those "dirty" slots (50-100 per placement) are re-evaluated with the current state; a block whose
// speculative winner is dirty is re-scanned; everything else keeps its speculative value, which is what a sequential
// scan would compute (same inputs, same arithmetic).  First minimum by (pendant length, slot) as before, so adjacency,
// lists and trace are those of the tip-by-tip schedule bit for bit (tests/test_gpu_mash_place.py, test_gpu_fullsize.py).
// ------------------------------------------------------------------------------------------------
constexpr int kMultiB = 4;

// calculateBranchLength for one slot (the arithmetic of place_tip_edges_kernel, from the slot-indexed arrays)
__device__ __forceinline__ void place_eval_slot(const PlaceBuffers& p, const double* __restrict__ dis, int sl, double& add, double& d1,
                                                int& eid, int& myrev)
{
    add = 2.0; d1 = 0.0; eid = 0; myrev = -1;
    const int bel = p.belong[sl], tgt = p.e[sl];
    const int oe = p.rev[sl];
    const double L = p.len[sl];
    if (bel >= tgt) {
        double cd[2 * K5];
        int ci[2 * K5];
#pragma unroll
        for (int i = 0; i < K5; ++i) { ci[i] = p.cid[sl * K5 + i]; cd[i] = p.cdis[sl * K5 + i]; }
#pragma unroll
        for (int i = 0; i < K5; ++i) { ci[K5 + i] = p.cid[oe * K5 + i]; cd[K5 + i] = p.cdis[oe * K5 + i]; }
        eid = sl;
        myrev = oe;
        double dv[2 * K5];
#pragma unroll
        for (int i = 0; i < 2 * K5; ++i) dv[i] = ci[i] != -1 ? dis[ci[i]] : 0.0;
        double dis1 = 0, dis2 = 0, val;
#pragma unroll
        for (int i = 0; i < K5; ++i)
            if (ci[i] != -1) { val = dv[i] - cd[i]; if (val > dis1) dis1 = val; }
#pragma unroll
        for (int i = 0; i < K5; ++i)
            if (ci[K5 + i] != -1) { val = dv[K5 + i] - cd[K5 + i]; if (val > dis2) dis2 = val; }
        double a = (dis1 + dis2 - L) / 2;
        if (a < 0) a = 0;
        dis1 -= a; dis2 -= a;
        if (dis1 < 0) dis1 = 0;
        if (dis2 < 0) dis2 = 0;
        if (dis1 > L) { a += dis1 - L; dis1 = L; }
        if (dis2 > L) { a += dis2 - L; dis2 = L; }
        const double rest = L - dis1 - dis2;
        dis1 += rest / 2; dis2 += rest / 2;
        add = a; d1 = dis1;
    }
}

// scan of the EDGES [0, 2 num0 - 2) (edge records, as place_tip_edges_kernel) for the tips num0 .. num0 + nb - 1 (rows dis0 + j ldb):
// partials[j * nblk + block]; block b covers the edges [256 b, 256 b + 256)
__global__ __launch_bounds__(kTipThreadsM) void place_tip_multi_kernel(PlaceBuffers p, const double* __restrict__ dis0, int64_t ldb,
                                                                   int64_t num0, int nb, PlacePartial* __restrict__ partials, int nblk)
{
    __builtin_amdgcn_s_setprio(3);      // chains of dependent steps: these waves go first where a distance kernel shares the SIMD
    __shared__ double sadd[kMultiB][kTipThreadsM / 64];
    __shared__ int sidx[kMultiB][kTipThreadsM / 64];
    const int64_t nedge = 2 * num0 - 2;
    const int64_t k = (int64_t)blockIdx.x * kTipThreadsM + threadIdx.x;
    const bool have = k < nedge;
    double add[kMultiB], d1[kMultiB];
    int sl0 = 0x7fffffff, sl1 = -1;
#pragma unroll
    for (int j = 0; j < kMultiB; ++j) { add[j] = 2.0; d1[j] = 0.0; }
    if (have) {
        double cd[2 * K5];
        int ci[2 * K5];
#pragma unroll
        for (int i = 0; i < 2 * K5; ++i) { cd[i] = p.er_d[(int64_t)i * p.ecap + k]; ci[i] = p.er_i[(int64_t)i * p.ecap + k]; }
        const double L = p.er_d[(int64_t)10 * p.ecap + k];
        sl0 = p.er_i[(int64_t)10 * p.ecap + k]; sl1 = p.er_i[(int64_t)11 * p.ecap + k];
#pragma unroll
        for (int j = 0; j < kMultiB; ++j) {
            if (j >= nb) break;
            const double* __restrict__ dis = dis0 + (int64_t)j * ldb;
            double dv[2 * K5];
#pragma unroll
            for (int i = 0; i < 2 * K5; ++i) dv[i] = ci[i] != -1 ? dis[ci[i]] : 0.0;
            double dis1 = 0, dis2 = 0, val;
#pragma unroll
            for (int i = 0; i < K5; ++i)
                if (ci[i] != -1) { val = dv[i] - cd[i]; if (val > dis1) dis1 = val; }
#pragma unroll
            for (int i = 0; i < K5; ++i)
                if (ci[K5 + i] != -1) { val = dv[K5 + i] - cd[K5 + i]; if (val > dis2) dis2 = val; }
            double a = (dis1 + dis2 - L) / 2;
            if (a < 0) a = 0;
            dis1 -= a; dis2 -= a;
            if (dis1 < 0) dis1 = 0;
            if (dis2 < 0) dis2 = 0;
            if (dis1 > L) { a += dis1 - L; dis1 = L; }
            if (dis2 > L) { a += dis2 - L; dis2 = L; }
            const double rest = L - dis1 - dis2;
            dis1 += rest / 2; dis2 += rest / 2;
            add[j] = a; d1[j] = dis1;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double badd[kMultiB];
    int bidx[kMultiB];
#pragma unroll
    for (int j = 0; j < kMultiB; ++j) {
        badd[j] = (have && add[j] == add[j]) ? add[j] : __builtin_inf();      // NaN never wins
        bidx[j] = have ? sl0 : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double oa = __shfl_down(badd[j], off, 64);
            const int oi = __shfl_down(bidx[j], off, 64);
            if (oa < badd[j] || (oa == badd[j] && oi < bidx[j])) { badd[j] = oa; bidx[j] = oi; }
        }
        if (lane == 0) { sadd[j][w] = badd[j]; sidx[j][w] = bidx[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kMultiB; ++j) {
        if (j >= nb) break;
        double ba = sadd[j][0];
        int bi = sidx[j][0];
#pragma unroll
        for (int i = 1; i < kTipThreadsM / 64; ++i)
            if (sadd[j][i] < ba || (sadd[j][i] == ba && sidx[j][i] < bi)) { ba = sadd[j][i]; bi = sidx[j][i]; }
        if (have && sl0 == bi) {
            PlacePartial pp; pp.add = add[j]; pp.idx = sl0; pp.eid = sl0; pp.frac = d1[j]; pp.rev = sl1; pp.pad = 0;
            partials[(int64_t)j * nblk + blockIdx.x] = pp;
        }
        if (threadIdx.x == 0 && bi == 0x7fffffff) {
            PlacePartial pp; pp.add = __builtin_inf(); pp.idx = 0x7fffffff; pp.eid = 0; pp.frac = 0; pp.rev = -1; pp.pad = 0;
            partials[(int64_t)j * nblk + blockIdx.x] = pp;
        }
    }
}

constexpr int kMultiMaxThreads = 1024;      // (launched with 256 threads, or 1024 when there are many block minima to go through)
__global__ __launch_bounds__(kMultiMaxThreads) void place_update_multi_kernel(PlaceBuffers p, const PlacePartial* __restrict__ partials, int nblk,
                                                                         int64_t num0, int nb, const double* __restrict__ dis0, int64_t ldb,
                                                                         double* __restrict__ trace)
{
    __builtin_amdgcn_s_setprio(3);      // chains of dependent steps: these waves go first where a distance kernel shares the SIMD
    __shared__ int s_hash[kDirtyHash];
    __shared__ int s_list[kDirtyCap];
    __shared__ int s_count, s_nrescan;
    __shared__ int s_rescan[kRescanCap];
    __shared__ int32_t sq_id[kQueueLds];
    __shared__ double sq_dis[kQueueLds];
    __shared__ double s_add[kMultiMaxThreads / 64], s_frac[kMultiMaxThreads / 64];
    __shared__ int s_idx[kMultiMaxThreads / 64], s_eid[kMultiMaxThreads / 64], s_rev[kMultiMaxThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, nthr = (int)blockDim.x;
    for (int i = tid; i < kDirtyHash; i += nthr) s_hash[i] = -1;
    __shared__ int s_sync;                    // place_split_wave: wavefront 0's loads are back
    if (tid == 0) { s_count = 0; s_nrescan = 0; s_sync = -1; }
    __syncthreads();
    DirtySet ds{ s_hash, s_list, &s_count };
    const int64_t nedge0 = 2 * num0 - 2;                // edges the scan launch covered
    for (int jt = 0; jt < nb; ++jt) {
        const int64_t num = num0 + jt;
        const int ec0 = (int)(4 * num - 4);             // live slots of this tip = slot id of its first new slot
        const double* __restrict__ dis = dis0 + (int64_t)jt * ldb;
        const PlacePartial* __restrict__ part = partials + (int64_t)jt * nblk;
        double badd = __builtin_inf(), bfrac = 0;
        int bidx = 0x7fffffff, beid = 0, brev = -1;
        auto consider = [&](double a, int idx, int eid, double frac, int rv) {
            const double aa = a == a ? a : __builtin_inf();
            if (aa < badd || (aa == badd && idx < bidx)) { badd = aa; bidx = idx; beid = eid; bfrac = frac; brev = rv; }
        };
        const unsigned long long ck0 = wall_clock64();
        // what the four new slots' lists hold (wavefronts 0 and 1, an entry per lane): loaded here, before the barriers of this
        // tip, because wavefront 0 overwrites three of them while wavefront 1 may still be on its way (place_split_wave)
        double i0d = 2.0, i1d = 2.0, i2d = 2.0, i3d = 2.0;
        int i0i = -1, i1i = -1, i2i = -1, i3i = -1;
        if (tid < 128 && lane < K5) {
            i0d = p.cdis[ec0 * K5 + lane]; i0i = p.cid[ec0 * K5 + lane];
            i1d = p.cdis[(ec0 + 1) * K5 + lane]; i1i = p.cid[(ec0 + 1) * K5 + lane];
            i2d = p.cdis[(ec0 + 2) * K5 + lane]; i2i = p.cid[(ec0 + 2) * K5 + lane];
            i3d = p.cdis[(ec0 + 3) * K5 + lane]; i3i = p.cid[(ec0 + 3) * K5 + lane];
        }
        const int ndirty = s_count;                      // (stable: written before the last barrier)
        const bool overflow = ndirty > kDirtyCap;
        int mis = 0x7fffffff;
        if (tid == 0) mis = p.misc[0];                   // (kept by the splits; the previous tip's is behind the barrier)
        if (!overflow) {
            // (A) the speculative block minima, their loads in flight together (a loop over them is one dependent round trip per
            // entry: 4 300 blocks at 550 000 tips).  A minimum whose winner is untouched stands; the others' blocks are evaluated
            // again in (C)
            auto take = [&](const PlacePartial& q, int b) {
                if (jt > 0 && q.idx != 0x7fffffff && dirty_has(ds, q.idx)) {
                    const int k = atomicAdd(&s_nrescan, 1);
                    if (k < kRescanCap) s_rescan[k] = b;
                } else {
                    consider(q.add, q.idx, q.eid, q.frac, q.rev);
                }
            };
            constexpr int kPB = 5;
            for (int b0 = tid; b0 < nblk; b0 += kPB * nthr) {
                PlacePartial pp[kPB];
#pragma unroll
                for (int u = 0; u < kPB; ++u) {
                    const int b = b0 + u * nthr;
                    if (b < nblk) pp[u] = part[b];
                    else { pp[u].add = __builtin_inf(); pp[u].idx = 0x7fffffff; pp[u].eid = 0; pp[u].frac = 0; pp[u].rev = -1; }
                }
#pragma unroll
                for (int u = 0; u < kPB; ++u)
                    if (b0 + u * nthr < nblk) take(pp[u], b0 + u * nthr);
            }
            // (B) the dirty slots with the current state (new slots included): three dependent hops
            for (int k = tid; k < ndirty; k += nthr) {
                const int sl = s_list[k];
                if (sl < ec0) {
                    double a, f; int e2, rv;
                    place_eval_slot(p, dis, sl, a, f, e2, rv);
                    consider(a, sl, e2, f, e2 ? rv : -1);
                }
            }
        }
        __syncthreads();
        const unsigned long long ck1 = wall_clock64();
        const int nres = s_nrescan;
        if (overflow || nres > kRescanCap) {
            // too much changed for the bookkeeping (small trees: every list still has room): scan everything here
            if (tid == 0) atomicAdd(&p.misc[overflow ? 2 : 3], 1);
            badd = __builtin_inf(); bfrac = 0; bidx = 0x7fffffff; beid = 0; brev = -1;
            for (int sl = tid; sl < ec0; sl += nthr) {
                double a, f; int e2, rv;
                place_eval_slot(p, dis, sl, a, f, e2, rv);
                consider(a, sl, e2, f, e2 ? rv : -1);
            }
        } else {
            // (C) blocks whose speculative winner is dirty: the evaluated slots of all their edges again (an edge created or
            // taken over since the scan launch -- its slots are dirty -- is evaluated here or in (B): same value, same key)
            for (int k = 0; k < nres; ++k) {
                const int64_t ek = (int64_t)s_rescan[k] * kTipThreadsM + tid;
                if (tid < kTipThreadsM && ek < nedge0) {
                    const int sl = p.er_i[(int64_t)10 * p.ecap + ek];
                    double a, f; int e2, rv;
                    place_eval_slot(p, dis, sl, a, f, e2, rv);
                    consider(a, sl, e2, f, e2 ? rv : -1);
                }
            }
        }
        // slots >= 4*num-4 (and < 4M-4) all carry the tuple (0,0,2): the first of them competes
        if (tid == 0 && (int64_t)ec0 < 4 * p.M - 4) consider(2.0, ec0, 0, 0.0, -1);
        // ... and so does the smallest live slot with belong < e (the edge scan does not visit those; p.misc[0] is kept by the splits)
        if (tid == 0) consider(2.0, mis, 0, 0.0, -1);
        {
            const double wa = wave_fmin(badd);
            const uint64_t wi = wave_umin64(badd == wa ? (uint64_t)(uint32_t)bidx : ~0ull);
            const unsigned long long own = __builtin_amdgcn_ballot_w64((badd == wa) & ((uint64_t)(uint32_t)bidx == wi));
            const int src = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(own));
            if (lane == src) { s_add[tid >> 6] = badd; s_idx[tid >> 6] = bidx; s_eid[tid >> 6] = beid; s_frac[tid >> 6] = bfrac; s_rev[tid >> 6] = brev; }
        }
        __syncthreads();
        const unsigned long long ck2 = wall_clock64();
        if (tid < 128) {                     // wavefronts 0 and 1: the split; wavefront 0: the closest-list update (as place_finish_and_update)
            const int role = __builtin_amdgcn_readfirstlane(tid >> 6);
            place_block_winner(s_add, s_idx, s_eid, s_frac, s_rev, nthr / 64, badd, bidx, beid, bfrac, brev);
            const int eid = beid;
            const double fracLen = bfrac, addLen = badd;
            const int placeId = (int)num;
            if (tid == 0) {
                if (trace) { trace[3 * num] = eid; trace[3 * num + 1] = fracLen; trace[3 * num + 2] = addLen; }
                s_nrescan = 0;
            }
            int xe, ye;
            BfsPre pre;
            const int bfs_ns = place_split_wave(p, num, ec0, eid, brev, fracLen, addLen, i0d, i0i, i1d, i1i, i2d, i2i, i3d, i3i, p.misc[0], role, pre, &s_sync, xe, ye);
            if (role == 0) {
                // what the split changed for later evaluations: the edge's two slots and the four new ones (a lane each)
                if (lane < 6) dirty_add(ds, lane == 0 ? xe : lane == 1 ? ye : ec0 + lane - 2);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (bfs_ns < 0) {            // (degree > 3 behind x or y) from the new leaf's only slot, outside -> middle, with distance 0
                    if (tid == 0) { sq_id[0] = ec0 + 2; sq_dis[0] = 0.0; }
                    const int reached = closest_update_wave_impl<true>(p, placeId, 1, ds, sq_id, sq_dis);
                    if (tid == 0) p.bfs_cnt[placeId] = -reached - 1;
                } else if (bfs_ns > 0) {     // from where the rounds done in registers got, records loaded ahead
                    const int reached = closest_update_wave_impl<true>(p, placeId, 0, ds, sq_id, sq_dis, &pre);
                    if (tid == 0) p.bfs_cnt[placeId] = reached;
                }
            }
        }
        // the other wavefronts evaluate the next tip against what wavefronts 0 and 1 have just stored
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if ((p.dbg & 4) && trace && tid == 0) {   // profiles/place_phases.py: evaluation (A + B) / rescans + winner / split + BFS, 10 ns units
            const unsigned long long ck3 = wall_clock64();
            trace[3 * num] = (double)(ck1 - ck0); trace[3 * num + 1] = (double)(ck2 - ck1); trace[3 * num + 2] = (double)(ck3 - ck2);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int place_alloc(PlaceBuffers& p, int64_t N, int64_t M)
{
    place_free(p);
    p.N = N;
    if (const char* e = std::getenv("DPR_PLACE_CLOCKS")) p.dbg = std::atoi(e) ? 4 : 0;   // profiles/place_phases.py
    p.M = M > 0 ? M : N;
    DPR_HIP(hipMalloc(&p.head, sizeof(int32_t) * (size_t)(2 * N)));
    DPR_HIP(hipMalloc(&p.e, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.nxt, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.belong, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.rev, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.len, sizeof(double) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.cont, sizeof(int32_t) * (size_t)(16 * N)));
    DPR_HIP(hipMalloc(&p.cid, sizeof(int32_t) * (size_t)(40 * N)));
    DPR_HIP(hipMalloc(&p.cdis, sizeof(double) * (size_t)(40 * N)));
    p.ecap = (2 * p.M + 63) / 64 * 64;                                         // 2M - 2 edges
    DPR_HIP(hipMalloc(&p.er_d, sizeof(double) * (size_t)(11 * p.ecap)));
    DPR_HIP(hipMalloc(&p.er_i, sizeof(int32_t) * (size_t)(12 * p.ecap)));
    DPR_HIP(hipMalloc(&p.eidx, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.misc, sizeof(int32_t) * 16));
    DPR_HIP(hipMemset(p.er_d, 0, sizeof(double) * (size_t)(11 * p.ecap)));
    DPR_HIP(hipMemset(p.er_i, 0xff, sizeof(int32_t) * (size_t)(12 * p.ecap)));
    DPR_HIP(hipMemset(p.eidx, 0, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMemset(p.misc, 0, sizeof(int32_t) * 16));
    DPR_HIP(hipMalloc(&p.bfs_cnt, sizeof(int32_t) * (size_t)(N + 64)));
    DPR_HIP(hipMemset(p.bfs_cnt, 0, sizeof(int32_t) * (size_t)(N + 64)));
    DPR_HIP(hipMalloc(&p.q_id, sizeof(int32_t) * (size_t)(4 * N + 64)));      // BFS frontier: every slot at most once
    DPR_HIP(hipMalloc(&p.q_from, sizeof(int32_t) * (size_t)(2 * N + 64)));
    DPR_HIP(hipMalloc(&p.q_dis, sizeof(double) * (size_t)(4 * N + 64)));
    // the arrays hold 8N slots, placement uses 4N - 4 of them: the rest is defined too (what dpr_place_run copies back)
    DPR_HIP(hipMemset(p.e, 0xff, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMemset(p.nxt, 0xff, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMemset(p.belong, 0xff, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMemset(p.rev, 0xff, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMemset(p.len, 0, sizeof(double) * (size_t)(8 * N)));
    DPR_HIP(hipDeviceSynchronize());      // (null-stream fills: the context's stream does not wait for them)
    p.nparts_max = (int)((4 * N + kThreads - 1) / kThreads + 1);
    DPR_HIP(hipMalloc(&p.partials, sizeof(PlacePartial) * (size_t)p.nparts_max));
    p.nparts_multi = (int64_t)p.nparts_max * kMultiB;
    DPR_HIP(hipMalloc(&p.partials_multi, sizeof(PlacePartial) * (size_t)p.nparts_multi));
    return DPR_OK;
}

void place_free(PlaceBuffers& p)
{
    void* ptrs[] = { p.head, p.e, p.nxt, p.belong, p.rev, p.len, p.cid, p.cdis, p.q_id, p.q_from, p.q_dis, p.partials, p.partials_multi, p.cont,
                     p.er_d, p.er_i, p.eidx, p.misc, p.bfs_cnt };
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    p = PlaceBuffers();
}

int place_init_fresh(PlaceBuffers& p, hipStream_t s)
{
    const int64_t lim = 4 * p.N - 4, nodes = 2 * p.N;
    const int64_t tot = lim > nodes ? lim : nodes;
    hipLaunchKernelGGL(place_init_kernel, dim3((unsigned)((tot + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p, lim, nodes);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int place_initial_tree(PlaceBuffers& p, const double* d_dis_row1, hipStream_t s)
{
    hipLaunchKernelGGL(place_initial_tree_kernel, dim3(1), dim3(64), 0, s, p, d_dis_row1);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int place_import_backbone(PlaceBuffers& p, int64_t m, hipStream_t s)
{
    const int64_t lim = 4 * p.N - 4;
    hipLaunchKernelGGL(place_init_lists_kernel, dim3((unsigned)((lim + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p, lim);
    const int64_t nslots = 4 * m - 4;
    hipLaunchKernelGGL(place_pair_rev_kernel, dim3((unsigned)((nslots + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p, nslots);
    hipLaunchKernelGGL(place_build_cont_kernel, dim3((unsigned)((nslots + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p, nslots);
    DPR_HIP(hipGetLastError());
    // level-by-level evaluation of the serial order's recurrence (see place_lists_level_kernel); the progress flag is
    // read every 16 rounds
    const bool serial = std::getenv("DPR_IMPORT_SERIAL") != nullptr;
    bool converged = false;
    if (!serial) {
        int32_t *level = nullptr, *pcnt = nullptr, *pid = nullptr;
        long long* poff = nullptr;
        double* pdis = nullptr;
        int* d_flags = nullptr;
        unsigned long long* pool_top = nullptr;
        // (a slot pointing away from the root has nearly all leaves behind it: its passers are the 5-records of their
        // distances in id order, O(5 ln m); measured mean over all slots of a 500 000-tip backbone: see DPR_LOG=import)
        const unsigned long long pool_cap = (unsigned long long)nslots * 192ull;
        auto release = [&]() { void* q[] = { level, pcnt, poff, pid, pdis, d_flags, pool_top }; for (void* x : q) if (x) (void)hipFree(x); };
        hipError_t ae = hipMalloc(&level, sizeof(int32_t) * (size_t)nslots);
        if (ae == hipSuccess) ae = hipMalloc(&pcnt, sizeof(int32_t) * (size_t)nslots);
        if (ae == hipSuccess) ae = hipMalloc(&poff, sizeof(long long) * (size_t)nslots);
        if (ae == hipSuccess) ae = hipMalloc(&pid, sizeof(int32_t) * (size_t)pool_cap);
        if (ae == hipSuccess) ae = hipMalloc(&pdis, sizeof(double) * (size_t)pool_cap);
        if (ae == hipSuccess) ae = hipMalloc(&d_flags, 2 * sizeof(int));
        if (ae == hipSuccess) ae = hipMalloc(&pool_top, sizeof(unsigned long long));
        if (ae != hipSuccess) { release(); return hip_fail(ae, "place_import_backbone: hipMalloc"); }
        DPR_HIP(hipMemsetAsync(level, 0xff, sizeof(int32_t) * (size_t)nslots, s));
        DPR_HIP(hipMemsetAsync(pcnt, 0, sizeof(int32_t) * (size_t)nslots, s));
        DPR_HIP(hipMemsetAsync(pool_top, 0, sizeof(unsigned long long), s));
        const unsigned grid = (unsigned)((nslots + kThreads - 1) / kThreads);
        const int max_rounds = 8192;
        int rounds = 0, rc = DPR_OK;
        bool bail = false;
        while (rounds < max_rounds && !converged && !bail) {
            DPR_HIP(hipMemsetAsync(d_flags, 0, 2 * sizeof(int), s));
            for (int k = 0; k < 16; ++k, ++rounds)
                hipLaunchKernelGGL(place_lists_level_kernel, dim3(grid), dim3(kThreads), 0, s, p, level, pcnt, poff, pid, pdis, pool_top,
                                   pool_cap, rounds, nslots, m, d_flags);
            int h[2] = { 1, 1 };
            if (hipMemcpyAsync(h, d_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = DPR_ERR_HIP; break; }
            if (h[1]) bail = true;                 // passer pool exhausted or a node of very high degree
            else converged = (h[0] == 0);          // no slot finished in the last 16 rounds: all are done
        }
        if (log_level("import") > 0) {
            std::vector<int32_t> hc((size_t)nslots);
            (void)hipMemcpy(hc.data(), pcnt, sizeof(int32_t) * (size_t)nslots, hipMemcpyDeviceToHost);
            int mx = 0; double sum = 0;
            for (int32_t v : hc) { mx = v > mx ? v : mx; sum += v; }
            std::fprintf(stderr, "[import] %d level rounds, bail %d, passer sequences: longest %d, mean %.1f\n", rounds, (int)bail, mx, sum / (double)nslots);
        }
        release();
        if (rc) { set_error("place_import_backbone: level rounds failed"); return rc; }
        if (bail) converged = false;
    }
    if (!converged) {
        // very deep trees (diameter > 8192 edges), an exhausted passer pool or DPR_IMPORT_SERIAL: the reference's order, leaf by leaf
        hipLaunchKernelGGL(place_init_lists_kernel, dim3((unsigned)((lim + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p, lim);
        const int64_t chunk = 4096;
        for (int64_t t0 = 0; t0 < m; t0 += chunk) {
            const int64_t t1 = t0 + chunk < m ? t0 + chunk : m;
            hipLaunchKernelGGL(place_backbone_lists_kernel, dim3(1), dim3(64), 0, s, p, t0, t1);
        }
    }
    {   // edge records of the imported tree (the lists are final now)
        const int32_t m0[2] = { 0x7fffffff, 0 };
        DPR_HIP(hipMemcpyAsync(p.misc, m0, sizeof m0, hipMemcpyHostToDevice, s));
        DPR_HIP(hipStreamSynchronize(s));      // (m0 is a stack variable)
        hipLaunchKernelGGL(place_pack_edges_kernel, dim3((unsigned)((nslots + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p, nslots);
    }
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// first tip of the four-tip launch pairs (smaller trees: one tip per pair of launches)
int64_t place_multi_min()
{
    return std::getenv("DPR_PLACE_MULTI_MIN") ? std::atoll(std::getenv("DPR_PLACE_MULTI_MIN")) : 150000;
}

// tips tip0 .. tip0 + count - 1, distance rows at d_dis0 + k * ldb: kMultiB tips per pair of launches once the tree is large
// enough for the shared scan to outweigh the fix-up chain of the update launch (measured: 100 000 tips 1.61 -> 1.91 s, i.e. NOT
// there; 300 000 tips 7.00 -> 6.76 s; 50 000 queries onto a 500 000-tip backbone 2.85 -> 2.59 s).  DPR_PLACE_MULTI_MIN moves the
// switch (the tests run the multi-tip path from the third tip on).
int place_tips(PlaceBuffers& p, const double* d_dis0, int64_t ldb, int64_t tip0, int64_t count, double* d_trace, hipStream_t s)
{
    // (read per call: the tests switch inside one process)
    const int64_t min_tip = place_multi_min();
    const bool big_block = std::getenv("DPR_PLACE_MULTI_BIG") != nullptr;      // tests: the 1024-thread update workgroup at any size
    int64_t k = 0;
    while (k < count) {
        const int64_t tip = tip0 + k;
        const int nblk = (int)((2 * tip - 2 + kTipThreadsM - 1) / kTipThreadsM);      // blocks of the four-tip edge scan
        const int nb = (int)(count - k < kMultiB ? count - k : kMultiB);
        if (tip < min_tip || nb < 2 || (int64_t)nblk * kMultiB > p.nparts_multi) {
            if (int rc = place_tip(p, d_dis0 + k * ldb, tip, d_trace, s)) return rc;
            k += 1;
            continue;
        }
        PlacePartial* parts = reinterpret_cast<PlacePartial*>(p.partials_multi);
        hipLaunchKernelGGL(place_tip_multi_kernel, dim3((unsigned)nblk), dim3(kTipThreadsM), 0, s, p, d_dis0 + k * ldb, ldb, tip, nb, parts, nblk);
        hipLaunchKernelGGL(place_update_multi_kernel, dim3(1), dim3((nblk > 2048 || big_block) ? kMultiMaxThreads : kUpdThreads), 0, s, p, (const PlacePartial*)parts, nblk, tip, nb,
                           d_dis0 + k * ldb, ldb, d_trace);
        DPR_HIP(hipGetLastError());
        k += nb;
    }
    return DPR_OK;
}

int place_tip(PlaceBuffers& p, const double* d_dis, int64_t tip, double* d_trace, hipStream_t s)
{
    PlacePartial* parts = reinterpret_cast<PlacePartial*>(p.partials);
    const int nblk = (int)((2 * tip - 2 + kTipThreads - 1) / kTipThreads);
    hipLaunchKernelGGL(place_tip_edges_kernel, dim3((unsigned)nblk), dim3(kTipThreads), 0, s, p, d_dis, tip, parts);
    hipLaunchKernelGGL(place_update_kernel, dim3(1), dim3(kUpdThreads), 0, s, p, (const PlacePartial*)parts, nblk, tip, d_trace);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

}  // namespace dpr
