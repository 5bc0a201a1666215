// Exact placement mode on gfx950.  Replaces PlacementDeviceArrays::findPlacementTree
// (src/placement.cu:508-789) and its kernels: initialize :119-140, buildInitialTree :245-293,
// updateFromBottomToTop :296-329, updateFromTopToBottom :331-364, calculateBranchLength :158-197 +
// thrust::min_element :688, updateTreeStructure :199-243, updateDfsRk :366-379, findEndRk :382-398 +
// thrust::reduce :746, updateDepth :400-416, stable_sort_by_key :766, updateLevelStEd :419-434.
//
// Per tip the reference launches one kernel per tree level (twice), a Thrust reduction, a Thrust
// stable sort of all node depths and four blocking device->host copies.  Here per tip FOUR launches (round 6):
//   px_patch_kernel       finish the argmin over the candidates' partials, split the edge, patch ranks / depths / sizes,
//                         rebuild the lists of small-subtree roots and top nodes (elementwise over the nodes)
//   px_small_up_kernel    bottom-up pass inside the small subtrees (a wavefront or a workgroup per subtree); spare workgroups
//                         pack the top nodes' records
//   px_top_kernel         both passes over the top tree (one workgroup; climbing / polling / level schedules)
//   px_small_down_kernel  top-down pass inside the small subtrees + the candidates of the NEXT placement (one per node)
// and the literal schedule of rounds 1-2 as the fallback for the reference's swap quirk (px_scan_kernel +
// px_step_literal_kernel: one workgroup, one barrier per level).  lim[], depths, ranks and lists live in HBM/L2 (O(N) per
// tip); between batches of 256 tips the host reads one counter (exact_adapt), nothing else returns to it.
#include <cstdlib>

#include "dpr_internal.hpp"

namespace dpr {

constexpr int kXT = 1024;   // threads of the single-workgroup top-tree kernel
// (a "small" subtree has at most ExactBuffers::sm nodes, one thread per node: 64 = one wavefront; 256 / 1 024 = one workgroup)
constexpr int kTopLds = 6144;   // top nodes whose pass values fit the workgroup's LDS (2 x 8 bytes each)
constexpr int kTopLevLds = 4094;   // levels of the top tree whose offsets fit LDS
constexpr int kTopReg = 4;         // top nodes per thread whose contexts stay in registers over both passes
constexpr int kTopClimb = 2048;    // the climbing schedule (px_top_climb) takes top trees of fewer nodes than this: 64 bytes of LDS each
constexpr size_t kTopDynLds = (size_t)kTopClimb * 64 + 2 * 1024 + 16;     // >= 2 * kTopLds doubles of the other schedules
static_assert(kTopDynLds >= sizeof(double) * 2 * kTopLds, "dynamic LDS of px_top_kernel");
constexpr unsigned long long kTopPollTicks = 20000000ull;      // 0.2 s of the 100 MHz clock: bound of one polling pass of px_top_poll

struct PlacePartialX {
    double add; int32_t idx; int32_t eid; double frac;
    // fast schedule only: what the split needs of the slot's state, carried with the candidate (px_patch_kernel must not read
    // p.e / p.rev / p.len of the winning slot while one of its threads rewrites them)
    int32_t xn, yn, ye, pad; double len;      // belong[eid], e[eid], rev[eid], len[eid]
};

__device__ __forceinline__ bool px_placed(int idx, int i, int N) { return !(idx > i && idx < N); }

// node record: (slot, reverse slot, target node) x 3, laid out slot[3], rslot[3], nb[3], pad[3]
__device__ __forceinline__ void px_set_node(const ExactBuffers& x, int node, int s0, int r0, int n0, int s1, int r1, int n1,
                                            int s2, int r2, int n2)
{
    int32_t* q = x.nd + 12 * (int64_t)node;
    q[0] = s0; q[1] = s1; q[2] = s2; q[3] = r0; q[4] = r1; q[5] = r2; q[6] = n0; q[7] = n1; q[8] = n2;
}
__device__ __forceinline__ void px_retarget(const ExactBuffers& x, int node, int slot, int new_rev, int new_nb)
{
    int32_t* q = x.nd + 12 * (int64_t)node;
    for (int k = 0; k < 3; ++k)
        if (q[k] == slot) { q[3 + k] = new_rev; q[6 + k] = new_nb; }
}

// The recurrences of the two passes (src/placement.cu:296-364) are functions of a node's neighbours only, so ANY schedule
// that evaluates children before parents (bottom-up) and parents before children (top-down) gives the reference's lim[] bit
// for bit.  Round 3 schedule (the reference: one launch per tree level and direction; rounds 1-2: one workgroup, one barrier
// per level, 18 MB per tip through one CU at 30 000 tips):
//   * the tree is cut, per tip, into SMALL subtrees (maximal subtrees of <= 64 nodes) and the TOP tree above them.  With
//     pre-order ranks and subtree sizes maintained next to the reference's depths, a small subtree is a contiguous run of ranks,
//     "v is a block root" is a local test (size(v) <= 64 < size(parent(v))), and one WAVEFRONT evaluates a whole small subtree
//     with one lane per node, the values travelling through LDS: all small subtrees in parallel on all CUs;
//   * the top tree (a few per cent of the nodes) is evaluated level by level by one workgroup as before, but its values live
//     in LDS, so a level costs an LDS round trip and a barrier instead of an L2 round trip;
//   * the O(N) bookkeeping after a split (rank shift, depth update, subtree sizes, the node-at-rank table, the two node lists)
//     is ONE elementwise kernel over all CUs: every new value is a pure function of the OLD arrays (double-buffered ranks and
//     sizes) and four scalars of the split.  The end of the moved subtree, which the reference finds with a reduction over all
//     nodes (findEndRk, :382-398 + thrust::reduce :746), is rank(y) + size(y) + 1 here -- the same number.

// node context of a lane / thread: its (up to three) edges
struct XCtx {
    int v;                   // node (-1: idle lane)
    int slot[3], rslot[3];
    int ref[3];              // small subtrees: lane of the neighbour; top tree: index of the neighbour in the top list (-1: not a top node)
    bool down[3];            // the edge leads to a child (pre-order rank of the neighbour is larger)
    double len[3];
    double init;             // 0, or the distance of the tip for a leaf (src/placement.cu:317-318)
};

__device__ __forceinline__ XCtx px_ctx(const ExactBuffers& x, const PlaceBuffers& p, const double* __restrict__ dis,
                                       const int32_t* __restrict__ rk, int v, int base_rank, bool top)
{
    XCtx c;
    c.v = v;
#pragma unroll
    for (int k = 0; k < 3; ++k) { c.slot[k] = -1; c.rslot[k] = -1; c.ref[k] = -1; c.down[k] = false; c.len[k] = 0; }
    c.init = 0;
    if (v < 0) return c;
    const int4 a = reinterpret_cast<const int4*>(x.nd + 12 * (int64_t)v)[0];
    const int4 b = reinterpret_cast<const int4*>(x.nd + 12 * (int64_t)v)[1];
    const int4 cc = reinterpret_cast<const int4*>(x.nd + 12 * (int64_t)v)[2];
    c.slot[0] = a.x; c.slot[1] = a.y; c.slot[2] = a.z;
    c.rslot[0] = a.w; c.rslot[1] = b.x; c.rslot[2] = b.y;
    const int nb[3] = { b.z, b.w, cc.x };
    const int myrk = rk[v];
    if (v < (int)p.N) c.init = dis[v];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (c.slot[k] >= 0) {
            const int nrk = rk[nb[k]];
            c.down[k] = nrk > myrk;
            c.ref[k] = top ? x.tix[nb[k]] : nrk - base_rank;
            c.len[k] = p.len[c.slot[k]];
        }
    return c;
}

// calculateBranchLength (src/placement.cu:158-197) for one slot: pendant length `add` and the position `d1` on the branch
__device__ __forceinline__ void px_candidate(double dis1, double dis2, double L, double& add, double& d1)
{
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

// first minimum over (pendant length, slot): the running best of a lane, of a wavefront, of a workgroup -> partials[block]
struct XBest { double key, add, frac, len; int idx, xn, yn, ye; };      // key: the pendant length, +inf for a NaN (which never wins, as in px_scan_kernel)
__device__ __forceinline__ void px_best_init(XBest& b) { b.key = __builtin_inf(); b.add = __builtin_inf(); b.frac = 0.0; b.len = 0.0; b.idx = 0x7fffffff; b.xn = b.yn = b.ye = -1; }
// candidate of slot idx = xn -> yn (reverse slot ye, length len)
__device__ __forceinline__ void px_best_take(XBest& b, double add, double frac, int idx, int xn, int yn, int ye, double len)
{
    const double key = add == add ? add : __builtin_inf();
    if (key < b.key || (key == b.key && idx < b.idx)) { b.key = key; b.add = add; b.frac = frac; b.len = len; b.idx = idx; b.xn = xn; b.yn = yn; b.ye = ye; }
}
// (all threads of the workgroup call this once; s_key / s_idx: one entry per wavefront of the workgroup)
__device__ __forceinline__ void px_best_store(XBest b, double* s_key, int* s_idx, PlacePartialX* __restrict__ out)
{
    // (the winner's lane within the wavefront first, then its payload with one shuffle per field)
    double k = b.key; int ki = b.idx;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ok = __shfl_xor(k, off, 64);
        const int oi = __shfl_xor(ki, off, 64);
        if (ok < k || (ok == k && oi < ki)) { k = ok; ki = oi; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const unsigned long long mine = __builtin_amdgcn_ballot_w64(b.idx == ki && (b.key == k));
    const int src = mine ? __builtin_ctzll(mine) : 0;      // (no candidate in the wavefront: every lane holds the initial record)
    b.key = __shfl(b.key, src, 64); b.add = __shfl(b.add, src, 64); b.frac = __shfl(b.frac, src, 64); b.len = __shfl(b.len, src, 64);
    b.idx = __shfl(b.idx, src, 64); b.xn = __shfl(b.xn, src, 64); b.yn = __shfl(b.yn, src, 64); b.ye = __shfl(b.ye, src, 64);
    __syncthreads();      // (the arrays may still be read as something else by a slower wavefront)
    if (lane == 0) { s_key[w] = b.key; s_idx[w] = b.idx; }
    __syncthreads();
    int win = 0;
    for (int i = 1; i < nw; ++i)
        if (s_key[i] < s_key[win] || (s_key[i] == s_key[win] && s_idx[i] < s_idx[win])) win = i;
    if (w == win && lane == 0) {
        PlacePartialX pp;
        pp.add = b.add; pp.idx = b.idx; pp.eid = b.idx == 0x7fffffff ? 0 : b.idx; pp.frac = b.frac;
        pp.xn = b.xn; pp.yn = b.yn; pp.ye = b.ye; pp.pad = 0; pp.len = b.len;
        *out = pp;
    }
}

// calculateBranchLength over the live slots + block-level first minimum (the literal schedule; the fast one evaluates the slots
// where their values are produced: px_small_down_kernel)
__global__ __launch_bounds__(kThreads) void px_scan_kernel(PlaceBuffers p, ExactBuffers x, int64_t num,
                                                           PlacePartialX* __restrict__ partials)
{
    __shared__ double sadd[kThreads / 64];
    __shared__ int sidx[kThreads / 64];
    const int64_t live = 4 * num - 4;
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    double add = 2.0, d1 = 0.0;
    int eid = 0;
    const bool have = idx < live;
    if (have && !(x.dep[p.belong[idx]] > x.dep[p.e[idx]])) {
        eid = (int)idx;
        px_candidate(x.lim[eid], x.lim[p.rev[eid]], p.len[eid], add, d1);
    }
    double badd = have ? add : __builtin_inf();
    int bidx = have ? (int)idx : 0x7fffffff;
    if (have && !(add == add)) badd = __builtin_inf();
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
        for (int i = 1; i < kThreads / 64; ++i)
            if (sadd[i] < badd || (sadd[i] == badd && sidx[i] < bidx)) { badd = sadd[i]; bidx = sidx[i]; }
        sadd[0] = badd; sidx[0] = bidx;
    }
    __syncthreads();
    if (have && (int)idx == sidx[0]) {
        PlacePartialX pp; pp.add = add; pp.idx = (int)idx; pp.eid = eid; pp.frac = d1;
        pp.xn = pp.yn = pp.ye = -1; pp.pad = 0; pp.len = 0.0;      // (the literal schedule reads the slot's state itself)
        partials[blockIdx.x] = pp;
    }
    if (threadIdx.x == 0 && sidx[0] == 0x7fffffff) {
        PlacePartialX pp; pp.add = __builtin_inf(); pp.idx = 0x7fffffff; pp.eid = 0; pp.frac = 0;
        pp.xn = pp.yn = pp.ye = -1; pp.pad = 0; pp.len = 0.0;
        partials[blockIdx.x] = pp;
    }
}

// ================================================================================================
// LITERAL schedule (rounds 1-2): one workgroup per tip, level lists over ALL nodes by the reference's depths, one barrier
// per level.  Kept as the fallback for the one situation in which the reference's own result depends on its level order:
// when the default tuple (slot 0, pendant length 2) wins the argmin, updateTreeStructure's swap (:236-239) leaves depths
// that are not the tree's, and only the level-by-depth schedule reproduces what the reference then computes.  The fast
// schedule notices (XStep::quirk) and dpr_place_exact_run repeats the run with this one; DPR_EXACT_LITERAL=1 forces it.
// ================================================================================================
// Node record (12 ints): the node's up to three slots, their reverse slots and their target nodes.  Slots
// are write-once per node (a split only retargets them), so a record changes only where the split
// happens.  With it a level step needs ONE dependent memory hop: everything that does not depend on the
// previous level (record, depths, lengths, the leaf's distance) is loaded one level ahead.
struct NodeCtx {
    int idx;            // -1: none
    int slot[3], rslot[3];
    bool down[3];       // edge leads to a deeper node (src/placement.cu:320,346: dep[e[i]] > dep[idx])
    double len[3];
    double init;        // 0, or the distance for a leaf (src/placement.cu:317-318)
};

__device__ __forceinline__ NodeCtx px_load_ctx(const ExactBuffers& x, const PlaceBuffers& p, const double* __restrict__ dis,
                                               int t, int t1)
{
    NodeCtx c;
    c.idx = -1;
#pragma unroll
    for (int k = 0; k < 3; ++k) { c.slot[k] = -1; c.rslot[k] = -1; c.down[k] = false; c.len[k] = 0; }
    c.init = 0;
    if (t < t1) {
        const int idx = x.order[t];
        c.idx = idx;
        const int4 a = reinterpret_cast<const int4*>(x.nd + 12 * (int64_t)idx)[0];
        const int4 b = reinterpret_cast<const int4*>(x.nd + 12 * (int64_t)idx)[1];
        const int4 cc = reinterpret_cast<const int4*>(x.nd + 12 * (int64_t)idx)[2];
        c.slot[0] = a.x; c.slot[1] = a.y; c.slot[2] = a.z;
        c.rslot[0] = a.w; c.rslot[1] = b.x; c.rslot[2] = b.y;
        const int nb[3] = { b.z, b.w, cc.x };
        const int dd = x.dep[idx];
        if (idx < (int)p.N) c.init = dis[idx];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (c.slot[k] >= 0) { c.down[k] = x.dep[nb[k]] > dd; c.len[k] = p.len[c.slot[k]]; }
    }
    return c;
}

// bottom-up: lim[slot to the parent] = max(init, lim[child -> node] - len) over the child edges
__device__ __forceinline__ void px_up(const ExactBuffers& x, const NodeCtx& c)
{
    if (c.idx < 0) return;
    double mx = c.init;
    int up = -1;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (c.slot[k] >= 0) {
            if (c.down[k]) { const double req = x.lim[c.rslot[k]] - c.len[k]; if (req > mx) mx = req; }
            else up = c.slot[k];
        }
    if (up >= 0) x.lim[up] = mx;
}

// top-down: lim[slot to a child] = max(0, lim[other -> node] - len) over the node's other edges
__device__ __forceinline__ void px_down(const ExactBuffers& x, const NodeCtx& c)
{
    if (c.idx < 0) return;
    double rq[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rq[k] = c.slot[k] >= 0 ? x.lim[c.rslot[k]] - c.len[k] : 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (c.slot[a] >= 0 && c.down[a]) {
            double mx = 0;
#pragma unroll
            for (int b = 0; b < 3; ++b)
                if (b != a && c.slot[b] >= 0 && rq[b] > mx) mx = rq[b];
            x.lim[c.slot[a]] = mx;
        }
}

// level-synchronous passes for one tip (its distance row `dis`): lim[slot x->y] = max(0 | dist[x] for a
// leaf, max over the other edges (x,z) of lim[z->x] - len) -- bottom-up fills child->parent slots,
// top-down parent->child slots (src/placement.cu:296-364); one workgroup barrier per level
__device__ void px_dp(const ExactBuffers& x, const PlaceBuffers& p, const double* __restrict__ dis, int maxdep)
{
    const int tid = threadIdx.x;
    NodeCtx cur = px_load_ctx(x, p, dis, x.lvoff[maxdep] + tid, x.lvoff[maxdep + 1]);
    for (int j = maxdep; j >= 0; --j) {
        const int t0 = x.lvoff[j], t1 = x.lvoff[j + 1];
        NodeCtx nxt;
        nxt.idx = -1;
        if (j > 0) nxt = px_load_ctx(x, p, dis, x.lvoff[j - 1] + tid, t0);       // one level ahead
        px_up(x, cur);
        for (int t = t0 + tid + kXT; t < t1; t += kXT) px_up(x, px_load_ctx(x, p, dis, t, t1));
        __syncthreads();
        cur = nxt;
    }
    cur = px_load_ctx(x, p, dis, x.lvoff[0] + tid, x.lvoff[1]);
    for (int j = 0; j <= maxdep; ++j) {
        const int t0 = x.lvoff[j], t1 = x.lvoff[j + 1];
        NodeCtx nxt;
        nxt.idx = -1;
        if (j < maxdep) nxt = px_load_ctx(x, p, dis, t1 + tid, x.lvoff[j + 2]);
        px_down(x, cur);
        for (int t = t0 + tid + kXT; t < t1; t += kXT) px_down(x, px_load_ctx(x, p, dis, t, t1));
        __syncthreads();
        cur = nxt;
    }
}

// dis_tree != nullptr: build the initial two-tip tree from it (row of tip 1), then the passes for tip 2.
// otherwise: place `tip` from the scan partials, patch ranks/depths/levels, passes for tip+1 (dis_next).
__global__ __launch_bounds__(kXT) void px_step_literal_kernel(PlaceBuffers p, ExactBuffers x,
                                                      const PlacePartialX* __restrict__ partials, int nparts,
                                                      int64_t tip, const double* __restrict__ dis_tree,
                                                      const double* __restrict__ dis_next, int has_next,
                                                      double* __restrict__ trace)
{
    __shared__ double s_add[kXT / 64], s_frac[kXT / 64];
    __shared__ int s_idx[kXT / 64], s_eid[kXT / 64], s_small[kXT / 64];
    __shared__ int s_ref_rk, s_ref_dep, s_maxdep, s_scan[kXT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int N = (int)p.N;
    int i;   // leaves placed so far are 0..i
    if (dis_tree) {
        if (tid == 0) {   // buildInitialTree (src/placement.cu:245-293)
            const int nv = N;
            const double d = dis_tree[0];
            int ec = 0;
            p.e[ec] = nv; p.len[ec] = d / 2; p.nxt[ec] = p.head[0]; p.head[0] = ec; p.belong[ec] = 0; ec++;
            p.e[ec] = nv; p.len[ec] = d / 2; p.nxt[ec] = p.head[1]; p.head[1] = ec; p.belong[ec] = 1; ec++;
            p.e[ec] = 0;  p.len[ec] = d / 2; p.nxt[ec] = p.head[nv]; p.head[nv] = ec; p.belong[ec] = nv; ec++;
            p.e[ec] = 1;  p.len[ec] = d / 2; p.nxt[ec] = p.head[nv]; p.head[nv] = ec; p.belong[ec] = nv; ec++;
            p.rev[0] = 2; p.rev[2] = 0; p.rev[1] = 3; p.rev[3] = 1;
            x.dep[nv] = 0; x.dep[0] = 1; x.dep[1] = 1;
            x.dfsrk[nv] = 0; x.dfsrk[0] = 1; x.dfsrk[1] = 2;
            px_set_node(x, 0, 0, 2, nv, -1, -1, -1, -1, -1, -1);
            px_set_node(x, 1, 1, 3, nv, -1, -1, -1, -1, -1, -1);
            px_set_node(x, nv, 2, 0, 0, 3, 1, 1, -1, -1, -1);
        }
        i = 1;
        __syncthreads();
    } else {
        i = (int)tip;
        // ---- finish the argmin (thrust::min_element over all 4N-4 tuples, first occurrence)
        double badd = __builtin_inf(), bfrac = 0;
        int bidx = 0x7fffffff, beid = 0;
        for (int k = tid; k < nparts; k += kXT) {
            const PlacePartialX pp = partials[k];
            if (pp.add < badd || (pp.add == badd && pp.idx < bidx)) { badd = pp.add; bidx = pp.idx; beid = pp.eid; bfrac = pp.frac; }
        }
        const int64_t live = 4 * (int64_t)i - 4, lim = 4 * (int64_t)N - 4;
        if (tid == 0 && live < lim)   // slots >= 4i-4 all carry (0,0,2): the first of them competes
            if (2.0 < badd || (2.0 == badd && (int)live < bidx)) { badd = 2.0; bidx = (int)live; beid = 0; bfrac = 0; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double oa = __shfl_down(badd, off, 64), of = __shfl_down(bfrac, off, 64);
            const int oi = __shfl_down(bidx, off, 64), oe = __shfl_down(beid, off, 64);
            if (oa < badd || (oa == badd && oi < bidx)) { badd = oa; bidx = oi; beid = oe; bfrac = of; }
        }
        if (lane == 0) { s_add[w] = badd; s_idx[w] = bidx; s_eid[w] = beid; s_frac[w] = bfrac; }
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < kXT / 64; ++k)
                if (s_add[k] < badd || (s_add[k] == badd && s_idx[k] < bidx)) { badd = s_add[k]; bidx = s_idx[k]; beid = s_eid[k]; bfrac = s_frac[k]; }
            const int eid = beid;
            const double fracLen = bfrac, addLen = badd;
            if (trace) { trace[3 * i] = eid; trace[3 * i + 1] = fracLen; trace[3 * i + 2] = addLen; }
            // ---- updateTreeStructure (src/placement.cu:199-243)
            int ec = 4 * i - 4;
            const int middle = i + N - 1, outside = i;
            int xn = p.belong[eid], yn = p.e[eid];
            const double originalDis = p.len[eid];
            const int xe = eid, ye = p.rev[eid];   // the reference finds them by walking head[x] / head[y]
            p.e[xe] = middle; p.len[xe] = fracLen; p.rev[xe] = ec;
            p.e[ye] = middle; p.len[ye] -= fracLen; p.rev[ye] = ec + 1;
            p.e[ec] = xn; p.len[ec] = fracLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle; p.rev[ec] = xe; ec++;
            p.e[ec] = yn; p.len[ec] = originalDis - fracLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle; p.rev[ec] = ye; ec++;
            p.e[ec] = middle; p.len[ec] = addLen; p.nxt[ec] = p.head[outside]; p.head[outside] = ec; p.belong[ec] = outside; p.rev[ec] = ec + 1; ec++;
            p.e[ec] = outside; p.len[ec] = addLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle; p.rev[ec] = ec - 1; ec++;
            // node records: x and y keep their slots, which now lead to `middle`
            px_retarget(x, xn, xe, ec - 4, middle);
            px_retarget(x, yn, ye, ec - 3, middle);
            px_set_node(x, middle, ec - 4, xe, xn, ec - 3, ye, yn, ec - 1, ec - 2, outside);
            px_set_node(x, outside, ec - 2, ec - 1, middle, -1, -1, -1, -1, -1, -1);
            if (x.dfsrk[xn] > x.dfsrk[yn]) { const int t2 = xn; yn = xn; xn = t2; }   // the reference's (ineffective) swap, :236-239
            x.dfsrk[middle] = x.dfsrk[yn];
            x.dfsrk[outside] = x.dfsrk[middle] + 1;
            x.dep[middle] = x.dep[xn]; x.dep[outside] = x.dep[middle] + 1;
            s_ref_rk = x.dfsrk[middle]; s_ref_dep = x.dep[middle];
        }
        __syncthreads();
        const int tot = N + i, ref = N + i - 1;
        const int rrk = s_ref_rk, rdep = s_ref_dep;
        // ---- updateDfsRk: ranks >= rank(middle) move up by 2 (middle and the new leaf excluded)
        for (int idx = tid; idx < tot; idx += kXT) {
            if (!px_placed(idx, i, N) || idx == ref || idx == i) continue;
            if (x.dfsrk[idx] >= rrk) x.dfsrk[idx] += 2;
        }
        __syncthreads();
        // ---- findEndRk + reduce(minimum, init N+i-1): last rank of the subtree that moves down
        int small = N + i - 1;
        for (int idx = tid; idx < tot; idx += kXT) {
            if (!px_placed(idx, i, N)) continue;
            const int rk = x.dfsrk[idx];
            if (rk <= rrk + 2 || x.dep[idx] > rdep + 1) continue;
            small = min(small, rk - 1);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) small = min(small, __shfl_xor(small, off, 64));
        if (lane == 0) s_small[w] = small;
        __syncthreads();
        small = s_small[0];
        for (int k = 1; k < kXT / 64; ++k) small = min(small, s_small[k]);
        // ---- updateDepth
        for (int idx = tid; idx < tot; idx += kXT) {
            if (!px_placed(idx, i, N)) continue;
            const int rk = x.dfsrk[idx];
            if (rk <= small && rk >= rrk) x.dep[idx]++;
        }
        __syncthreads();
    }
    if (!has_next) return;
    // ---- level lists by counting sort on depth (replaces stable_sort_by_key + updateLevelStEd; the order
    // inside a level is irrelevant to every result)
    const int tot = N + i;
    const int nplaced = 2 * i + 1;
    int mymax = 0;
    for (int idx = tid; idx < tot; idx += kXT)
        if (px_placed(idx, i, N)) mymax = max(mymax, x.dep[idx]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mymax = max(mymax, __shfl_xor(mymax, off, 64));
    if (lane == 0) s_small[w] = mymax;
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int k = 0; k < kXT / 64; ++k) m = max(m, s_small[k]);
        s_maxdep = m;
    }
    __syncthreads();
    const int maxdep = s_maxdep;
    for (int k = tid; k <= maxdep + 1; k += kXT) x.hist[k] = 0;
    __syncthreads();
    for (int idx = tid; idx < tot; idx += kXT)
        if (px_placed(idx, i, N)) atomicAdd(&x.hist[x.dep[idx]], 1);
    __syncthreads();
    // exclusive scan of hist[0..maxdep] -> lvoff, chunked over the workgroup
    {
        const int nlev = maxdep + 1;
        const int per = (nlev + kXT - 1) / kXT;
        const int b0 = tid * per, b1 = min(nlev, b0 + per);
        int sum = 0;
        for (int k = b0; k < b1; ++k) sum += x.hist[k];
        s_scan[tid] = sum;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int k = 0; k < kXT; ++k) { const int v = s_scan[k]; s_scan[k] = run; run += v; }
        }
        __syncthreads();
        int run = s_scan[tid];
        for (int k = b0; k < b1; ++k) { const int v = x.hist[k]; x.lvoff[k] = run; x.hist[k] = run; run += v; }
        if (tid == 0) x.lvoff[nlev] = nplaced;
    }
    __syncthreads();
    for (int idx = tid; idx < tot; idx += kXT)
        if (px_placed(idx, i, N)) x.order[atomicAdd(&x.hist[x.dep[idx]], 1)] = idx;
    __syncthreads();
    px_dp(x, p, dis_next, maxdep);
}


// ------------------------------------------------------------------------------------------------
// One launch per placed tip for everything between the passes: finish the argmin over the candidates' partials (thrust::min_element
// :688), split the edge (updateTreeStructure, src/placement.cu:199-243), updateDfsRk (:366-379), updateDepth (:400-416), subtree
// sizes, node-at-rank table, small-subtree roots and top nodes.  (Was: px_split_kernel, one workgroup, + the elementwise
// px_patch_kernel.)  EVERY workgroup finishes the argmin for itself -- a few hundred 48-byte records -- and derives the scalars of
// the split from it and from the OLD rank / size buffers, which nobody writes here; then every new value is a pure function of the
// old arrays and those scalars, as before.  The split itself is written by the threads that own the four nodes it touches: x and
// y retarget their own records, the thread of the new internal node writes the adjacency arrays, the trace and the status words.
// What the split needs of the winning slot (its ends, reverse slot and length) travels in the partial: the arrays being
// rewritten are not read by anybody else.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void px_patch_kernel(PlaceBuffers p, ExactBuffers x, const PlacePartialX* __restrict__ partials,
                                                            int nparts, int64_t tip, double* __restrict__ trace, int sm)
{
    __shared__ int s_cnt[2], s_base[2];
    __shared__ double s_add[kThreads / 64], s_key[kThreads / 64];
    __shared__ int s_idx[kThreads / 64], s_k[kThreads / 64], s_fidx[kThreads / 64], s_fk[kThreads / 64];
    const int N = (int)p.N, i = (int)tip;
    const int tot = N + i;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int idx = (int)((int64_t)blockIdx.x * kThreads + tid);
    if (tid < 2) s_cnt[tid] = 0;
    const int middle = i + N - 1, outside = i, ec0 = 4 * i - 4;
    const int32_t* __restrict__ rk_in = x.rk[i & 1];
    const int32_t* __restrict__ sz_in = x.sz[i & 1];
    int32_t* __restrict__ rk_out = x.rk[(i + 1) & 1];
    int32_t* __restrict__ sz_out = x.sz[(i + 1) & 1];
    const bool live = idx < tot && px_placed(idx, i, N);
    // ---- what the elementwise part reads of the OLD state does not depend on the split: in flight while the argmin is finished
    // (the node's record, old rank and size; old ranks and sizes of its neighbours)
    int32_t* const q = x.nd + 12 * (int64_t)(live ? idx : 0);
    int rec[9], r_old = 0, s_old = 0, d_old = 0, rnb[3] = { 0, 0, 0 }, snb[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 9; ++k) rec[k] = -1;
    if (live && idx != middle && idx != outside) {
        const int4 a = reinterpret_cast<const int4*>(q)[0], b = reinterpret_cast<const int4*>(q)[1];
        rec[0] = a.x; rec[1] = a.y; rec[2] = a.z; rec[3] = a.w; rec[4] = b.x; rec[5] = b.y; rec[6] = b.z; rec[7] = b.w; rec[8] = q[8];
        r_old = rk_in[idx]; s_old = sz_in[idx]; d_old = x.dep[idx];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (rec[k] >= 0) { rnb[k] = rk_in[rec[6 + k]]; snb[k] = sz_in[rec[6 + k]]; }
    }
    // ---- thrust::min_element over all 4N-4 tuples, first occurrence (and, beside it, the first minimum with a NaN counted as
    // +inf: the edge that is split when the default tuple wins, see below)
    double badd = __builtin_inf(), fkey = __builtin_inf();
    int bidx = 0x7fffffff, bk = -1, fidx = 0x7fffffff, fk = -1;
    for (int k = tid; k < nparts; k += kThreads) {
        const double a = partials[k].add;
        const int id = partials[k].idx;
        if (a < badd || (a == badd && id < bidx)) { badd = a; bidx = id; bk = k; }
        const double key = a == a ? a : __builtin_inf();
        if (id != 0x7fffffff && (key < fkey || (key == fkey && id < fidx))) { fkey = key; fidx = id; fk = k; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oa = __shfl_xor(badd, off, 64), ok = __shfl_xor(fkey, off, 64);
        const int oi = __shfl_xor(bidx, off, 64), obk = __shfl_xor(bk, off, 64), ofi = __shfl_xor(fidx, off, 64), ofk = __shfl_xor(fk, off, 64);
        if (oa < badd || (oa == badd && oi < bidx)) { badd = oa; bidx = oi; bk = obk; }
        if (ok < fkey || (ok == fkey && ofi < fidx)) { fkey = ok; fidx = ofi; fk = ofk; }
    }
    if (lane == 0) { s_add[w] = badd; s_idx[w] = bidx; s_k[w] = bk; s_key[w] = fkey; s_fidx[w] = fidx; s_fk[w] = fk; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kThreads / 64; ++k) {
        if (s_add[k] < badd || (s_add[k] == badd && s_idx[k] < bidx)) { badd = s_add[k]; bidx = s_idx[k]; bk = s_k[k]; }
        if (s_key[k] < fkey || (s_key[k] == fkey && s_fidx[k] < fidx)) { fkey = s_key[k]; fidx = s_fidx[k]; fk = s_fk[k]; }
    }
    const int64_t live_slots = 4 * (int64_t)i - 4, lim = 4 * (int64_t)N - 4;
    // slots >= 4i-4 all carry (0,0,2): the first of them competes.  When it wins, the reference splits slot 0 -- a slot that leads
    // UP the tree -- and its swap (:236-239) leaves depths that are no tree depths: the run is repeated with the literal schedule
    // (XStep::quirk), and all that matters here is that the tree stays a tree: the best real candidate is split instead.
    const bool dflt = (live_slots < lim && (2.0 < badd || (2.0 == badd && (int)live_slots < bidx))) || bk < 0;
    const int wk = dflt ? fk : bk;
    if (wk < 0) return;      // (no candidate at all: impossible for a tree with an edge; nothing to split)
    const PlacePartialX win = partials[wk];
    const int eid = win.eid;
    const double fracLen = win.frac, addLen = win.add, originalDis = win.len;
    const int xe = eid, ye = win.ye, xn0 = win.xn, yn0 = win.yn;      // the slot x -> y and its reverse
    const int rkx = rk_in[xn0], rky = rk_in[yn0], szx = sz_in[xn0], szy = sz_in[yn0], dxn = x.dep[xn0];
    const bool swapq = rkx > rky;      // the reference's swap, :236-239 (`yn = xn; xn = t2`: both become x)
    const bool quirk = dflt || swapq;
    // dfsrk[middle] = dfsrk[y], dfsrk[outside] = dfsrk[middle] + 1 and the shift of the ranks >= it
    const int rrk = swapq ? rkx : rky, ysz = swapq ? szx : szy;
    const int small = rrk + 1 + ysz;      // last rank of the moved subtree after the shift (findEndRk + reduce of the reference)
    const int cpar = (i + 1) & 1;
    // new rank / size of node v from its old ones
    auto new_rank = [&](int v, int r) { return v == middle ? rrk : v == outside ? rrk + 1 : (r >= rrk ? r + 2 : r); };
    auto new_size = [&](int v, int r, int sz) {
        return v == middle ? ysz + 2 : v == outside ? 1 : ((r < rrk && r + sz > rrk) ? sz + 2 : sz);      // the ancestors of y gain the two new nodes
    };
    int kind = -1, mine = 0;          // 0: root of a small subtree, 1: top node
    int4 root_rec = make_int4(0, 0, 0, 0);
    if (live) {
        // the node's record after the split
        if (idx == middle) {
            const int r[9] = { ec0, ec0 + 1, ec0 + 3, xe, ye, ec0 + 2, xn0, yn0, outside };
#pragma unroll
            for (int k = 0; k < 9; ++k) { rec[k] = r[k]; q[k] = r[k]; }
            rnb[0] = rkx; snb[0] = szx; rnb[1] = rky; snb[1] = szy;
        } else if (idx == outside) {
            const int r[9] = { ec0 + 2, -1, -1, ec0 + 3, -1, -1, middle, -1, -1 };
#pragma unroll
            for (int k = 0; k < 9; ++k) { rec[k] = r[k]; q[k] = r[k]; }
        } else if (idx == xn0 || idx == yn0) {      // x and y keep their slots, which now lead to `middle`
            const int slot = idx == xn0 ? xe : ye, nrev = idx == xn0 ? ec0 : ec0 + 1;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (rec[k] == slot) { rec[3 + k] = nrev; rec[6 + k] = middle; q[3 + k] = nrev; q[6 + k] = middle; }
        }
        const int rn = new_rank(idx, r_old), sn = new_size(idx, r_old, s_old);
        rk_out[idx] = rn;
        sz_out[idx] = sn;
        x.nar[rn] = idx;
        x.tix[idx] = -1;
        int dn = d_old;
        if (idx == middle) dn = dxn + 1;            // (dep[middle] = dep[x], then + 1 with the moved subtree, :400-416)
        else if (idx == outside) dn = dxn + 2;
        else if (rn >= rrk && rn <= small) dn = d_old + 1;
        if (dn != d_old || idx == middle || idx == outside) x.dep[idx] = dn;
        // parent = the neighbour with the smaller rank
        int parent = -1, psize = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (rec[k] >= 0 && new_rank(rec[6 + k], rnb[k]) < rn) { parent = rec[6 + k]; psize = new_size(rec[6 + k], rnb[k], snb[k]); }
        if (sn > sm) kind = 1;
        else if (parent < 0 || psize > sm) kind = 0;
        root_rec = make_int4(idx, rn, sn, dn);      // (what the small-subtree launches need of a root: no second round trip for it)
        if (kind >= 0) mine = atomicAdd(&s_cnt[kind], 1);
        if (idx == middle) {
            // ---- updateTreeStructure: the adjacency arrays
            if (trace) { trace[3 * i] = dflt ? 0.0 : (double)eid; trace[3 * i + 1] = dflt ? 0.0 : fracLen; trace[3 * i + 2] = dflt ? 2.0 : addLen; }
            int ec = ec0;
            p.e[xe] = middle; p.len[xe] = fracLen; p.rev[xe] = ec;
            p.e[ye] = middle; p.len[ye] -= fracLen; p.rev[ye] = ec + 1;
            p.e[ec] = xn0; p.len[ec] = fracLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle; p.rev[ec] = xe; ec++;
            p.e[ec] = yn0; p.len[ec] = originalDis - fracLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle; p.rev[ec] = ye; ec++;
            p.e[ec] = middle; p.len[ec] = addLen; p.nxt[ec] = p.head[outside]; p.head[outside] = ec; p.belong[ec] = outside; p.rev[ec] = ec + 1; ec++;
            p.e[ec] = outside; p.len[ec] = addLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle; p.rev[ec] = ec - 1; ec++;
            if (quirk) x.st->quirk = 1;      // sticky: the host repeats the run with the literal schedule
            x.st->nroot[cpar ^ 1] = 0; x.st->ntop[cpar ^ 1] = 0;      // the lists of the passes that have run: free for the next tip's
        }
    }
    // one append per block and list (a few thousand single appends to one counter cost more than the rest of the kernel)
    __syncthreads();
    if (tid < 2 && s_cnt[tid] > 0) s_base[tid] = atomicAdd(tid == 0 ? &x.st->nroot[cpar] : &x.st->ntop[cpar], s_cnt[tid]);
    __syncthreads();
    if (kind == 0) reinterpret_cast<int4*>(x.roots)[s_base[0] + mine] = root_rec;
    else if (kind == 1) { x.tops[s_base[1] + mine] = idx; x.tix[idx] = s_base[1] + mine; }      // (tix: position in the top list, what px_top_kernel's contexts refer to)
}

// ---- records of the top tree's climbing schedule (px_top_climb below)
struct __attribute__((aligned(16))) TopUp { uint32_t m; uint32_t upoff; double lenp; };      // what a climb needs of the parent
struct __attribute__((aligned(16))) TopDn { uint32_t refs; uint32_t offA, offB; uint32_t fv; };   // what the walk down needs besides
struct __attribute__((aligned(16))) TopCC { double a, b; };                                   // clipped bottom-up terms of child A / B
// TopUp::m: bits 0-12 parent's index, 13 has a parent, 14 the node is its parent's child B, 15 / 16 child A / B is a top node,
// 17 / 18 child A / B exists.  (Child A: the child that follows the node in pre-order.)  TopDn::refs: index of A | index of B << 16
// (a child outside the top tree: the spare record kTopClimb - 1); TopDn::fv: the flag bits of m, and in bit 0 the child the
// node's climb came from (written by the climb).  upoff / offA / offB: BYTE offsets into lim[] of the slot to the parent / to
// the children, the spare double at the end of lim[] for an edge that is not there (the loops store without asking).
constexpr uint32_t kUpPar = 1u << 13, kUpB = 1u << 14, kUpTopA = 1u << 15, kUpTopB = 1u << 16, kUpHasA = 1u << 17, kUpHasB = 1u << 18;
constexpr uint32_t kUpFlags = ~0x1fffu;
// the structural part of a top node's records (everything but the values of this tip): packed by spare workgroups of the
// small-subtree launch that runs before px_top_kernel, so that the single workgroup starts from ONE coalesced load instead of
// the chain top list -> node record -> neighbours' ranks / indices / lengths
struct __attribute__((aligned(16))) TopPack { TopUp u; TopDn d; double lenup, lenA, lenB; int32_t rslotA, rslotB; };
static_assert(sizeof(TopPack) == 64, "TopPack");
static_assert((size_t)kTopClimb * 64 + 2 * kXT + 16 <= kTopDynLds, "LDS records of px_top_climb");
constexpr int kPackBlocks = kTopClimb / kThreads;      // spare workgroups of px_small_up_kernel

__device__ __forceinline__ void px_top_pack(const PlaceBuffers& p, const ExactBuffers& x, const int32_t* __restrict__ rk, int t)
{
    const int v = x.tops[t];
    const int4 a = reinterpret_cast<const int4*>(x.nd + 12 * (int64_t)v)[0];
    const int4 b = reinterpret_cast<const int4*>(x.nd + 12 * (int64_t)v)[1];
    const int4 c4 = reinterpret_cast<const int4*>(x.nd + 12 * (int64_t)v)[2];
    const int slot[3] = { a.x, a.y, a.z }, rslot[3] = { a.w, b.x, b.y }, nb[3] = { b.z, b.w, c4.x };
    const int myrk = rk[v];
    TopPack k;
    const uint32_t spare = (uint32_t)((8 * p.N - 1) * 8);      // (lim[] has 8N entries, slots end below 4N)
    constexpr uint32_t kNoRef = (uint32_t)(kTopClimb - 1);
    k.u.m = 0; k.u.upoff = spare; k.u.lenp = 0.0;
    k.d.refs = kNoRef | (kNoRef << 16); k.d.offA = spare; k.d.offB = spare; k.d.fv = 0;
    k.lenup = 0.0; k.lenA = 0.0; k.lenB = 0.0; k.rslotA = -1; k.rslotB = -1;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        if (slot[e] < 0) continue;
        const int nrk = rk[nb[e]];
        const int ti = x.tix[nb[e]];
        const double len = p.len[slot[e]];
        if (nrk > myrk) {                        // a child; outside the top tree (ti < 0) its value is read from memory by px_top_climb
            const uint32_t r = ti < 0 ? kNoRef : (uint32_t)ti;
            if (nrk == myrk + 1) { k.d.offA = (uint32_t)slot[e] * 8u; k.lenA = len; k.rslotA = ti < 0 ? rslot[e] : -1; k.d.refs = (k.d.refs & 0xffff0000u) | r; k.u.m |= kUpHasA | (ti >= 0 ? kUpTopA : 0u); }
            else { k.d.offB = (uint32_t)slot[e] * 8u; k.lenB = len; k.rslotB = ti < 0 ? rslot[e] : -1; k.d.refs = (k.d.refs & 0xffffu) | (r << 16); k.u.m |= kUpHasB | (ti >= 0 ? kUpTopB : 0u); }
        } else {                                 // the parent (a top node, as its subtree is larger)
            k.u.upoff = (uint32_t)slot[e] * 8u; k.lenup = len; k.u.lenp = p.len[rslot[e]];
            k.u.m |= kUpPar | (uint32_t)ti | (myrk == nrk + 1 ? 0u : kUpB);
        }
    }
    k.d.fv = k.u.m & kUpFlags;
    reinterpret_cast<TopPack*>(x.tpack)[t] = k;
}

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// The passes inside the small subtrees.  A small subtree (at most sm = B x NPT nodes: a contiguous run of pre-order ranks) is evaluated
// by B threads with NPT nodes each, the values travelling through LDS, all small subtrees in parallel on all CUs.  sm = 64: one
// WAVEFRONT per subtree, four subtrees per workgroup, no barrier (rounds 3-6).  sm = 256 / 512 / 1 024 (round 6): one 256-thread
// WORKGROUP per subtree with a barrier per level of the subtree and 1 / 2 / 4 nodes per thread (a workgroup of 512 or 1 024 threads
// per subtree was built first: most subtrees are far below the limit, the idle threads cost the residency of the launch -- two rounds
// of workgroups instead of one) -- for trees whose top tree (the nodes above the small subtrees) would outgrow the 2 047 nodes the
// climbing schedule of px_top_kernel keeps in LDS: with 64-node subtrees that happens at ~30 000 - 60 000 tips, which is where the
// reference's command starts to use this mode at all (`-m 0` with 30 000 <= N < 1 000 000); 256 nodes divide the top tree
// by ~3.5, 1 024 by ~12 (exact_adapt picks sm from the top tree's size as the run goes).  Same recurrences, any sm: lim[] bit for bit.
// ------------------------------------------------------------------------------------------------
template <int B> struct SmallShape {
    static constexpr int kTpb = B < kThreads ? kThreads : B;      // threads per workgroup
    static constexpr int kSubs = kTpb / B;                        // subtrees a workgroup works on at a time
};
template <int B>
__device__ __forceinline__ void small_sync()
{
    if constexpr (B == 64) wave_lds_sync();
    else __syncthreads();
}
// maximum over the B threads of a subtree (s_red: one int per wavefront of the workgroup)
template <int B>
__device__ __forceinline__ int small_max(int v, int* s_red)
{
    v = wave_max_i32(v);
    if constexpr (B == 64) return v;
    else {
        constexpr int kW = SmallShape<B>::kTpb / 64;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        int m = s_red[0];
#pragma unroll
        for (int i = 1; i < kW; ++i) m = max(m, s_red[i]);
        return m;
    }
}

// bottom-up pass inside the small subtrees (updateFromBottomToTop, :296-329): a node's value = lim[node -> parent]; children's values
// come from LDS.  B threads per subtree, NPT nodes per thread (node j * B + t of the subtree's pre-order run belongs to thread t).
template <int B, int NPT>
__global__ __launch_bounds__(SmallShape<B>::kTpb) void px_small_up_kernel(PlaceBuffers p, ExactBuffers x, const double* __restrict__ dis, int par)
{
    constexpr int kTpb = SmallShape<B>::kTpb, kSubs = SmallShape<B>::kSubs;
    __shared__ double s_val[kSubs][B * NPT];
    __shared__ int s_red[kTpb / 64];
    const int t = (int)threadIdx.x % B, sub = (int)threadIdx.x / B;
    const int nroot = x.st->nroot[par];
    const int32_t* __restrict__ rk = x.rk[par];
    const int grid = (int)gridDim.x - kPackBlocks;
    if ((int)blockIdx.x >= grid) {      // the spare workgroups: structural records of the top nodes for px_top_climb
        const int T = x.st->ntop[par];
        if (T < kTopClimb)
            for (int q = ((int)blockIdx.x - grid) * kTpb + (int)threadIdx.x; q < T; q += kPackBlocks * kTpb) px_top_pack(p, x, rk, q);
        return;
    }
    for (int r = (int)blockIdx.x * kSubs + sub; r - sub < nroot; r += grid * kSubs) {      // (B > 64: the same trip count for the whole workgroup)
        const bool have = r < nroot;
        const int4 root = have ? reinterpret_cast<const int4*>(x.roots)[r] : make_int4(0, 0, 0, 0);      // node, rank, size, depth
        const int r0 = root.y, s = root.z, d0 = root.w;
        XCtx c[NPT];
        int ld[NPT], mld = -1;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int v = j * B + t < s ? x.nar[r0 + j * B + t] : -1;
            c[j] = px_ctx(x, p, dis, rk, v, r0, false);
            ld[j] = v >= 0 ? x.dep[v] - d0 : -1;
            mld = max(mld, ld[j]);
        }
        const int maxld = small_max<B>(mld, s_red);
        for (int lev = maxld; lev >= 0; --lev) {
#pragma unroll
            for (int j = 0; j < NPT; ++j)
                if (ld[j] == lev) {
                    double mx = c[j].init;
                    int up = -1;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (c[j].slot[k] >= 0) {
                            if (c[j].down[k]) { const double req = s_val[sub][c[j].ref[k]] - c[j].len[k]; if (req > mx) mx = req; }
                            else up = c[j].slot[k];
                        }
                    s_val[sub][j * B + t] = mx;
                    if (up >= 0) x.lim[up] = mx;
                }
            small_sync<B>();
        }
    }
}

// top-down pass inside the small subtrees (updateFromTopToBottom, :331-364): the value a node receives from its parent
// comes from LDS (the subtree's root: from memory, written by the top-tree kernel); children's bottom-up values from memory
template <int B, int NPT>
__global__ __launch_bounds__(SmallShape<B>::kTpb) void px_small_down_kernel(PlaceBuffers p, ExactBuffers x, const double* __restrict__ dis, int par,
                                                                            PlacePartialX* __restrict__ partials)
{
    constexpr int kTpb = SmallShape<B>::kTpb, kSubs = SmallShape<B>::kSubs;
    __shared__ double s_in[kSubs][B * NPT];
    __shared__ double s_key[kTpb / 64];
    __shared__ int s_idx[kTpb / 64], s_red[kTpb / 64];
    const int t = (int)threadIdx.x % B, sub = (int)threadIdx.x / B;
    const int nroot = x.st->nroot[par];
    const int32_t* __restrict__ rk = x.rk[par];
    // calculateBranchLength (:158-197) rides along: every node but the root owns ONE candidate, the slot parent -> node (the
    // direction the reference evaluates: the shallower end first), whose two values are final here -- the one from above has just
    // been computed (or, for the root of a small subtree and for the top nodes, written by px_top_kernel), the one from below is
    // the bottom-up pass's.  The first minimum over (pendant length, slot) does not depend on who evaluates which slot: one
    // partial per workgroup, finished by px_patch_kernel.  (Was: px_scan_kernel, a launch of its own per tip.)
    XBest best;
    px_best_init(best);
    const int grid = (int)gridDim.x - kPackBlocks;
    if ((int)blockIdx.x >= grid) {      // the spare workgroups: the top nodes' candidates
        const int T = x.st->ntop[par];
        for (int q0 = ((int)blockIdx.x - grid) * kTpb + (int)threadIdx.x; q0 < T; q0 += kPackBlocks * kTpb) {
            const int v = x.tops[q0];
            const int32_t* q = x.nd + 12 * (int64_t)v;
            const int myrk = rk[v];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (q[k] >= 0 && rk[q[6 + k]] < myrk) {      // the edge to the parent
                    const int idx = q[3 + k];
                    const double plen = p.len[idx];
                    double add, d1;
                    px_candidate(x.lim[idx], x.lim[q[k]], plen, add, d1);
                    px_best_take(best, add, d1, idx, q[6 + k], v, q[k], plen);
                }
        }
        px_best_store(best, s_key, s_idx, partials + blockIdx.x);
        return;
    }
    for (int r = (int)blockIdx.x * kSubs + sub; r - sub < nroot; r += grid * kSubs) {
        const bool have = r < nroot;
        const int4 root = have ? reinterpret_cast<const int4*>(x.roots)[r] : make_int4(0, 0, 0, 0);      // node, rank, size, depth
        const int r0 = root.y, s = root.z, d0 = root.w;
        XCtx c[NPT];
        int ld[NPT], vv[NPT], mld = -1;
        // what does not depend on this pass: lim[child -> node] of the bottom-up pass, and for the subtree's root lim[parent -> root];
        // the candidate's operands that are there already: the node's own bottom-up value and the length of the parent's slot
        double inc[NPT][3], below[NPT], plen[NPT], above0[NPT];
        int cand[NPT], cand_x[NPT], cand_rev[NPT];
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int v = j * B + t < s ? x.nar[r0 + j * B + t] : -1;
            vv[j] = v;
            c[j] = px_ctx(x, p, dis, rk, v, r0, false);
            ld[j] = v >= 0 ? x.dep[v] - d0 : -1;
            mld = max(mld, ld[j]);
            const bool is_root = j == 0 && t == 0;
            cand[j] = -1; cand_x[j] = -1; cand_rev[j] = -1; below[j] = 0.0; plen[j] = 0.0; above0[j] = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                inc[j][k] = 0.0;
                if (c[j].slot[k] >= 0 && (c[j].down[k] || is_root)) inc[j][k] = x.lim[c[j].rslot[k]];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (c[j].slot[k] >= 0 && !c[j].down[k]) {
                    cand[j] = c[j].rslot[k]; cand_rev[j] = c[j].slot[k]; cand_x[j] = p.belong[c[j].rslot[k]];
                    below[j] = x.lim[c[j].slot[k]]; plen[j] = p.len[c[j].rslot[k]]; above0[j] = inc[j][k];
                }
        }
        const int maxld = small_max<B>(mld, s_red);
        for (int lev = 0; lev <= maxld; ++lev) {
#pragma unroll
            for (int j = 0; j < NPT; ++j)
                if (ld[j] == lev) {
                    const bool is_root = j == 0 && t == 0;
                    double rq[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double in = inc[j][k];
                        if (c[j].slot[k] >= 0 && !c[j].down[k] && !is_root) in = s_in[sub][j * B + t];
                        rq[k] = c[j].slot[k] >= 0 ? in - c[j].len[k] : 0.0;
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        if (c[j].slot[a] >= 0 && c[j].down[a]) {
                            double mx = 0;
#pragma unroll
                            for (int b = 0; b < 3; ++b)
                                if (b != a && c[j].slot[b] >= 0 && rq[b] > mx) mx = rq[b];
                            x.lim[c[j].slot[a]] = mx;
                            s_in[sub][c[j].ref[a]] = mx;
                        }
                }
            small_sync<B>();
        }
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (cand[j] >= 0) {
                double add, d1;
                px_candidate((j == 0 && t == 0) ? above0[j] : s_in[sub][j * B + t], below[j], plen[j], add, d1);
                px_best_take(best, add, d1, cand[j], cand_x[j], vv[j], cand_rev[j], plen[j]);
            }
        small_sync<B>();      // (s_in is rewritten by the next subtree)
    }
    px_best_store(best, s_key, s_idx, partials + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// The top tree without level lists and without a barrier per level (round 6).  A node's bottom-up value is final once its
// children's are, its top-down values once its parent's is: every thread keeps the contexts of its (at most R) top nodes in
// registers and POLLS the LDS words of the values it waits for (a value is a maximum over terms >= 0, so -1 marks "not there
// yet"); a node that has become ready is evaluated in the same iteration, and a wavefront none of whose nodes is ready only
// runs the cheap check (two LDS reads per node) and yields its issue slots for a moment.  A level then costs the LDS round
// trips of the wavefronts at the frontier instead of a 16-wavefront barrier, and the counting sort by depth that only fed the
// level loop (maximum depth, histogram, scan, scatter: six barriers and global atomics) is gone -- a node's index is its position
// in the top list (px_patch_kernel stores it in tix).  Same recurrences, same operands and the same order of the max() terms per
// node as the level loops: lim[] bit for bit (tests/test_gpu_exact.py runs all three schedules against the CPU restatement).
// R = nodes per thread: 1 up to 1 024 top nodes, 2 up to 2 048, else 4 (the loop bodies are unrolled R times).
// ------------------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void px_top_poll(const PlaceBuffers& p, const ExactBuffers& x, const double* __restrict__ dis, const int32_t* __restrict__ rk,
                                            int T, double* up_l, double* in_l)
{
    const int tid = threadIdx.x;
    struct RCtx { int slot[3]; int rf[3]; double len[3]; double cv[3]; };     // rf: -1 no edge; else (ref + 1) << 1 | down
    RCtx rc[R];
    const unsigned long long ck0 = wall_clock64();
    unsigned long long* up_w = reinterpret_cast<unsigned long long*>(up_l);
    unsigned long long* in_w = reinterpret_cast<unsigned long long*>(in_l);
    const unsigned long long kNone = (unsigned long long)__double_as_longlong(-1.0);
    for (int t = tid; t < T; t += kXT) { up_w[t] = kNone; in_w[t] = kNone; }
#pragma unroll
    for (int m = 0; m < R; ++m) {
        const int t = tid + m * kXT;
        const XCtx c = px_ctx(x, p, dis, rk, t < T ? x.tops[t] : -1, 0, true);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rc[m].slot[k] = c.slot[k];
            rc[m].rf[k] = c.slot[k] >= 0 ? (((c.ref[k] + 1) << 1) | (c.down[k] ? 1 : 0)) : -1;
            rc[m].len[k] = c.len[k];
            rc[m].cv[k] = (c.slot[k] >= 0 && c.down[k] && c.ref[k] < 0) ? x.lim[c.rslot[k]] : 0.0;    // child outside the top tree
        }
    }
    __syncthreads();               // the marks are in place
    if (x.clk) __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long ck1 = wall_clock64();
    auto ld = [](const unsigned long long* w) { return __longlong_as_double((long long)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); };
    auto stv = [](unsigned long long* w, double v) { __hip_atomic_store(w, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    // ---- bottom-up: lim[node -> parent] = max(0, max over the child edges of lim[child -> node] - len)
    unsigned pend = 0;
#pragma unroll
    for (int m = 0; m < R; ++m) pend |= (tid + m * kXT < T ? 1u : 0u) << m;
    // (every poll is bounded: a value that never arrives -- a broken tree, a bug -- ends the launch with the stuck node on record
    //  instead of hanging the device: status word x.st->poll_fail, checked by the host after the run)
    const unsigned long long t_start = wall_clock64();
    unsigned spins = 0;
    // The loop condition is WAVE-UNIFORM (every lane stays until the whole wavefront is done): with `while (pend)` the compiler
    // turned the R = 1 loop into "poll until ready, evaluate after the loop" -- a lane that is ready then waits at the loop exit
    // for the lanes of its own wavefront that wait for ITS value: a parent and its child in one wavefront never finish
    // (seen on hardware: the two top nodes of a 66-node tree).  Lanes without a pending node idle through the iterations.
    while (__builtin_amdgcn_ballot_w64(pend != 0u) != 0ull) {
        if ((++spins & 1023u) == 0u && wall_clock64() - t_start > kTopPollTicks) {      // (scalar clock: the whole wavefront leaves)
            if (pend && atomicCAS(&x.st->poll_fail, 0, 1) == 0) { x.st->poll_node = x.tops[tid + (__builtin_ctz(pend)) * kXT]; x.st->poll_pass = 0; }
            break;
        }
        unsigned rdy = 0;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            if (!((pend >> m) & 1u)) continue;
            bool ready = true;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (rc[m].rf[k] >= 0 && (rc[m].rf[k] & 1) && (rc[m].rf[k] >> 1) > 0) {      // a child inside the top tree
                    const double v = ld(up_w + (rc[m].rf[k] >> 1) - 1);
                    rc[m].cv[k] = v;                                   // (kept: the top-down pass needs the final value again)
                    if (v < 0.0) ready = false;
                }
            if (ready) rdy |= 1u << m;
        }
        if (__builtin_amdgcn_ballot_w64(rdy != 0u) == 0ull) __builtin_amdgcn_s_sleep(1);      // (wave-uniform; no `continue`: structured flow around the ballot)
#pragma unroll
        for (int m = 0; m < R; ++m) {
            if (!((rdy >> m) & 1u)) continue;
            double mx = 0.0;
            int up = -1;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (rc[m].rf[k] >= 0) {
                    if (rc[m].rf[k] & 1) { const double req = rc[m].cv[k] - rc[m].len[k]; if (req > mx) mx = req; }
                    else up = rc[m].slot[k];
                }
            if (up >= 0) x.lim[up] = mx;
            stv(up_w + tid + m * kXT, mx);
        }
        pend &= ~rdy;
    }
    if (x.clk) __syncthreads();
    const unsigned long long ck2 = wall_clock64();
    // ---- top-down: lim[node -> child] = max(0, max over the node's other edges of lim[other -> node] - len); a node's children's
    // bottom-up values are final (its own is) and sit in cv; its parent's contribution arrives in in_l
#pragma unroll
    for (int m = 0; m < R; ++m) pend |= (tid + m * kXT < T ? 1u : 0u) << m;
    spins = 0;
    while (__builtin_amdgcn_ballot_w64(pend != 0u) != 0ull) {
        if ((++spins & 1023u) == 0u && wall_clock64() - t_start > 2 * kTopPollTicks) {
            if (pend && atomicCAS(&x.st->poll_fail, 0, 1) == 0) { x.st->poll_node = x.tops[tid + (__builtin_ctz(pend)) * kXT]; x.st->poll_pass = 1; }
            break;
        }
        unsigned rdy = 0;
        double inv[R];
#pragma unroll
        for (int m = 0; m < R; ++m) {
            inv[m] = 0.0;
            if (!((pend >> m) & 1u)) continue;
            const bool has_up = (rc[m].rf[0] >= 0 && !(rc[m].rf[0] & 1)) || (rc[m].rf[1] >= 0 && !(rc[m].rf[1] & 1)) || (rc[m].rf[2] >= 0 && !(rc[m].rf[2] & 1));
            if (has_up) inv[m] = ld(in_w + tid + m * kXT);
            if (!(inv[m] < 0.0)) rdy |= 1u << m;
        }
        if (__builtin_amdgcn_ballot_w64(rdy != 0u) == 0ull) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int m = 0; m < R; ++m) {
            if (!((rdy >> m) & 1u)) continue;
            double rq[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double in = rc[m].rf[k] >= 0 ? ((rc[m].rf[k] & 1) ? rc[m].cv[k] : inv[m]) : 0.0;
                rq[k] = rc[m].rf[k] >= 0 ? in - rc[m].len[k] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (rc[m].rf[a] >= 0 && (rc[m].rf[a] & 1)) {
                    double mx = 0;
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        if (b != a && rc[m].rf[b] >= 0 && rq[b] > mx) mx = rq[b];
                    x.lim[rc[m].slot[a]] = mx;
                    const int ref = (rc[m].rf[a] >> 1) - 1;
                    if (ref >= 0) stv(in_w + ref, mx);
                }
        }
        pend &= ~rdy;
    }
    if (x.clk) {
        __syncthreads();
        if (tid == 0) {
            const unsigned long long ck3 = wall_clock64();
            atomicAdd(&x.clk[0], ck1 - ck0); atomicAdd(&x.clk[1], ck2 - ck1); atomicAdd(&x.clk[2], ck3 - ck2);
            atomicAdd(&x.clk[3], 1ull); atomicAdd(&x.clk[4], (unsigned long long)T);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// The top tree by CLIMBING (round 6, second step; top trees of fewer than kTopClimb nodes).  px_top_poll still spends ~0.3 us
// per level: a node that becomes ready is noticed by its thread's next polling iteration, and an iteration walks over all of
// the thread's nodes (the time is the instructions of that iteration, not contention: longer sleeps of the idle wavefronts and
// s_setprio for the busy ones change nothing -- profiles/r6/exact_top_pass_variants.txt).  Here nobody waits in the bottom-up
// pass.  Every top node WITHOUT top children is the start of a climb: its thread evaluates it and moves on to the parent; at a
// parent with two top children the first to arrive leaves its contribution in the parent's LDS word (one 64-bit atomic max --
// the contributions are doubles >= 0, whose bit patterns order like integers, and the word starts at the pattern of -1.0) and
// stops, the second finds it there and goes on with the parent.  A level of the critical path costs ONE LDS round trip (the
// atomic, with the parent's record loaded beside it) and ~35 instructions.  The starts are compacted (at most half of the
// nodes: one climb per thread, wavefronts without one leave at once).  The climbs cut the top tree into vertical chains, each
// evaluated by one thread; the top-down pass hands every chain to the SAME thread again, which waits for the value its top
// node receives from above (the only polling left: one LDS word per thread, bounded as in px_top_poll) and then walks the
// chain down without waiting, leaving the values for the chains that hang off it on the way.
// Same recurrences and operands as the level loops; a maximum over terms "taken only if larger than the running maximum that
// starts at 0" (src/placement.cu:320-326, :346-358) is the maximum of the terms clipped at 0, in any order: lim[] bit for bit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void px_top_climb(const PlaceBuffers& p, const ExactBuffers& x, int T, double* s_dyn)
{
    (void)p;
    constexpr int R = kTopClimb / kXT;
    const int tid = threadIdx.x, lane = tid & 63;
    TopUp* up = reinterpret_cast<TopUp*>(s_dyn);
    TopDn* dn = reinterpret_cast<TopDn*>(up + kTopClimb);
    TopCC* cc = reinterpret_cast<TopCC*>(dn + kTopClimb);
    double* lenup = reinterpret_cast<double*>(cc + kTopClimb);        // length of the node's own slot towards the parent (operand of its top-down term)
    long long* acc = reinterpret_cast<long long*>(lenup + kTopClimb);  // bottom-up: meeting word of the children's climbs; then: the value received from the parent
    unsigned short* starts = reinterpret_cast<unsigned short*>(acc + kTopClimb);     // [kXT] nodes without top children
    int* nstart = reinterpret_cast<int*>(starts + kXT);
    const long long kSent = __double_as_longlong(-1.0);
    const unsigned long long ck0 = wall_clock64();
    if (tid == 0) *nstart = 0;
    __syncthreads();

    // ---- records of this thread's nodes: the packed structure + the values of the children outside the top tree
    const TopPack* __restrict__ pack = reinterpret_cast<const TopPack*>(x.tpack);
#pragma unroll
    for (int m = 0; m < R; ++m) {
        const int t = tid + m * kXT;
        bool is_start = false;
        if (t < T) {
            const TopPack k = pack[t];
            TopCC c; c.a = 0.0; c.b = 0.0;
            if (k.rslotA >= 0) { const double req = x.lim[k.rslotA] - k.lenA; if (req > 0.0) c.a = req; }
            if (k.rslotB >= 0) { const double req = x.lim[k.rslotB] - k.lenB; if (req > 0.0) c.b = req; }
            up[t] = k.u; dn[t] = k.d; cc[t] = c; lenup[t] = k.lenup;
            const int ntop = ((k.u.m & kUpTopA) ? 1 : 0) + ((k.u.m & kUpTopB) ? 1 : 0);
            // (one top child: the other child's term is there from the start and the climb finds it like a sibling's; none: unused)
            acc[t] = ntop == 1 ? __double_as_longlong(c.a > c.b ? c.a : c.b) : kSent;
            is_start = ntop == 0;
        }
        // the starts, compacted: one LDS counter update per wavefront
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(is_start);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(nstart, __builtin_popcountll(bal));
        base = __shfl(base, 0, 64);
        const int at = base + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        if (is_start && at < kXT) starts[at] = (unsigned short)t;      // (a binary tree of T < 2 kXT nodes has at most kXT leaves)
    }
    __syncthreads();
    if (x.clk) __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long ck1 = wall_clock64();

    // ---- bottom-up: lim[node -> parent] = max(0, max over the child edges of lim[child -> node] - len)
    if (*nstart > kXT) {      // not a binary tree: broken records
        if (tid == 0 && atomicCAS(&x.st->poll_fail, 0, 1) == 0) { x.st->poll_node = -1; x.st->poll_pass = 0; }
        return;
    }
    int n = -1;           // node the climb stands on (evaluated)
    double val = 0.0;     // its bottom-up value
    TopUp cur; cur.m = 0; cur.upoff = 0; cur.lenp = 0.0;
    char* const lim8 = reinterpret_cast<char*>(x.lim);
    auto lim_store = [&](uint32_t off, double v) { *reinterpret_cast<double*>(lim8 + off) = v; };
    if (tid < *nstart) {
        n = starts[tid];
        cur = up[n];
        const TopCC c = cc[n];
        val = c.a > c.b ? c.a : c.b;
        lim_store(cur.upoff, val);
    }
    bool act = n >= 0;
    int steps = 0;
    while (__builtin_amdgcn_ballot_w64(act) != 0ull) {
        if (++steps > kTopClimb) {                   // (a chain of parents longer than the tree: broken records -- never loop for ever)
            if (act && atomicCAS(&x.st->poll_fail, 0, 1) == 0) { x.st->poll_node = x.tops[n]; x.st->poll_pass = 0; }
            act = false;
            break;
        }
        if (act) {
            if (!(cur.m & kUpPar)) act = false;          // the root: the chain ends here
            else {
                const int par = (int)(cur.m & 0x1fffu);
                const double req = val - cur.lenp;
                const double c = req > 0.0 ? req : 0.0;
                // one LDS round trip: the atomic on the parent's meeting word and the parent's record travel together
                const long long old = __hip_atomic_fetch_max(acc + par, __double_as_longlong(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const TopUp pu = up[par];
                asm volatile("" :: "v"(old), "v"(pu.m), "v"(pu.upoff), "v"(pu.lenp));
                if (old == kSent) act = false;           // first of two: the sibling's climb goes on
                else {
                    const double o = __longlong_as_double(old);
                    const double v = __builtin_fmax(c, o);      // (both >= +0 and no NaN: the larger one)
                    const bool fromB = (cur.m & kUpB) != 0u;
                    TopCC w; w.a = fromB ? o : c; w.b = fromB ? c : o;      // (one top child: o IS the other child's term, already there)
                    cc[par] = w;
                    dn[par].fv = (pu.m & kUpFlags) | (fromB ? 1u : 0u);
                    __hip_atomic_store(acc + par, kSent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // from now on: the value from above
                    lim_store(pu.upoff, v);
                    n = par; val = v; cur = pu;
                }
            }
        }
    }
    if (x.clk) __syncthreads();
    const unsigned long long ck2 = wall_clock64();

    // ---- top-down: lim[node -> child] = max(0, max over the node's other edges of lim[other -> node] - len)
    // n: the top node of this thread's chain (-1: no chain).  The records of the node the walk stands on are in registers; those
    // of the next node of the chain are loaded while the current one is evaluated: a level costs the LDS round trip of that load.
    bool wait = n >= 0, walk = false;
    double win = 0.0;     // value the node on which the walk stands received from above
    const bool chain_has_par = (cur.m & kUpPar) != 0u;
    const int n0 = n >= 0 ? n : kTopClimb - 1;
    TopDn d = dn[n0];
    TopCC c = cc[n0];
    double lu = lenup[n0];
    const unsigned long long t_start = wall_clock64();
    unsigned spins = 0;
    while (__builtin_amdgcn_ballot_w64(wait || walk) != 0ull) {
        if ((++spins & 1023u) == 0u && wall_clock64() - t_start > kTopPollTicks) {      // (scalar clock: the whole wavefront leaves)
            if ((wait || walk) && atomicCAS(&x.st->poll_fail, 0, 1) == 0) { x.st->poll_node = x.tops[n]; x.st->poll_pass = 1; }
            break;
        }
        if (wait) {
            const long long got = chain_has_par ? __hip_atomic_load(acc + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ll;
            if (got != kSent) { win = __longlong_as_double(got); wait = false; walk = true; }
        }
        if (__builtin_amdgcn_ballot_w64(walk) == 0ull) __builtin_amdgcn_s_sleep(1);      // (wave-uniform: nobody has anything to do yet)
        if (walk) {
            // (a top node has both children; the root receives +0 over a slot of length 0: no case distinctions on the way down)
            const uint32_t m = d.fv;
            const bool viaB = (m & 1u) != 0u;
            const int ra = (int)(d.refs & 0xffffu), rb = (int)(d.refs >> 16);
            const int nn = viaB ? rb : ra;                 // (the spare record when that child is outside the top tree)
            const TopDn d2 = dn[nn];
            const TopCC c2 = cc[nn];
            const double lu2 = lenup[nn];
            const double rq = win - lu;
            const double base = rq > 0.0 ? rq : 0.0;
            const double la = __builtin_fmax(base, c.b), lb = __builtin_fmax(base, c.a);      // (all >= +0, no NaN)
            lim_store(d.offA, la);
            lim_store(d.offB, lb);
            // the value of the via child travels in a register; the chain that hangs off the other child gets its through LDS
            if ((m & (kUpTopA | kUpTopB)) == (kUpTopA | kUpTopB))
                __hip_atomic_store(acc + (viaB ? ra : rb), __double_as_longlong(viaB ? la : lb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!(m & (kUpTopA | kUpTopB))) walk = false;      // the chain's first node (where the climb started): done
            win = viaB ? lb : la;
            n = nn; d = d2; c = c2; lu = lu2;
        }
    }
    if (x.clk) {
        __syncthreads();
        if (tid == 0) {
            const unsigned long long ck3 = wall_clock64();
            atomicAdd(&x.clk[0], ck1 - ck0); atomicAdd(&x.clk[1], ck2 - ck1); atomicAdd(&x.clk[2], ck3 - ck2);
            atomicAdd(&x.clk[3], 1ull); atomicAdd(&x.clk[4], (unsigned long long)T);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the top tree: level lists by counting sort on depth (replaces stable_sort_by_key :766 + updateLevelStEd :419-434; the
// order inside a level is irrelevant to every result), then both passes level by level, one workgroup barrier per level;
// values in LDS when the top tree fits (else in memory, as rounds 1-2 did for the whole tree)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kXT) void px_top_kernel(PlaceBuffers p, ExactBuffers x, const double* __restrict__ dis, int par)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];      // [2][kTopLds]: bottom-up value, value received from the parent
    __shared__ int s_red[kXT / 64], s_maxdep;
    __shared__ int s_lv[kTopLevLds + 2];                                // level offsets (lists of at most kTopLevLds levels: else in memory)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int T = x.st->ntop[par];
    if (T == 0 || x.st->poll_fail) return;      // (a poll of an earlier tip ran into its bound: the run has failed, its launches drain)
    const int32_t* __restrict__ rk = x.rk[par];
    const bool lds = T <= kTopLds && !x.top_in_memory;
    double* up_l = s_dyn;
    double* in_l = s_dyn + kTopLds;
    if (lds && T < kTopClimb && !x.top_levels && !x.top_poll) {
        // ---- round 6, the common case: no level lists, no barrier per level, no waiting in the bottom-up pass (px_top_climb above)
        px_top_climb(p, x, T, s_dyn);
        return;
    }
    if (lds && T <= kTopReg * kXT && !x.top_levels) {
        // ---- larger top trees: NO level lists and NO barrier per level (px_top_poll above)
        if (T <= kXT) px_top_poll<1>(p, x, dis, rk, T, up_l, in_l);
        else if (T <= 2 * kXT) px_top_poll<2>(p, x, dis, rk, T, up_l, in_l);
        else px_top_poll<kTopReg>(p, x, dis, rk, T, up_l, in_l);
        return;
    }
    int mymax = 0;
    for (int t = tid; t < T; t += kXT) mymax = max(mymax, x.dep[x.tops[t]]);
    mymax = wave_max_i32(mymax);
    if (lane == 0) s_red[w] = mymax;
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int k = 0; k < kXT / 64; ++k) m = max(m, s_red[k]);
        s_maxdep = m;
    }
    __syncthreads();
    const int maxdep = s_maxdep;
    const bool lvl = maxdep + 2 <= kTopLevLds + 2;                       // histogram / level offsets in LDS
    int* hist = lvl ? s_lv : x.hist;
    for (int k = tid; k <= maxdep + 1; k += kXT) hist[k] = 0;
    __syncthreads();
    for (int t = tid; t < T; t += kXT) atomicAdd(&hist[x.dep[x.tops[t]]], 1);
    __syncthreads();
    {
        const int nlev = maxdep + 1;
        const int per = (nlev + kXT - 1) / kXT;
        const int b0 = tid * per, b1 = min(nlev, b0 + per);
        int sum = 0;
        for (int k = b0; k < b1; ++k) sum += hist[k];
        // exclusive scan of the per-thread sums over the workgroup: wave scans + the 16 wave totals (thread 0 walking all
        // 1 024 LDS entries one after the other, as the literal kernel does, was ~100 us of this kernel's 128)
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_red[w] = incl;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int k = 0; k < kXT / 64; ++k) { const int v = s_red[k]; s_red[k] = run; run += v; }
        }
        __syncthreads();
        int run = s_red[w] + incl - sum;
        for (int k = b0; k < b1; ++k) { const int v = hist[k]; x.lvoff[k] = run; hist[k] = run; run += v; }
        if (tid == 0) x.lvoff[nlev] = T;
    }
    __syncthreads();
    for (int t = tid; t < T; t += kXT) {
        const int idx = x.tops[t];
        const int pos = atomicAdd(&hist[x.dep[idx]], 1);
        x.order[pos] = idx;
        x.tix[idx] = pos;
    }
    __syncthreads();
    // after the scatter hist[k] = first index of level k + 1: the level offsets without another pass (LDS copy)
    auto lv0 = [&](int j) { return j == 0 ? 0 : (lvl ? s_lv[j - 1] : x.lvoff[j]); };
    auto lv1 = [&](int j) { return lvl ? s_lv[j] : x.lvoff[j + 1]; };
    // context of this thread's first node of a level, loaded ONE LEVEL AHEAD of its use: what a level then costs is an LDS
    // round trip and the barrier.  The values of children outside the top tree (small-subtree roots: written by the small
    // bottom-up kernel before this launch) are loaded with it.
    struct TCtx { XCtx c; double cv[3]; };
    auto load = [&](int j) {
        TCtx r;
        const int t = lv0(j) + tid;
        r.c = px_ctx(x, p, dis, rk, t < lv1(j) ? x.order[t] : -1, 0, true);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            r.cv[k] = (r.c.slot[k] >= 0 && r.c.down[k] && !(lds && r.c.ref[k] >= 0)) ? x.lim[r.c.rslot[k]] : 0.0;
        return r;
    };
    auto up_node = [&](const XCtx& c, const double* cv, int t, bool fresh) {
        double mx = c.init;
        int up = -1;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (c.slot[k] >= 0) {
                if (c.down[k]) {
                    double v;
                    if (lds && c.ref[k] >= 0) v = up_l[c.ref[k]];
                    else v = fresh ? x.lim[c.rslot[k]] : cv[k];       // memory mode: a top child's value is one level old -> read now
                    const double req = v - c.len[k];
                    if (req > mx) mx = req;
                } else up = c.slot[k];
            }
        if (lds) up_l[t] = mx;
        if (up >= 0) x.lim[up] = mx;
    };
    auto down_node = [&](const XCtx& c, const double* cv, int t) {
        double rq[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double in = 0.0;
            if (c.slot[k] >= 0) {
                if (!c.down[k]) in = lds ? in_l[t] : x.lim[c.rslot[k]];
                else in = (lds && c.ref[k] >= 0) ? up_l[c.ref[k]] : cv[k];      // bottom-up values are final by now
            }
            rq[k] = c.slot[k] >= 0 ? in - c.len[k] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (c.slot[a] >= 0 && c.down[a]) {
                double mx = 0;
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    if (b != a && c.slot[b] >= 0 && rq[b] > mx) mx = rq[b];
                x.lim[c.slot[a]] = mx;
                if (lds && c.ref[a] >= 0) in_l[c.ref[a]] = mx;
            }
    };
    if (lds && T <= kTopReg * kXT) {
        // ---- the common case: every thread keeps the contexts of its (at most kTopReg) top nodes in registers for both
        // passes, all their loads in flight together up front.  Loading a level's contexts inside the level loop -- even one
        // level ahead -- left a chain of three dependent memory round trips per level (list entry -> node record -> the
        // neighbours' ranks / indices / lengths): 1.15 us per level, 118 us per tip at 30 000 tips.  A level now costs LDS
        // traffic and the barrier.  (A top node is never a leaf: its start value is 0.)
        struct RCtx { int slot[3]; int rf[3]; double len[3]; double cv[3]; };     // rf: -1 no edge; else (ref + 1) << 1 | down
        RCtx rc[kTopReg];
#pragma unroll
        for (int m = 0; m < kTopReg; ++m) {
            const int t = tid + m * kXT;
            const XCtx c = px_ctx(x, p, dis, rk, t < T ? x.order[t] : -1, 0, true);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                rc[m].slot[k] = c.slot[k];
                rc[m].rf[k] = c.slot[k] >= 0 ? (((c.ref[k] + 1) << 1) | (c.down[k] ? 1 : 0)) : -1;
                rc[m].len[k] = c.len[k];
                rc[m].cv[k] = (c.slot[k] >= 0 && c.down[k] && c.ref[k] < 0) ? x.lim[c.rslot[k]] : 0.0;    // child outside the top tree
            }
        }
        for (int j = maxdep; j >= 0; --j) {
            const int t0 = lv0(j), t1 = lv1(j);
#pragma unroll
            for (int m = 0; m < kTopReg; ++m) {
                const int t = tid + m * kXT;
                if (t >= t0 && t < t1) {
                    double mx = 0.0;
                    int up = -1;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (rc[m].rf[k] >= 0) {
                            if (rc[m].rf[k] & 1) {
                                const int ref = (rc[m].rf[k] >> 1) - 1;
                                const double req = (ref >= 0 ? up_l[ref] : rc[m].cv[k]) - rc[m].len[k];
                                if (req > mx) mx = req;
                            } else up = rc[m].slot[k];
                        }
                    up_l[t] = mx;
                    if (up >= 0) x.lim[up] = mx;
                }
            }
            __syncthreads();
        }
        for (int j = 0; j <= maxdep; ++j) {
            const int t0 = lv0(j), t1 = lv1(j);
#pragma unroll
            for (int m = 0; m < kTopReg; ++m) {
                const int t = tid + m * kXT;
                if (t >= t0 && t < t1) {
                    double rq[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double in = 0.0;
                        if (rc[m].rf[k] >= 0) {
                            const int ref = (rc[m].rf[k] >> 1) - 1;
                            if (!(rc[m].rf[k] & 1)) in = in_l[t];
                            else in = ref >= 0 ? up_l[ref] : rc[m].cv[k];
                        }
                        rq[k] = rc[m].rf[k] >= 0 ? in - rc[m].len[k] : 0.0;
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        if (rc[m].rf[a] >= 0 && (rc[m].rf[a] & 1)) {
                            double mx = 0;
#pragma unroll
                            for (int b = 0; b < 3; ++b)
                                if (b != a && rc[m].rf[b] >= 0 && rq[b] > mx) mx = rq[b];
                            x.lim[rc[m].slot[a]] = mx;
                            const int ref = (rc[m].rf[a] >> 1) - 1;
                            if (ref >= 0) in_l[ref] = mx;
                        }
                }
            }
            __syncthreads();
        }
        return;
    }
    // ---- bottom-up: lim[node -> parent] = max(init, max over the child edges of lim[child -> node] - len)
    {
        TCtx cur = load(maxdep);
        for (int j = maxdep; j >= 0; --j) {
            const int t0 = lv0(j), t1 = lv1(j);
            TCtx nxt;
            nxt.c.v = -1;
            if (j > 0) nxt = load(j - 1);
            if (cur.c.v >= 0) up_node(cur.c, cur.cv, t0 + tid, !lds);
            for (int t = t0 + tid + kXT; t < t1; t += kXT) {
                const XCtx c = px_ctx(x, p, dis, rk, x.order[t], 0, true);
                double cv[3] = { 0.0, 0.0, 0.0 };
                up_node(c, cv, t, true);
            }
            __syncthreads();
            cur = nxt;
        }
    }
    // ---- top-down: lim[node -> child] = max(0, max over the node's other edges of lim[other -> node] - len)
    {
        auto load_dn = [&](int j) {      // children's bottom-up values are final now: all of them can come with the context
            TCtx r;
            const int t = lv0(j) + tid;
            r.c = px_ctx(x, p, dis, rk, t < lv1(j) ? x.order[t] : -1, 0, true);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                r.cv[k] = (r.c.slot[k] >= 0 && r.c.down[k] && !(lds && r.c.ref[k] >= 0)) ? x.lim[r.c.rslot[k]] : 0.0;
            return r;
        };
        TCtx cur = load_dn(0);
        for (int j = 0; j <= maxdep; ++j) {
            const int t0 = lv0(j), t1 = lv1(j);
            TCtx nxt;
            nxt.c.v = -1;
            if (j < maxdep) nxt = load_dn(j + 1);
            if (cur.c.v >= 0) down_node(cur.c, cur.cv, t0 + tid);
            for (int t = t0 + tid + kXT; t < t1; t += kXT) {
                const XCtx c = px_ctx(x, p, dis, rk, x.order[t], 0, true);
                double cv[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) cv[k] = (c.slot[k] >= 0 && c.down[k]) ? x.lim[c.rslot[k]] : 0.0;
                down_node(c, cv, t);
            }
            __syncthreads();
            cur = nxt;
        }
    }
}

// initialize (src/placement.cu:119-140)
__global__ __launch_bounds__(kThreads) void px_init_kernel(PlaceBuffers p, ExactBuffers x, int64_t lim, int64_t nodes)
{
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx < lim) { p.nxt[idx] = -1; p.e[idx] = -1; p.belong[idx] = -1; p.rev[idx] = -1; }
    if (idx < nodes) { p.head[idx] = -1; x.dep[idx] = (int)(nodes * 10); x.rk[0][idx] = -1; x.rk[1][idx] = -1; x.sz[0][idx] = 0; x.sz[1][idx] = 0; x.tix[idx] = -1; }
}

// buildInitialTree (src/placement.cu:245-293) from the row of tip 1; ranks, sizes and lists of the three-node tree
__global__ void px_init_tree_kernel(PlaceBuffers p, ExactBuffers x, const double* __restrict__ dis_tree)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int nv = (int)p.N;
    const double d = dis_tree[0];
    int ec = 0;
    p.e[ec] = nv; p.len[ec] = d / 2; p.nxt[ec] = p.head[0]; p.head[0] = ec; p.belong[ec] = 0; ec++;
    p.e[ec] = nv; p.len[ec] = d / 2; p.nxt[ec] = p.head[1]; p.head[1] = ec; p.belong[ec] = 1; ec++;
    p.e[ec] = 0;  p.len[ec] = d / 2; p.nxt[ec] = p.head[nv]; p.head[nv] = ec; p.belong[ec] = nv; ec++;
    p.e[ec] = 1;  p.len[ec] = d / 2; p.nxt[ec] = p.head[nv]; p.head[nv] = ec; p.belong[ec] = nv; ec++;
    p.rev[0] = 2; p.rev[2] = 0; p.rev[1] = 3; p.rev[3] = 1;
    x.dep[nv] = 0; x.dep[0] = 1; x.dep[1] = 1;
    // (the state before tip 2 is placed: buffers of parity 2 & 1 = 0)
    x.rk[0][nv] = 0; x.rk[0][0] = 1; x.rk[0][1] = 2;
    x.sz[0][nv] = 3; x.sz[0][0] = 1; x.sz[0][1] = 1;
    x.nar[0] = nv; x.nar[1] = 0; x.nar[2] = 1;
    px_set_node(x, 0, 0, 2, nv, -1, -1, -1, -1, -1, -1);
    px_set_node(x, 1, 1, 3, nv, -1, -1, -1, -1, -1, -1);
    px_set_node(x, nv, 2, 0, 0, 3, 1, 1, -1, -1, -1);
    XStep st;
    st.nroot[0] = 1; st.ntop[0] = 0; st.nroot[1] = 0; st.ntop[1] = 0;      // (the passes for tip 2 read parity 0)
    st.poll_fail = 0; st.poll_node = -1; st.poll_pass = 0;
    st.quirk = 0;
    reinterpret_cast<int4*>(x.roots)[0] = make_int4(nv, 0, 3, 0);
    *x.st = st;
}

int exact_alloc(ExactBuffers& x, int64_t N)
{
    exact_free(x);
    DPR_HIP(hipMalloc(&x.lim, sizeof(double) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&x.dep, sizeof(int32_t) * (size_t)(2 * N)));
    for (int k = 0; k < 2; ++k) {
        DPR_HIP(hipMalloc(&x.rk[k], sizeof(int32_t) * (size_t)(2 * N)));
        DPR_HIP(hipMalloc(&x.sz[k], sizeof(int32_t) * (size_t)(2 * N)));
    }
    DPR_HIP(hipMalloc(&x.nar, sizeof(int32_t) * (size_t)(2 * N + 2)));
    DPR_HIP(hipMalloc(&x.tix, sizeof(int32_t) * (size_t)(2 * N)));
    DPR_HIP(hipMalloc(&x.roots, sizeof(int32_t) * 4 * (size_t)(2 * N)));
    DPR_HIP(hipMalloc(&x.tops, sizeof(int32_t) * (size_t)(2 * N)));
    DPR_HIP(hipMalloc(&x.order, sizeof(int32_t) * (size_t)(2 * N)));
    DPR_HIP(hipMalloc(&x.lvoff, sizeof(int32_t) * (size_t)(2 * N + 2)));
    DPR_HIP(hipMalloc(&x.hist, sizeof(int32_t) * (size_t)(2 * N + 2)));
    DPR_HIP(hipMalloc(&x.nd, sizeof(int32_t) * (size_t)(12 * 2 * N)));
    x.dfsrk = x.rk[0];
    x.top_in_memory = std::getenv("DPR_EXACT_TOP_MEM") != nullptr;
    x.top_levels = std::getenv("DPR_EXACT_TOP_LEVELS") != nullptr;
    x.top_poll = std::getenv("DPR_EXACT_TOP_POLL") != nullptr;
    x.sm = 64; x.pass_sm = 64; x.sm_forced = 0;
    if (const char* e = std::getenv("DPR_EXACT_SM")) {
        const int v = std::atoi(e);
        if (v == 64 || v == 256 || v == 512 || v == 1024) { x.sm = v; x.sm_forced = v; }
    }
    if (std::getenv("DPR_EXACT_CLOCKS")) { DPR_HIP(hipMalloc(&x.clk, 8 * sizeof(unsigned long long))); DPR_HIP(hipMemset(x.clk, 0, 8 * sizeof(unsigned long long))); }
    DPR_HIP(hipMalloc(&x.tpack, sizeof(TopPack) * (size_t)kTopClimb));
    DPR_HIP(hipMalloc(&x.st, sizeof(XStep)));
    DPR_HIP(hipMemset(x.st, 0, sizeof(XStep)));
    {   // literal schedule: one partial per workgroup of px_scan_kernel; fast schedule: one per workgroup of px_small_down_kernel
        const size_t scan = (size_t)((4 * N + kThreads - 1) / kThreads + 1), down = (size_t)(2048 + kPackBlocks + 1);
        DPR_HIP(hipMalloc(&x.partials, sizeof(PlacePartialX) * (scan > down ? scan : down)));
    }
    DPR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(px_top_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kTopDynLds));
    return DPR_OK;
}

void exact_free(ExactBuffers& x)
{
    if (x.clk) {      // profiling: where px_top_kernel's time went, per tip
        unsigned long long h[8] = { 0 };
        if (hipMemcpy(h, x.clk, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[3] > 0)
            std::fprintf(stderr, "[exact] px_top_kernel over %llu tips: %.1f top nodes; contexts %.2f us, bottom-up %.2f us, top-down %.2f us per tip\n", h[3],
                         (double)h[4] / (double)h[3], h[0] * 0.01 / h[3], h[1] * 0.01 / h[3], h[2] * 0.01 / h[3]);
    }
    void* ptrs[] = { x.lim, x.dep, x.rk[0], x.rk[1], x.sz[0], x.sz[1], x.nar, x.tix, x.roots, x.tops, x.order, x.lvoff, x.hist, x.partials, x.nd, x.st, x.clk, x.tpack };
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    const bool literal = x.literal;
    x = ExactBuffers();
    x.literal = literal;
}

// the two passes for the tip whose distance row is `dis`, on the tree as it stands (rank / size buffers of parity `par`)
static int64_t exact_pass_grid(int64_t placed_nodes, int sm)
{
    // one wavefront (sm = 64: four per workgroup) or workgroup per sm / 4 nodes: about as many as there are small subtrees (a workgroup
    // that gets a second subtree works on it after the first)
    const int64_t subs = sm < kThreads ? kThreads / sm : 1;                  // subtrees a workgroup works on at a time
    const int64_t g = (4 * placed_nodes + sm * subs - 1) / (sm * subs);
    return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}
static int exact_passes(PlaceBuffers& p, ExactBuffers& x, const double* dis, int par, int64_t placed_nodes, hipStream_t s)
{
    // (the lists these passes read were made by the patch launch before them, with x.sm nodes per small subtree)
    const int64_t g = exact_pass_grid(placed_nodes, x.sm);
    PlacePartialX* parts = reinterpret_cast<PlacePartialX*>(x.partials);
    const dim3 grid((unsigned)g + kPackBlocks);
    if (x.sm == 64) hipLaunchKernelGGL((px_small_up_kernel<64, 1>), grid, dim3(SmallShape<64>::kTpb), 0, s, p, x, dis, par);
    else if (x.sm == 256) hipLaunchKernelGGL((px_small_up_kernel<256, 1>), grid, dim3(256), 0, s, p, x, dis, par);
    else if (x.sm == 512) hipLaunchKernelGGL((px_small_up_kernel<256, 2>), grid, dim3(256), 0, s, p, x, dis, par);
    else hipLaunchKernelGGL((px_small_up_kernel<256, 4>), grid, dim3(256), 0, s, p, x, dis, par);
    hipLaunchKernelGGL(px_top_kernel, dim3(1), dim3(kXT), kTopDynLds, s, p, x, dis, par);
    if (x.sm == 64) hipLaunchKernelGGL((px_small_down_kernel<64, 1>), grid, dim3(SmallShape<64>::kTpb), 0, s, p, x, dis, par, parts);
    else if (x.sm == 256) hipLaunchKernelGGL((px_small_down_kernel<256, 1>), grid, dim3(256), 0, s, p, x, dis, par, parts);
    else if (x.sm == 512) hipLaunchKernelGGL((px_small_down_kernel<256, 2>), grid, dim3(256), 0, s, p, x, dis, par, parts);
    else hipLaunchKernelGGL((px_small_down_kernel<256, 4>), grid, dim3(256), 0, s, p, x, dis, par, parts);
    x.pass_sm = x.sm;
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// initialize + buildInitialTree (row of tip 1) + the passes for tip 2 (its row)
int exact_init(PlaceBuffers& p, ExactBuffers& x, const double* d_dis_row1, const double* d_dis_row2, bool has_tip2,
               hipStream_t s)
{
    const int64_t lim = 4 * p.N - 4, nodes = 2 * p.N - 1;
    DPR_HIP(hipMemsetAsync(x.lim, 0, sizeof(double) * (size_t)(8 * p.N), s));
    hipLaunchKernelGGL(px_init_kernel, dim3((unsigned)((lim + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p, x, lim, nodes);
    if (x.literal) {
        hipLaunchKernelGGL(px_step_literal_kernel, dim3(1), dim3(kXT), 0, s, p, x, (const PlacePartialX*)nullptr, 0, (int64_t)1,
                           d_dis_row1, d_dis_row2, has_tip2 ? 1 : 0, (double*)nullptr);
        DPR_HIP(hipGetLastError());
        return DPR_OK;
    }
    hipLaunchKernelGGL(px_init_tree_kernel, dim3(1), dim3(64), 0, s, p, x, d_dis_row1);
    DPR_HIP(hipGetLastError());
    if (has_tip2) return exact_passes(p, x, d_dis_row2, 0, 3, s);
    return DPR_OK;
}

// place tip `tip` (its passes were run by the previous step) and run the passes for tip+1
int exact_tip(PlaceBuffers& p, ExactBuffers& x, int64_t tip, const double* d_dis_next, bool has_next, double* d_trace,
              hipStream_t s)
{
    const int64_t live = 4 * tip - 4;
    int nblk = (int)((live + kThreads - 1) / kThreads);
    PlacePartialX* parts = reinterpret_cast<PlacePartialX*>(x.partials);
    if (x.literal) {
        hipLaunchKernelGGL(px_scan_kernel, dim3((unsigned)nblk), dim3(kThreads), 0, s, p, x, tip, parts);
        hipLaunchKernelGGL(px_step_literal_kernel, dim3(1), dim3(kXT), 0, s, p, x, (const PlacePartialX*)parts, nblk, tip,
                           (const double*)nullptr, d_dis_next, has_next ? 1 : 0, d_trace);
        DPR_HIP(hipGetLastError());
        return DPR_OK;
    }
    // the candidates of this tip were evaluated by the top-down pass that ran for it (exact_passes of the previous step): one partial
    // per workgroup of that launch
    nblk = (int)exact_pass_grid(2 * tip - 1, x.pass_sm) + kPackBlocks;
    const int64_t tot = p.N + tip;
    hipLaunchKernelGGL(px_patch_kernel, dim3((unsigned)((tot + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p, x, (const PlacePartialX*)parts, nblk, tip,
                       d_trace, x.sm);
    DPR_HIP(hipGetLastError());
    if (has_next) return exact_passes(p, x, d_dis_next, (int)((tip + 1) & 1), 2 * tip + 1, s);
    return DPR_OK;
}

// Called between batches of tips (the stream is drained: a few microseconds per 256 tips).  The top tree -- the nodes above the small
// subtrees -- must stay below kTopClimb nodes for the climbing schedule of px_top_kernel; it grows by at most two nodes per tip.  The
// small subtrees grow with the run (64 -> 256 -> 512 -> 1 024 nodes: one workgroup each instead of one wavefront), which divides the
// top tree by ~3.5, ~2, ~2; beyond that px_top_kernel's other schedules take over.
int exact_adapt(ExactBuffers& x, hipStream_t s)
{
    if (x.literal || !x.st || x.sm_forced || x.sm >= 1024) return DPR_OK;
    XStep st;
    DPR_HIP(hipMemcpyAsync(&st, x.st, sizeof(XStep), hipMemcpyDeviceToHost, s));
    DPR_HIP(hipStreamSynchronize(s));
    const int ntop = st.ntop[0] > st.ntop[1] ? st.ntop[0] : st.ntop[1];      // (the list of the other parity is empty)
    // 64 -> 256 early (a top tree of 100 nodes, ~2 000 tips): from there on the workgroup-sized subtrees cost less in their two launches
    // than they save in px_top_kernel (30 000 tips: 47.5 us per tip against 53.8 with 64-node subtrees throughout; 128-node subtrees:
    // 48.6); 256 -> 512 -> 1 024 late (600 nodes before the limit: a batch of 256 tips adds at most 512): those steps cost more than they save
    if (x.sm == 64 ? ntop > 100 : ntop > kTopClimb - 600) x.sm = x.sm == 64 ? 256 : 2 * x.sm;
    return DPR_OK;
}

int exact_quirk(ExactBuffers& x, hipStream_t s, bool* quirk)
{
    *quirk = false;
    if (x.literal || !x.st) return DPR_OK;
    XStep st;
    DPR_HIP(hipMemcpyAsync(&st, x.st, sizeof(XStep), hipMemcpyDeviceToHost, s));
    DPR_HIP(hipStreamSynchronize(s));
    *quirk = st.quirk != 0;
    if (st.poll_fail) {
        set_error("exact placement: the top-tree pass waited for a value that never arrived (node " + std::to_string(st.poll_node) + ", " +
                  (st.poll_pass ? "top-down" : "bottom-up") + " pass): internal error; DPR_EXACT_TOP_LEVELS=1 selects the level-by-level schedule");
        return DPR_ERR_HIP;
    }
    return DPR_OK;
}

}  // namespace dpr
