// Exact placement mode on gfx950.  Replaces PlacementDeviceArrays::findPlacementTree
// (src/placement.cu:508-789) and its kernels: initialize :119-140, buildInitialTree :245-293,
// updateFromBottomToTop :296-329, updateFromTopToBottom :331-364, calculateBranchLength :158-197 +
// thrust::min_element :688, updateTreeStructure :199-243, updateDfsRk :366-379, findEndRk :382-398 +
// thrust::reduce :746, updateDepth :400-416, stable_sort_by_key :766, updateLevelStEd :419-434.
//
// Per tip the reference launches one kernel per tree level (twice), a Thrust reduction, a Thrust
// stable sort of all node depths and four blocking device->host copies.  Here per tip:
//   px_scan_kernel  (many blocks)  candidates of all live slots -> block-level first minima
//   px_step_kernel  (one 1024-thread workgroup, everything that is inherently sequential):
//       finish the argmin, split the edge, patch DFS ranks / depths (the reference's O(N) parallel
//       scheme, unchanged), rebuild the level lists by a counting sort on depth (the order inside a
//       level does not influence any result), then run the level-synchronous bottom-up / top-down
//       pass for the NEXT tip with workgroup barriers between levels instead of kernel launches.
// lim[], depths and level lists live in HBM/L2 (O(N) per tip); nothing returns to the host.
#include <cstdlib>

#include "dpr_internal.hpp"

namespace dpr {

constexpr int kXT = 1024;   // threads of the single-workgroup step kernel

struct PlacePartialX { double add; int32_t idx; int32_t eid; double frac; };

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

// calculateBranchLength over the live slots + block-level first minimum
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
        double dis1 = x.lim[eid], dis2 = x.lim[p.rev[eid]];
        const double L = p.len[eid];
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
        partials[blockIdx.x] = pp;
    }
    if (threadIdx.x == 0 && sidx[0] == 0x7fffffff) {
        PlacePartialX pp; pp.add = __builtin_inf(); pp.idx = 0x7fffffff; pp.eid = 0; pp.frac = 0;
        partials[blockIdx.x] = pp;
    }
}

// dis_tree != nullptr: build the initial two-tip tree from it (row of tip 1), then the passes for tip 2.
// otherwise: place `tip` from the scan partials, patch ranks/depths/levels, passes for tip+1 (dis_next).
__global__ __launch_bounds__(kXT) void px_step_kernel(PlaceBuffers p, ExactBuffers x,
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

// initialize (src/placement.cu:119-140)
__global__ __launch_bounds__(kThreads) void px_init_kernel(PlaceBuffers p, ExactBuffers x, int64_t lim, int64_t nodes)
{
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx < lim) { p.nxt[idx] = -1; p.e[idx] = -1; p.belong[idx] = -1; p.rev[idx] = -1; }
    if (idx < nodes) { p.head[idx] = -1; x.dep[idx] = (int)(nodes * 10); x.dfsrk[idx] = -1; }
}

int exact_alloc(ExactBuffers& x, int64_t N)
{
    exact_free(x);
    DPR_HIP(hipMalloc(&x.lim, sizeof(double) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&x.dep, sizeof(int32_t) * (size_t)(2 * N)));
    DPR_HIP(hipMalloc(&x.dfsrk, sizeof(int32_t) * (size_t)(2 * N)));
    DPR_HIP(hipMalloc(&x.order, sizeof(int32_t) * (size_t)(2 * N)));
    DPR_HIP(hipMalloc(&x.lvoff, sizeof(int32_t) * (size_t)(2 * N + 2)));
    DPR_HIP(hipMalloc(&x.hist, sizeof(int32_t) * (size_t)(2 * N + 2)));
    DPR_HIP(hipMalloc(&x.nd, sizeof(int32_t) * (size_t)(12 * 2 * N)));
    DPR_HIP(hipMalloc(&x.partials, sizeof(PlacePartialX) * (size_t)((4 * N + kThreads - 1) / kThreads + 1)));
    return DPR_OK;
}

void exact_free(ExactBuffers& x)
{
    void* ptrs[] = { x.lim, x.dep, x.dfsrk, x.order, x.lvoff, x.hist, x.partials, x.nd };
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    x = ExactBuffers();
}

// initialize + buildInitialTree (row of tip 1) + the passes for tip 2 (its row)
int exact_init(PlaceBuffers& p, ExactBuffers& x, const double* d_dis_row1, const double* d_dis_row2, bool has_tip2,
               hipStream_t s)
{
    const int64_t lim = 4 * p.N - 4, nodes = 2 * p.N - 1;
    DPR_HIP(hipMemsetAsync(x.lim, 0, sizeof(double) * (size_t)(8 * p.N), s));
    hipLaunchKernelGGL(px_init_kernel, dim3((unsigned)((lim + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p, x, lim, nodes);
    hipLaunchKernelGGL(px_step_kernel, dim3(1), dim3(kXT), 0, s, p, x, (const PlacePartialX*)nullptr, 0, (int64_t)1,
                       d_dis_row1, d_dis_row2, has_tip2 ? 1 : 0, (double*)nullptr);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// place tip `tip` (its passes were run by the previous step) and run the passes for tip+1
int exact_tip(PlaceBuffers& p, ExactBuffers& x, int64_t tip, const double* d_dis_next, bool has_next, double* d_trace,
              hipStream_t s)
{
    const int64_t live = 4 * tip - 4;
    const int nblk = (int)((live + kThreads - 1) / kThreads);
    PlacePartialX* parts = reinterpret_cast<PlacePartialX*>(x.partials);
    hipLaunchKernelGGL(px_scan_kernel, dim3((unsigned)nblk), dim3(kThreads), 0, s, p, x, tip, parts);
    hipLaunchKernelGGL(px_step_kernel, dim3(1), dim3(kXT), 0, s, p, x, (const PlacePartialX*)parts, nblk, tip,
                       (const double*)nullptr, d_dis_next, has_next ? 1 : 0, d_trace);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

}  // namespace dpr
