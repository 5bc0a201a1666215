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
#include "dpr_internal.hpp"

namespace dpr {

constexpr int K5 = 5;

struct PlacePartial { double add; int32_t idx; int32_t eid; double frac; };

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

// updateClosestNodes, one wave: frontier entries l..r processed 64 at a time.
__device__ void closest_update_wave(const PlaceBuffers& p, int x)
{
    const int lane = threadIdx.x & 63;
    int l = 0, r = 1;  // queue [l, r)
    if (lane == 0) { p.q_id[0] = x; p.q_dis[0] = 0.0; p.q_from[0] = -1; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    while (l < r) {
        const int cnt = min(64, r - l);
        int node = -1, fb = -1;
        double d = 0.0;
        if (lane < cnt) { node = p.q_id[l + lane]; fb = p.q_from[l + lane]; d = p.q_dis[l + lane]; }
        // each lane walks the out-edges of its node (pass 1: insert, remember which edges took it)
        int nnew = 0;
        unsigned long long took = 0ull;
        if (lane < cnt) {
            int pos = 0;
            for (int i = p.head[node]; i != -1; i = p.nxt[i], ++pos) {
                if (p.e[i] == fb) continue;
                bool ins;
                list_insert(p.cdis, p.cid, i, x, d, ins);
                if (ins && pos < 64) { took |= 1ull << pos; nnew++; }
            }
        }
        // append in lane order (order is irrelevant for the result, kept deterministic anyway)
        int incl = nnew;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        const int total = __shfl(incl, 63, 64);
        if (nnew) {
            int w = r + incl - nnew, pos = 0;
            for (int i = p.head[node]; i != -1; i = p.nxt[i], ++pos)
                if (pos < 64 && ((took >> pos) & 1ull)) {
                    p.q_id[w] = p.e[i]; p.q_dis[w] = d + p.len[i]; p.q_from[w] = node; ++w;
                }
        }
        l += cnt;
        r += total;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
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
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    closest_update_wave(p, 0);
    closest_update_wave(p, 1);
}

// closest lists of an imported backbone: leaves 0..m-1 in order (src/placement_close_k.cu:247-260)
__global__ __launch_bounds__(64) void place_backbone_lists_kernel(PlaceBuffers p, int64_t t0, int64_t t1)
{
    for (int64_t t = t0; t < t1; ++t) closest_update_wave(p, (int)t);
}

// calculateBranchLength for live slots idx < 4*num-4 and block-level first-minimum
__global__ __launch_bounds__(kThreads) void place_scan_kernel(PlaceBuffers p, const double* __restrict__ dis,
                                                              int64_t num, PlacePartial* __restrict__ partials)
{
    __shared__ double sadd[kThreads / 64];
    __shared__ int sidx[kThreads / 64];
    const int64_t live = 4 * num - 4;
    const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    double add = 2.0, d1 = 0.0;
    int eid = 0;
    bool have = idx < live;
    if (have && p.belong[idx] >= p.e[idx]) {
        const int x = p.belong[idx];
        (void)x;
        eid = (int)idx;
        const int oe = p.rev[idx];
        double dis1 = 0, dis2 = 0, val;
        for (int i = 0; i < K5; ++i) {
            const int c = p.cid[eid * K5 + i];
            if (c != -1) { val = dis[c] - p.cdis[eid * K5 + i]; if (val > dis1) dis1 = val; }
        }
        for (int i = 0; i < K5; ++i) {
            const int c = p.cid[oe * K5 + i];
            if (c != -1) { val = dis[c] - p.cdis[oe * K5 + i]; if (val > dis2) dis2 = val; }
        }
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
    // first minimum of add over idx (thrust::min_element): key (add, idx); NaN never wins
    double badd = have ? add : __builtin_inf();
    int bidx = have ? (int)idx : 0x7fffffff;
    if (have && !(add == add)) { badd = __builtin_inf(); }
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
    // the winner of the block publishes its tuple
    if (have && (int)idx == sidx[0]) {
        PlacePartial pp; pp.add = add; pp.idx = (int)idx; pp.eid = eid; pp.frac = d1;
        partials[blockIdx.x] = pp;
    }
    if (threadIdx.x == 0 && sidx[0] == 0x7fffffff) {
        PlacePartial pp; pp.add = __builtin_inf(); pp.idx = 0x7fffffff; pp.eid = 0; pp.frac = 0;
        partials[blockIdx.x] = pp;
    }
}

// finish the argmin, updateTreeStructure, updateClosestNodes -- one wave
__global__ __launch_bounds__(64) void place_update_kernel(PlaceBuffers p, const PlacePartial* __restrict__ partials,
                                                          int nparts, int64_t num, int64_t edge_count,
                                                          double* __restrict__ trace)
{
    const int lane = threadIdx.x;
    double badd = __builtin_inf(), bfrac = 0;
    int bidx = 0x7fffffff, beid = 0;
    for (int i = lane; i < nparts; i += 64) {
        const PlacePartial pp = partials[i];
        if (pp.add < badd || (pp.add == badd && pp.idx < bidx)) { badd = pp.add; bidx = pp.idx; beid = pp.eid; bfrac = pp.frac; }
    }
    // slots >= 4*num-4 (and < 4N-4) all carry the tuple (0,0,2): the first of them competes
    const int64_t live = 4 * num - 4, lim = 4 * p.M - 4;
    if (lane == 0 && live < lim) {
        if (2.0 < badd || (2.0 == badd && (int)live < bidx)) { badd = 2.0; bidx = (int)live; beid = 0; bfrac = 0; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oa = __shfl_down(badd, off, 64);
        const int oi = __shfl_down(bidx, off, 64);
        const int oe = __shfl_down(beid, off, 64);
        const double of = __shfl_down(bfrac, off, 64);
        if (oa < badd || (oa == badd && oi < bidx)) { badd = oa; bidx = oi; beid = oe; bfrac = of; }
    }
    const int eid = __shfl(beid, 0, 64);
    const double fracLen = __shfl(bfrac, 0, 64), addLen = __shfl(badd, 0, 64);
    const int placeId = (int)num;
    if (lane == 0) {
        if (trace) { trace[3 * num] = eid; trace[3 * num + 1] = fracLen; trace[3 * num + 2] = addLen; }
        int ec = (int)edge_count;
        const int N = (int)p.N;
        const int middle = placeId + N - 1, outside = placeId;
        const int x = p.belong[eid], y = p.e[eid];
        const double originalDis = p.len[eid];
        const int xe = eid, ye = p.rev[eid];   // the reference finds them by walking head[x] / head[y]
        p.e[xe] = middle; p.len[xe] = fracLen;
        p.e[ye] = middle; p.len[ye] -= fracLen;
        // middle -> x
        p.e[ec] = x; p.len[ec] = fracLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle;
        for (int i = 0; i < K5; ++i)
            if (p.cid[ye * K5 + i] != -1) {
                p.cid[ec * K5 + i] = p.cid[ye * K5 + i];
                p.cdis[ec * K5 + i] = p.cdis[ye * K5 + i] + originalDis - fracLen;
            }
        p.rev[ec] = xe; p.rev[xe] = ec;
        ec++;
        // middle -> y
        p.e[ec] = y; p.len[ec] = originalDis - fracLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle;
        for (int i = 0; i < K5; ++i)
            if (p.cid[xe * K5 + i] != -1) {
                p.cid[ec * K5 + i] = p.cid[xe * K5 + i];
                p.cdis[ec * K5 + i] = p.cdis[xe * K5 + i] + fracLen;
            }
        p.rev[ec] = ye; p.rev[ye] = ec;
        ec++;
        // outside -> middle
        p.e[ec] = middle; p.len[ec] = addLen; p.nxt[ec] = p.head[outside]; p.head[outside] = ec; p.belong[ec] = outside;
        p.rev[ec] = ec + 1;
        ec++;
        // middle -> outside
        p.e[ec] = outside; p.len[ec] = addLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle;
        p.rev[ec] = ec - 1;
        const int e1 = ec - 2, e2 = ec - 3;
        for (int pass = 0; pass < 2; ++pass) {
            const int src = pass == 0 ? e1 : e2;
            for (int i = 0; i < K5; ++i) {
                if (p.cid[src * K5 + i] == -1) break;
                for (int j = 0; j < K5; ++j)
                    if (p.cdis[ec * K5 + j] > p.cdis[src * K5 + i]) {
                        for (int k = K5 - 1; k > j; --k) {
                            p.cdis[ec * K5 + k] = p.cdis[ec * K5 + k - 1];
                            p.cid[ec * K5 + k] = p.cid[ec * K5 + k - 1];
                        }
                        p.cdis[ec * K5 + j] = p.cdis[src * K5 + i];
                        p.cid[ec * K5 + j] = p.cid[src * K5 + i];
                        break;
                    }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    closest_update_wave(p, placeId);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int place_alloc(PlaceBuffers& p, int64_t N, int64_t M)
{
    place_free(p);
    p.N = N;
    p.M = M > 0 ? M : N;
    DPR_HIP(hipMalloc(&p.head, sizeof(int32_t) * (size_t)(2 * N)));
    DPR_HIP(hipMalloc(&p.e, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.nxt, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.belong, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.rev, sizeof(int32_t) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.len, sizeof(double) * (size_t)(8 * N)));
    DPR_HIP(hipMalloc(&p.cid, sizeof(int32_t) * (size_t)(40 * N)));
    DPR_HIP(hipMalloc(&p.cdis, sizeof(double) * (size_t)(40 * N)));
    DPR_HIP(hipMalloc(&p.q_id, sizeof(int32_t) * (size_t)(2 * N + 64)));
    DPR_HIP(hipMalloc(&p.q_from, sizeof(int32_t) * (size_t)(2 * N + 64)));
    DPR_HIP(hipMalloc(&p.q_dis, sizeof(double) * (size_t)(2 * N + 64)));
    p.nparts_max = (int)((4 * N + kThreads - 1) / kThreads + 1);
    DPR_HIP(hipMalloc(&p.partials, sizeof(PlacePartial) * (size_t)p.nparts_max));
    return DPR_OK;
}

void place_free(PlaceBuffers& p)
{
    void* ptrs[] = { p.head, p.e, p.nxt, p.belong, p.rev, p.len, p.cid, p.cdis, p.q_id, p.q_from, p.q_dis, p.partials };
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
    // sequential over the backbone leaves (ties in the lists depend on this order); chunked so that no
    // single launch runs for too long
    const int64_t chunk = 4096;
    for (int64_t t0 = 0; t0 < m; t0 += chunk) {
        const int64_t t1 = t0 + chunk < m ? t0 + chunk : m;
        hipLaunchKernelGGL(place_backbone_lists_kernel, dim3(1), dim3(64), 0, s, p, t0, t1);
    }
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

int place_tip(PlaceBuffers& p, const double* d_dis, int64_t tip, double* d_trace, hipStream_t s)
{
    const int64_t live = 4 * tip - 4;
    const int nblk = (int)((live + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(place_scan_kernel, dim3((unsigned)nblk), dim3(kThreads), 0, s, p, d_dis, tip,
                       reinterpret_cast<PlacePartial*>(p.partials));
    hipLaunchKernelGGL(place_update_kernel, dim3(1), dim3(64), 0, s, p, reinterpret_cast<const PlacePartial*>(p.partials),
                       nblk, tip, live, d_trace);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

}  // namespace dpr
