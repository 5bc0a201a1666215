// Divide-and-conquer mode on gfx950.  Replaces findClustersDC (src/divide_and_conquer/
// placement_close_k.cu:937-1113) and findClusterTreeDC (:1251-1535) with their kernels
// calculateBranchLengthDC (:128-181), calculateBranchLengthSpecialIDDC (:184-240),
// initializeClusterDC (:611-646), updateTreeStructureInClusterDC (:443-527), updateClusterInfoDC
// (:555-575), updateClosestNodesInClusterDC (:313-357).  The backbone tree (findBackboneTreeDC
// :731-935) is the k-closest placement of place.hip with node ids offset by the total tip count.
//
// The reference walks the query tips one by one (one distance launch, one O(backbone) scan, one
// Thrust reduction and a device->host copy per tip) and the clusters one by one (five launches and a
// host round trip per member, sequences staged through host memory in batches).  Here:
//  * everything stays resident in HBM (planes / sketches of ALL tips; 288 GB);
//  * cluster assignment is batched: distances of Q queries to all backbone tips in one launch,
//    written query-minor, then ONE scan where a lane is a query and the backbone edge (its two
//    closest lists) is wave-uniform; a workgroup stages the distance rows its chunk of edges names
//    in LDS once (neighbours in the tree share most of their closest leaves);
//  * clusters are independent (disjoint edge slots, node ids and slots known from a prefix sum over
//    the cluster sizes), so all cluster trees are built concurrently, one wavefront per cluster, from
//    per-cluster distance blocks computed beforehand by the tiled pair kernels.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "dpr_internal.hpp"

namespace dpr {

constexpr int K5 = 5;
constexpr int kAE = 64;       // backbone edges per scan chunk at most (their records sit in LDS: 176 B each)
constexpr int kDcRows = 48;   // distinct closest leaves per chunk = distance rows staged in LDS (64 queries x 8 B each) + one row of -inf

// ------------------------------------------------------------------------------------------------
// cluster assignment
// ------------------------------------------------------------------------------------------------
// gather the eligible backbone slots (belong >= e, ascending) into a dense table: 2 x 5 closest ids
// and path lengths (own list, reverse list) and the edge length
__global__ __launch_bounds__(kThreads) void dc_edge_table_kernel(PlaceBuffers p, const int32_t* __restrict__ vslots, int nv,
                                                                 int32_t* __restrict__ et_cid, double* __restrict__ et_cdis,
                                                                 double* __restrict__ et_len)
{
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= nv) return;
    const int s = vslots[idx], o = p.rev[s];
    for (int i = 0; i < K5; ++i) {
        et_cid[idx * 10 + i] = p.cid[s * K5 + i];
        et_cdis[idx * 10 + i] = p.cdis[s * K5 + i];
        et_cid[idx * 10 + 5 + i] = p.cid[o * K5 + i];
        et_cdis[idx * 10 + 5 + i] = p.cdis[o * K5 + i];
    }
    et_len[idx] = p.len[s];
}

// calculateBranchLengthDC for (chunk of table entries blockIdx.x) x (64 queries blockIdx.y); lane = query.
// dT[c * ldq + q] = distance(query q, backbone tip c).  Writes the chunk's first minimum per query.
constexpr int kDcRec = 11;    // 16-byte words per packed table entry
__global__ __launch_bounds__(kThreads) void dc_pack_records_kernel(const int32_t* __restrict__ off, const double* __restrict__ et_cdis,
                                                                   const double* __restrict__ et_len, const int32_t* __restrict__ vslots,
                                                                   int nv, uint4* __restrict__ rec)
{
    const int e = blockIdx.x * kThreads + threadIdx.x;
    if (e >= nv) return;
    uint4* r = rec + (int64_t)e * kDcRec;
    for (int i = 0; i < 10; ++i) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(et_cdis[e * 10 + i]);
        r[i] = make_uint4((uint32_t)off[e * 10 + i], 0u, (uint32_t)b, (uint32_t)(b >> 32));
    }
    const unsigned long long lb = (unsigned long long)__double_as_longlong(et_len[e]);
    r[10] = make_uint4((uint32_t)vslots[e], 0u, (uint32_t)lb, (uint32_t)(lb >> 32));
}

// One workgroup = 64 queries x one chunk of table entries.  Round 5: the chunk's distinct closest leaves (<= kDcRows; neighbours in
// the tree share most of theirs: 47 rows for 56 entries' 560 references on average) are staged in LDS once -- a lane per query,
// the four wavefronts a quarter of the rows each, all loads in flight together -- and the 10 look-ups per entry read LDS at a
// wave-uniform row; the former scan read every reference from L2 / Infinity Cache (7.6 TB per 950 000 queries x 100 000 edges,
// 3.5 ms per launch).  The four wavefronts then take every fourth entry of the chunk for the same 64 queries and combine their
// minima.  The entries' scalars (row offset and path length per list entry, edge length, slot) are copied to LDS with the rows and
// read there at a wave-uniform address (a broadcast): as scalar loads they shared a counter with the LDS reads and made every
// entry a chain of five dependent round trips (2.3 ms); as one record per entry read out with v_readlane they were a third of
// the loop's vector instructions (1.4 ms).  An absent list entry points at a row of -inf: its candidate never exceeds the running
// maximum, as the reference's `!= -1` test.  Same arithmetic per (query, edge); the minimum does not depend on the order (ties
// by slot).
__global__ __launch_bounds__(256) void dc_assign_scan_kernel(const int32_t* __restrict__ ch_e0, const int32_t* __restrict__ ch_l0,
                                                             const int32_t* __restrict__ ch_leaf, const uint4* __restrict__ et_rec,
                                                             const double* __restrict__ dT, int64_t ldq, int Q,
                                                             double* __restrict__ part_add, int32_t* __restrict__ part_pos)
{
    __shared__ double rows[(kDcRows + 1) * 64];
    __shared__ uint4 meta[kAE * kDcRec];
    __shared__ double s_best[3][64];
    __shared__ int s_pos[3][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch = blockIdx.x;
    const int q = blockIdx.y * 64 + lane;
    const int qq = q < Q ? q : Q - 1;
    const int e0 = ch_e0[ch], e1 = ch_e0[ch + 1], l0 = ch_l0[ch], nrow = ch_l0[ch + 1] - l0;
    const double* col = dT + qq;
    {
        const int nm = (e1 - e0) * kDcRec;
        const uint4* __restrict__ src = et_rec + (int64_t)e0 * kDcRec;
        uint4 m[(kAE * kDcRec + 255) / 256];
#pragma unroll
        for (int k = 0; k < (kAE * kDcRec + 255) / 256; ++k) {
            const int idx = (int)threadIdx.x + 256 * k;
            m[k] = idx < nm ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
        }
        double v[kDcRows / 4];
#pragma unroll
        for (int k = 0; k < kDcRows / 4; ++k) {
            const int r = w + 4 * k;
            v[k] = r < nrow ? col[(int64_t)ch_leaf[l0 + r] * ldq] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < (kAE * kDcRec + 255) / 256; ++k) {
            const int idx = (int)threadIdx.x + 256 * k;
            if (idx < nm) meta[idx] = m[k];
        }
#pragma unroll
        for (int k = 0; k < kDcRows / 4; ++k) {
            const int r = w + 4 * k;
            if (r < nrow) rows[r * 64 + lane] = v[k];
        }
        if (w == 0) rows[kDcRows * 64 + lane] = -__builtin_inf();
    }
    __syncthreads();
    double best = __builtin_inf();
    int bpos = 0x7fffffff;      // the SLOT of the best edge: ties go to the lowest slot
    auto f64_of = [](const uint4& u) -> double { return __longlong_as_double((long long)(((unsigned long long)u.w << 32) | u.z)); };
    const int ne = e1 - e0;
#pragma unroll 2
    for (int el = w; el < ne; el += 4) {
        const uint4* __restrict__ mr = meta + el * kDcRec;
        // (fmax for the reference's `if (val > dis) dis = val`: dis starts at +0 and only ever takes a larger value, a NaN candidate is
        //  passed over by both forms)
        double dis1 = 0, dis2 = 0;
#pragma unroll
        for (int i = 0; i < K5; ++i) { const uint4 u = mr[i]; dis1 = fmax(dis1, rows[u.x + lane] - f64_of(u)); }
#pragma unroll
        for (int i = 0; i < K5; ++i) { const uint4 u = mr[5 + i]; dis2 = fmax(dis2, rows[u.x + lane] - f64_of(u)); }
        const uint4 t = mr[10];
        const double L = f64_of(t);
        double a = (dis1 + dis2 - L) / 2;
        if (a < 0) a = 0;
        dis1 -= a; dis2 -= a;
        if (dis1 < 0) dis1 = 0;
        if (dis2 < 0) dis2 = 0;
        if (dis1 > L) { a += dis1 - L; dis1 = L; }
        if (dis2 > L) { a += dis2 - L; dis2 = L; }
        const int slot = (int)t.x;
        if (a < best || (a == best && slot < bpos)) { best = a; bpos = slot; }
    }
    if (w > 0) { s_best[w - 1][lane] = best; s_pos[w - 1][lane] = bpos; }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a = s_best[k][lane];
            const int sl = s_pos[k][lane];
            if (a < best || (a == best && sl < bpos)) { best = a; bpos = sl; }
        }
        if (q < Q) {
            part_add[(int64_t)blockIdx.x * ldq + q] = best;
            part_pos[(int64_t)blockIdx.x * ldq + q] = bpos;
        }
    }
}

// thrust::min_element over all 4B-4 tuples: ineligible slots carry (0,0,2) and slot 0 is always one
// of them (belong 0 < e), so the winner is the first eligible minimum if it is < 2, else tuple eid 0.
// (64 queries per workgroup, its 16 wavefronts every 16th chunk: a thread per query walking all chunks was 1 200 dependent steps)
constexpr int kRedWaves = 16;
__global__ __launch_bounds__(64 * kRedWaves) void dc_assign_reduce_kernel(const double* __restrict__ part_add,
                                                                         const int32_t* __restrict__ part_pos, int nchunks,
                                                                         int64_t ldq, int Q, int32_t* __restrict__ cluster_id)
{
    __shared__ double s_best[kRedWaves][64];
    __shared__ int s_pos[kRedWaves][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const int qq = q < Q ? q : Q - 1;
    double best = __builtin_inf();
    int bpos = 0x7fffffff;
#pragma unroll 4
    for (int c = w; c < nchunks; c += kRedWaves) {
        const double a = part_add[(int64_t)c * ldq + qq];
        const int sl = part_pos[(int64_t)c * ldq + qq];
        if (a < best || (a == best && sl < bpos)) { best = a; bpos = sl; }
    }
    s_best[w][lane] = best; s_pos[w][lane] = bpos;
    __syncthreads();
    if (w == 0 && q < Q) {
#pragma unroll
        for (int k = 1; k < kRedWaves; ++k) {
            const double a = s_best[k][lane];
            const int sl = s_pos[k][lane];
            if (a < best || (a == best && sl < bpos)) { best = a; bpos = sl; }
        }
        cluster_id[q] = (best < 2.0) ? bpos : 0;
    }
}

int dc_table_build(PlaceBuffers& p, int64_t B, DcTable& t, hipStream_t s)
{
    dc_table_free(t);
    const auto tb0 = std::chrono::steady_clock::now();
    const int64_t lim = 4 * B - 4;
    std::vector<int32_t> hb((size_t)lim), he((size_t)lim), hn((size_t)lim), hh((size_t)(2 * p.N));
    DPR_HIP(hipMemcpyAsync(hb.data(), p.belong, sizeof(int32_t) * (size_t)lim, hipMemcpyDeviceToHost, s));
    DPR_HIP(hipMemcpyAsync(he.data(), p.e, sizeof(int32_t) * (size_t)lim, hipMemcpyDeviceToHost, s));
    DPR_HIP(hipMemcpyAsync(hn.data(), p.nxt, sizeof(int32_t) * (size_t)lim, hipMemcpyDeviceToHost, s));
    DPR_HIP(hipMemcpyAsync(hh.data(), p.head, sizeof(int32_t) * (size_t)(2 * p.N), hipMemcpyDeviceToHost, s));
    DPR_HIP(hipStreamSynchronize(s));
    // eligible slots (belong >= e) in TREE order (depth-first from node N): edges that are neighbours in
    // the backbone share most of their closest leaves, so a scan block's 256 edges touch a few hundred
    // distance rows instead of ~2500 and they stay in L2.  The minimum is order-independent (ties by slot).
    std::vector<int32_t> vs;
    vs.reserve((size_t)lim / 2 + 1);
    {
        std::vector<std::pair<int32_t, int32_t>> st;   // (node, slot we came through or -1)
        st.emplace_back((int32_t)p.N, -1);
        while (!st.empty()) {
            const auto [node, via] = st.back();
            st.pop_back();
            for (int32_t i = hh[(size_t)node]; i != -1; i = hn[(size_t)i]) {
                if (via >= 0 && he[(size_t)i] == hb[(size_t)via]) continue;      // the edge back to where we came from
                // undirected edge {node, e[i]}: its eligible direction is slot i or its reverse
                const int32_t to = he[(size_t)i];
                int32_t r = hh[(size_t)to];
                while (r != -1 && he[(size_t)r] != node) r = hn[(size_t)r];
                vs.push_back(hb[(size_t)i] >= he[(size_t)i] ? i : r);
                st.emplace_back(to, i);
            }
        }
    }
    if ((int64_t)vs.size() != lim / 2) { set_error("divide-and-conquer: backbone is not a tree over its slots"); return DPR_ERR_STATE; }
    t.nv = (int)vs.size();
    if (t.nv == 0) { set_error("divide-and-conquer: backbone has no eligible edge"); return DPR_ERR_STATE; }
    DPR_HIP(hipMalloc(&t.vslots, sizeof(int32_t) * vs.size()));
    DPR_HIP(hipMalloc(&t.et_cid, sizeof(int32_t) * vs.size() * 10));
    DPR_HIP(hipMalloc(&t.et_cdis, sizeof(double) * vs.size() * 10));
    DPR_HIP(hipMalloc(&t.et_len, sizeof(double) * vs.size()));
    DPR_HIP(hipMemcpyAsync(t.vslots, vs.data(), sizeof(int32_t) * vs.size(), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(dc_edge_table_kernel, dim3((unsigned)((t.nv + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p,
                       t.vslots, t.nv, t.et_cid, t.et_cdis, t.et_len);
    DPR_HIP(hipGetLastError());
    DPR_HIP(hipStreamSynchronize(s));   // vs goes out of scope
    // chunks for the assignment scan: consecutive entries while their lists name at most kDcRows distinct tips (and kAE entries)
    {
        std::vector<int32_t> hc(vs.size() * 10), off(vs.size() * 10), e0v{ 0 }, l0v{ 0 }, leaf, cur;
        DPR_HIP(hipMemcpy(hc.data(), t.et_cid, sizeof(int32_t) * hc.size(), hipMemcpyDeviceToHost));
        std::vector<int32_t> loc((size_t)p.N, -1);          // tip -> row of the current chunk
        leaf.reserve(vs.size());
        int32_t fresh[10];
        for (size_t e = 0; e < vs.size(); ++e) {
            auto count_fresh = [&]() {
                int nf = 0;
                for (int i = 0; i < 10; ++i) {
                    const int32_t id = hc[e * 10 + (size_t)i];
                    if (id < 0 || loc[(size_t)id] >= 0) continue;
                    bool seen = false;
                    for (int k = 0; k < nf; ++k) seen = seen || fresh[k] == id;
                    if (!seen) fresh[nf++] = id;
                }
                return nf;
            };
            int nf = count_fresh();
            if ((int)cur.size() + nf > kDcRows || (int64_t)e - (int64_t)e0v.back() >= kAE) {      // close the chunk
                for (int32_t id : cur) loc[(size_t)id] = -1;
                cur.clear();
                e0v.push_back((int32_t)e);
                l0v.push_back((int32_t)leaf.size());
                nf = count_fresh();
            }
            for (int k = 0; k < nf; ++k) { loc[(size_t)fresh[k]] = (int32_t)cur.size(); cur.push_back(fresh[k]); leaf.push_back(fresh[k]); }
            for (int i = 0; i < 10; ++i) {
                const int32_t id = hc[e * 10 + (size_t)i];
                off[e * 10 + (size_t)i] = (id < 0 ? kDcRows : loc[(size_t)id]) * 64;
            }
        }
        e0v.push_back((int32_t)vs.size());
        l0v.push_back((int32_t)leaf.size());
        t.nch = (int)e0v.size() - 1;
        if (leaf.empty()) leaf.push_back(0);
        DPR_HIP(hipMalloc(&t.ch_e0, sizeof(int32_t) * e0v.size()));
        DPR_HIP(hipMalloc(&t.ch_l0, sizeof(int32_t) * l0v.size()));
        DPR_HIP(hipMalloc(&t.ch_leaf, sizeof(int32_t) * leaf.size()));
        int32_t* d_off = nullptr;
        DPR_HIP(hipMalloc(&d_off, sizeof(int32_t) * off.size()));
        DPR_HIP(hipMalloc(&t.et_rec, sizeof(uint4) * vs.size() * kDcRec));
        DPR_HIP(hipMemcpy(t.ch_e0, e0v.data(), sizeof(int32_t) * e0v.size(), hipMemcpyHostToDevice));
        DPR_HIP(hipMemcpy(t.ch_l0, l0v.data(), sizeof(int32_t) * l0v.size(), hipMemcpyHostToDevice));
        DPR_HIP(hipMemcpy(t.ch_leaf, leaf.data(), sizeof(int32_t) * leaf.size(), hipMemcpyHostToDevice));
        DPR_HIP(hipMemcpy(d_off, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(dc_pack_records_kernel, dim3((unsigned)((t.nv + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, d_off, t.et_cdis, t.et_len,
                           t.vslots, t.nv, t.et_rec);
        DPR_HIP(hipGetLastError());
        DPR_HIP(hipStreamSynchronize(s));
        (void)hipFree(d_off);
        if (log_level("dc") > 0)
            std::fprintf(stderr, "[dc] assignment table: %d entries in %d chunks (%.1f entries, %.1f distinct closest leaves per chunk), built in %.1f ms\n", t.nv, t.nch,
                         (double)t.nv / (double)t.nch, (double)leaf.size() / (double)t.nch,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
    }
    return DPR_OK;
}

void dc_table_free(DcTable& t)
{
    void* ptrs[] = { t.vslots, t.et_cid, t.et_cdis, t.et_len, t.ch_e0, t.ch_l0, t.ch_leaf, t.et_rec, t.part_add, t.part_pos };
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    t = DcTable();
}

int dc_assign(DcTable& t, const double* dT, int64_t ldq, int Q, int32_t* d_cluster_id, hipStream_t s)
{
    const int nchunks = t.nch;
    const size_t need = (size_t)nchunks * (size_t)ldq;
    if (need > t.part_cap) {
        if (t.part_add) (void)hipFree(t.part_add);
        if (t.part_pos) (void)hipFree(t.part_pos);
        t.part_add = nullptr; t.part_pos = nullptr;
        DPR_HIP(hipMalloc(&t.part_add, sizeof(double) * need));
        DPR_HIP(hipMalloc(&t.part_pos, sizeof(int32_t) * need));
        t.part_cap = need;
    }
    // chunks are the fast grid index: the blocks in flight share few query groups, whose distance columns then stay in L2 /
    // Infinity Cache while all chunks sweep them
    dim3 grid((unsigned)nchunks, (unsigned)((Q + 63) / 64));
    hipLaunchKernelGGL(dc_assign_scan_kernel, grid, dim3(256), 0, s, t.ch_e0, t.ch_l0, t.ch_leaf, t.et_rec, dT, ldq, Q,
                       t.part_add, t.part_pos);
    hipLaunchKernelGGL(dc_assign_reduce_kernel, dim3((unsigned)((Q + 63) / 64)), dim3(64 * kRedWaves), 0, s,
                       t.part_add, t.part_pos, nchunks, ldq, Q, d_cluster_id);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// ------------------------------------------------------------------------------------------------
// cluster trees
// ------------------------------------------------------------------------------------------------
struct DcCluster {
    int32_t slot;        // cluster edge j (eligible backbone slot)
    int32_t m;           // members
    int32_t base_slot;   // first new slot: 4B-4 + 4 * (members of clusters with a smaller slot)
    int32_t base_leaf;   // insertLeafCount before its first member: B + the same count
    int64_t moff;        // members[moff + t]
    int64_t coff;        // cols[coff + u]
    int64_t out;         // distance block offset (doubles)
    int64_t qoff;        // BFS queue scratch offset
    int32_t ld;          // row stride of the distance block
    int32_t pad;
};

// leaf list of a cluster (initializeClusterDC): closest ids of edge j, of its reverse, then the members
__global__ __launch_bounds__(64) void dc_cols_kernel(PlaceBuffers p, const DcCluster* __restrict__ cl,
                                                     const int32_t* __restrict__ members, int32_t* __restrict__ cols,
                                                     int32_t* __restrict__ clx)
{
    const DcCluster C = cl[blockIdx.x];
    const int lane = threadIdx.x;
    const int j = C.slot, oth = p.rev[j];
    if (lane < K5) { cols[C.coff + lane] = p.cid[j * K5 + lane]; clx[j * K5 + lane] = lane; }
    else if (lane < 2 * K5) { cols[C.coff + lane] = p.cid[oth * K5 + lane - K5]; clx[oth * K5 + lane - K5] = lane; }
    for (int t = lane; t < C.m; t += 64) cols[C.coff + kDcLeaves + t] = members[C.moff + t];
}

__device__ __forceinline__ bool dc_list_insert(double* cdis, int32_t* cid, int32_t* clx, int slot, int x, int xl, double d)
{
    for (int j = 0; j < K5; ++j) {
        const double nowd = cdis[slot * K5 + j];
        if (nowd > d) {
            for (int k = K5 - 1; k > j; --k) {
                cdis[slot * K5 + k] = cdis[slot * K5 + k - 1];
                cid[slot * K5 + k] = cid[slot * K5 + k - 1];
                clx[slot * K5 + k] = clx[slot * K5 + k - 1];
            }
            cdis[slot * K5 + j] = d;
            cid[slot * K5 + j] = x;
            clx[slot * K5 + j] = xl;
            return true;
        }
    }
    return false;
}

// One wavefront per cluster: members in ascending tip order; per member the masked edge scan
// (positions in the reference's edgeMask order), the edge split and the in-cluster closest update.
// clx mirrors cid with the column of the leaf inside the cluster's distance block.
// kW wavefronts per cluster: the masked edge scan of a member (2 + 4t positions, the part that grows with the cluster)
// is shared by all of them, the split and the in-cluster closest update stay with wavefront 0.  kW = 1 for the many
// small clusters (one wavefront each, thousands resident), kW = 16 for the few large ones, whose serial member loops
// would otherwise bound the phase (cost ~ m^2 / lanes).
template <int kW>
__global__ __launch_bounds__(64 * kW) void dc_cluster_kernel(PlaceBuffers p, int32_t* __restrict__ clx,
                                                             const DcCluster* __restrict__ cl,
                                                             const int32_t* __restrict__ members,
                                                             const double* __restrict__ Dc, int32_t* __restrict__ qid,
                                                             int32_t* __restrict__ qfrom, double* __restrict__ qdis,
                                                             int32_t* __restrict__ status, double* __restrict__ trace)
{
    __shared__ double s_add[kW], s_frac[kW];
    __shared__ int s_pos[kW], s_eid[kW];
    const DcCluster C = cl[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = C.slot, oth = p.rev[j];
    const int N = (int)p.N;
    int32_t* q_id = qid + C.qoff; int32_t* q_from = qfrom + C.qoff; double* q_dis = qdis + C.qoff;
    for (int t = 0; t < C.m; ++t) {
        const int leaf = members[C.moff + t];
        const double* row = Dc + C.out + (int64_t)t * C.ld;
        const int edge_count = 2 + 4 * t;
        // ---- calculateBranchLengthSpecialIDDC + first minimum over mask positions
        double badd = __builtin_inf(), bfrac = 0;
        int bpos = 0x7fffffff, beid = 0;
        for (int pos = tid; pos < edge_count; pos += 64 * kW) {
            const int slot = pos == 0 ? j : pos == 1 ? oth : C.base_slot + 4 * ((pos - 2) >> 2) + (3 - ((pos - 2) & 3));
            double add = 2.0, d1 = 0.0;
            int eid = 0;
            if (p.belong[slot] >= p.e[slot]) {
                eid = slot;
                const int oe = p.rev[slot];
                double dis1 = 0, dis2 = 0, val;
                for (int i = 0; i < K5; ++i)
                    if (p.cid[eid * K5 + i] != -1) { val = row[clx[eid * K5 + i]] - p.cdis[eid * K5 + i]; if (val > dis1) dis1 = val; }
                for (int i = 0; i < K5; ++i)
                    if (p.cid[oe * K5 + i] != -1) { val = row[clx[oe * K5 + i]] - p.cdis[oe * K5 + i]; if (val > dis2) dis2 = val; }
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
            if (add < badd) { badd = add; bpos = pos; beid = eid; bfrac = d1; }   // positions ascend per lane
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double oa = __shfl_down(badd, off, 64);
            const int op = __shfl_down(bpos, off, 64);
            const int oe = __shfl_down(beid, off, 64);
            const double of = __shfl_down(bfrac, off, 64);
            if (oa < badd || (oa == badd && op < bpos)) { badd = oa; bpos = op; beid = oe; bfrac = of; }
        }
        if (kW > 1) {     // wave winners -> block winner (smallest add, then smallest position), the same in every thread
            if (lane == 0) { s_add[wave] = badd; s_pos[wave] = bpos; s_eid[wave] = beid; s_frac[wave] = bfrac; }
            __syncthreads();
            badd = s_add[0]; bpos = s_pos[0]; beid = s_eid[0]; bfrac = s_frac[0];
#pragma unroll
            for (int w = 1; w < kW; ++w)
                if (s_add[w] < badd || (s_add[w] == badd && s_pos[w] < bpos)) { badd = s_add[w]; bpos = s_pos[w]; beid = s_eid[w]; bfrac = s_frac[w]; }
        }
        const int eid = kW > 1 ? beid : __shfl(beid, 0, 64);
        const int wpos = kW > 1 ? bpos : __shfl(bpos, 0, 64);
        const double fracLen = kW > 1 ? bfrac : __shfl(bfrac, 0, 64), addLen = kW > 1 ? badd : __shfl(badd, 0, 64);
        // an ineligible tuple (eid 0, add 2) or nothing comparable won: the reference would split slot 0,
        // which belongs to another cluster; reported instead (needs distances >= 2)
        if (wpos == 0x7fffffff || !(addLen < 2.0)) {     // block-uniform
            if (tid == 0) atomicExch(status, 1 + blockIdx.x);
            return;
        }
        const int ec0 = C.base_slot + 4 * t;
        if (tid == 0) {
            if (trace) { trace[3 * leaf + 1] = fracLen; trace[3 * leaf + 2] = addLen; }   // [3*leaf] keeps the cluster id
            int ec = ec0;
            const int middle = C.base_leaf + t + N - 1, outside = leaf;
            const int x = p.belong[eid], y = p.e[eid];
            const double originalDis = p.len[eid];
            const int xe = eid, ye = p.rev[eid];
            p.e[xe] = middle; p.len[xe] = fracLen;
            p.e[ye] = middle; p.len[ye] -= fracLen;
            // middle -> x
            p.e[ec] = x; p.len[ec] = fracLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle;
            for (int i = 0; i < K5; ++i)
                if (p.cid[ye * K5 + i] != -1) {
                    p.cid[ec * K5 + i] = p.cid[ye * K5 + i];
                    clx[ec * K5 + i] = clx[ye * K5 + i];
                    p.cdis[ec * K5 + i] = p.cdis[ye * K5 + i] + originalDis - fracLen;
                }
            p.rev[ec] = xe; p.rev[xe] = ec;
            ec++;
            // middle -> y
            p.e[ec] = y; p.len[ec] = originalDis - fracLen; p.nxt[ec] = p.head[middle]; p.head[middle] = ec; p.belong[ec] = middle;
            for (int i = 0; i < K5; ++i)
                if (p.cid[xe * K5 + i] != -1) {
                    p.cid[ec * K5 + i] = p.cid[xe * K5 + i];
                    clx[ec * K5 + i] = clx[xe * K5 + i];
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
                    dc_list_insert(p.cdis, p.cid, clx, ec, p.cid[src * K5 + i], clx[src * K5 + i], p.cdis[src * K5 + i]);
                }
            }
            q_id[0] = leaf; q_dis[0] = 0.0; q_from[0] = -1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (kW == 1) __builtin_amdgcn_s_barrier();      // (kW > 1: wavefront 0 goes on alone, program order + the fences)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // ---- updateClosestNodesInClusterDC, frontier-parallel (every directed edge of a tree is reached at
        // most once, so the insertions do not depend on the visiting order)
        const int ed1 = p.e[j], ed2 = p.belong[j];
        const int hi_slot = ec0 + 4;   // cluster slots: j, oth, [base_slot, hi_slot)
        int l = 0, r = 1;
        while (wave == 0 && l < r) {
            const int cnt = min(64, r - l);
            int node = -1, fb = -1;
            double d = 0.0;
            if (lane < cnt) { node = q_id[l + lane]; fb = q_from[l + lane]; d = q_dis[l + lane]; }
            const bool expand = lane < cnt && node != ed1 && node != ed2;
            int nnew = 0;
            unsigned long long took = 0ull;
            if (expand) {
                int pos = 0;
                for (int i = p.head[node]; i != -1; i = p.nxt[i], ++pos) {
                    if (!(i == j || i == oth || (i >= C.base_slot && i < hi_slot))) continue;   // edge mask first
                    if (p.e[i] == fb) continue;
                    if (dc_list_insert(p.cdis, p.cid, clx, i, leaf, kDcLeaves + t, d) && pos < 64) { took |= 1ull << pos; nnew++; }
                }
            }
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
                    if (pos < 64 && ((took >> pos) & 1ull)) { q_id[w] = p.e[i]; q_dis[w] = d + p.len[i]; q_from[w] = node; ++w; }
            }
            l += cnt;
            r += total;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (kW == 1) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (kW > 1) {     // the other wavefronts scan the next member against what wavefront 0 has just stored
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
}

void dc_deal_clusters(const int64_t* sizes_desc, int64_t count, int world, int32_t* owner)
{
    std::vector<double> load((size_t)world, 0.0);
    for (int64_t i = 0; i < count; ++i) {
        int best = 0;
        for (int r = 1; r < world; ++r)
            if (load[(size_t)r] < load[(size_t)best]) best = r;
        load[(size_t)best] += (double)sizes_desc[i] * (double)(sizes_desc[i] + 2 * kDcLeaves);
        owner[i] = best;
    }
}

void dc_query_share(int64_t n, int64_t B, int rank, int world, int64_t* q0, int64_t* q1)
{
    const int64_t nq = n - B;
    const int64_t share = ((nq + world - 1) / world + 255) / 256 * 256;
    int64_t a = B + (int64_t)rank * share, b = a + share;
    if (a > n) a = n;
    if (b > n) b = n;
    *q0 = a; *q1 = b;
}

__global__ __launch_bounds__(kThreads) void dc_delta_kernel(unsigned long long* __restrict__ cur,
                                                            const unsigned long long* __restrict__ old, int64_t n, int add)
{
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        cur[i] = add ? cur[i] + old[i] : cur[i] - old[i];
}
// 32-bit arrays are processed as pairs in one 64-bit word: a borrow/carry between the halves cancels in
// old + ((new - old) summed over ranks), because at most one rank contributes a non-zero difference
int dc_delta_sub(void* cur, const void* old, int64_t words64, hipStream_t s)
{
    if (words64 <= 0) return DPR_OK;
    hipLaunchKernelGGL(dc_delta_kernel, dim3(2048), dim3(kThreads), 0, s, (unsigned long long*)cur, (const unsigned long long*)old, words64, 0);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}
int dc_delta_add(void* cur, const void* old, int64_t words64, hipStream_t s)
{
    if (words64 <= 0) return DPR_OK;
    hipLaunchKernelGGL(dc_delta_kernel, dim3(2048), dim3(kThreads), 0, s, (unsigned long long*)cur, (const unsigned long long*)old, words64, 1);
    DPR_HIP(hipGetLastError());
    return DPR_OK;
}

// host: group the query tips by cluster, lay out the per-cluster blocks, run cols -> pair jobs -> trees
int dc_cluster_phase(PlaceBuffers& p, const int32_t* h_cluster_id, int64_t N, int64_t B, int source, int dist_type,
                     const MsaBuffers* msa, const MashBuffers* mash, double* d_trace, size_t budget_bytes,
                     DcStats* stats, int rank, int world, hipStream_t s)
{
    const int64_t lim = 4 * B - 4, nq = N - B;
    if (nq <= 0) return DPR_OK;
    std::vector<int64_t> cnt((size_t)lim, 0);
    for (int64_t t = B; t < N; ++t) {
        const int32_t c = h_cluster_id[t];
        if (c < 0 || c >= lim) { set_error("divide-and-conquer: cluster id out of range"); return DPR_ERR_STATE; }
        cnt[(size_t)c]++;
    }
    // members grouped by ascending slot (ascending tip inside), as the reference's contains[] vectors
    std::vector<int64_t> start((size_t)lim + 1, 0);
    for (int64_t c = 0; c < lim; ++c) start[(size_t)c + 1] = start[(size_t)c] + cnt[(size_t)c];
    std::vector<int32_t> members((size_t)nq);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t t = B; t < N; ++t) members[(size_t)fill[(size_t)h_cluster_id[t]]++] = (int32_t)t;
    }
    std::vector<DcCluster> cl;
    int64_t maxm = 0;
    for (int64_t c = 0; c < lim; ++c) {
        const int64_t m = cnt[(size_t)c];
        if (m == 0) continue;
        if (m >= B) {   // src/divide_and_conquer/placement_close_k.cu:1339-1346: exit(1) above B, endless loop at B
            set_error("divide-and-conquer: cluster " + std::to_string(c) + " has " + std::to_string(m) +
                      " members, not fewer than the backbone size " + std::to_string(B));
            return DPR_ERR_STATE;
        }
        DcCluster C{};
        C.slot = (int32_t)c; C.m = (int32_t)m;
        C.base_slot = (int32_t)(lim + 4 * start[(size_t)c]);
        C.base_leaf = (int32_t)(B + start[(size_t)c]);
        C.moff = start[(size_t)c];
        C.ld = (int32_t)((kDcLeaves + m + 1) & ~1);
        cl.push_back(C);
        maxm = std::max(maxm, m);
    }
    // largest clusters first: their serial member loops bound the phase
    std::stable_sort(cl.begin(), cl.end(), [](const DcCluster& a, const DcCluster& b) { return a.m > b.m; });
    if (stats) { stats->clusters = (int64_t)cl.size(); stats->max_cluster = maxm; stats->pairs = 0; stats->groups = 0; stats->jobs = 0; }
    if (world > 1) {
        // clusters are independent: deal them to the ranks, largest first onto the least loaded rank
        // (cost ~ members^2; identical on every rank), and keep this rank's share.  Slots and node ids of a
        // cluster do not depend on who builds it.
        std::vector<int64_t> sizes(cl.size());
        std::vector<int32_t> owner(cl.size());
        for (size_t i = 0; i < cl.size(); ++i) sizes[i] = cl[i].m;
        dc_deal_clusters(sizes.data(), (int64_t)sizes.size(), world, owner.data());
        std::vector<DcCluster> mine;
        for (size_t i = 0; i < cl.size(); ++i)
            if (owner[i] == rank) mine.push_back(cl[i]);
        cl.swap(mine);
    }
    const int64_t ncl = (int64_t)cl.size();
    int64_t coff = 0, qoff = 0;
    for (auto& C : cl) {
        C.coff = coff; coff += kDcLeaves + C.m;
        C.qoff = qoff; qoff += 2 * (int64_t)C.m + 8 + 64;
    }
    if (ncl == 0) return DPR_OK;

    int32_t *d_members = nullptr, *d_cols = nullptr, *d_clx = nullptr, *d_qid = nullptr, *d_qfrom = nullptr, *d_status = nullptr;
    double* d_qdis = nullptr;
    DcCluster* d_cl = nullptr;
    DPR_HIP(hipMalloc(&d_members, sizeof(int32_t) * (size_t)nq));
    DPR_HIP(hipMalloc(&d_cols, sizeof(int32_t) * (size_t)coff));
    DPR_HIP(hipMalloc(&d_clx, sizeof(int32_t) * (size_t)(40 * N)));
    DPR_HIP(hipMalloc(&d_qid, sizeof(int32_t) * (size_t)qoff));
    DPR_HIP(hipMalloc(&d_qfrom, sizeof(int32_t) * (size_t)qoff));
    DPR_HIP(hipMalloc(&d_qdis, sizeof(double) * (size_t)qoff));
    DPR_HIP(hipMalloc(&d_status, sizeof(int32_t)));
    DPR_HIP(hipMalloc(&d_cl, sizeof(DcCluster) * (size_t)ncl));
    DPR_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    DPR_HIP(hipMemcpyAsync(d_members, members.data(), sizeof(int32_t) * (size_t)nq, hipMemcpyHostToDevice, s));

    const int tr_rows = source == DPR_SRC_MSA ? msa_dist_tile_edge(dist_type) : mash_jobs_rows();
    const int tr_cols = source == DPR_SRC_MSA ? msa_dist_tile_edge(dist_type) : mash_jobs_cols();
    std::vector<int64_t> h_moff, h_coff, h_out;
    std::vector<int32_t> h_m, h_ld;
    std::vector<int4> jobs;
    int rc = DPR_OK;
    int64_t g0 = 0;
    int32_t* d_i32 = nullptr; int64_t* d_i64 = nullptr; int4* d_jobs = nullptr; double* d_out = nullptr;
    auto cleanup_group = [&]() {
        if (d_i32) (void)hipFree(d_i32);
        if (d_i64) (void)hipFree(d_i64);
        if (d_jobs) (void)hipFree(d_jobs);
        if (d_out) (void)hipFree(d_out);
        d_i32 = nullptr; d_i64 = nullptr; d_jobs = nullptr; d_out = nullptr;
    };
    const int64_t big_m = 64;   // clusters above this size get a whole workgroup
    auto run = [&]() -> int {
        while (g0 < ncl) {
            // ---- a group of clusters whose distance blocks fit the budget
            int64_t g1 = g0, outsz = 0;
            while (g1 < ncl) {
                const int64_t add = (int64_t)cl[(size_t)g1].m * cl[(size_t)g1].ld;
                if (g1 > g0 && (size_t)(outsz + add) * sizeof(double) > budget_bytes) break;
                cl[(size_t)g1].out = outsz; outsz += add; ++g1;
            }
            const int64_t gn = g1 - g0;
            h_moff.assign((size_t)gn, 0); h_coff.assign((size_t)gn, 0); h_out.assign((size_t)gn, 0);
            h_m.assign((size_t)gn, 0); h_ld.assign((size_t)gn, 0);
            jobs.clear();
            for (int64_t i = 0; i < gn; ++i) {
                const DcCluster& C = cl[(size_t)(g0 + i)];
                h_moff[(size_t)i] = C.moff; h_coff[(size_t)i] = C.coff; h_out[(size_t)i] = C.out; h_m[(size_t)i] = C.m; h_ld[(size_t)i] = C.ld;
                for (int t0 = 0; t0 < C.m; t0 += tr_rows) {
                    const int tlast = std::min(C.m, t0 + tr_rows) - 1;
                    const int ncol = kDcLeaves + tlast;               // positions u < 10 + t
                    for (int u0 = 0; u0 < ncol; u0 += tr_cols) jobs.push_back(make_int4((int)i, t0, u0, 0));
                }
                if (stats) stats->pairs += (int64_t)C.m * kDcLeaves + (int64_t)C.m * (C.m - 1) / 2;
            }
            if (stats) { stats->groups++; stats->jobs += (int64_t)jobs.size(); }
            DPR_HIP(hipMalloc(&d_i64, sizeof(int64_t) * (size_t)(3 * gn)));
            DPR_HIP(hipMalloc(&d_i32, sizeof(int32_t) * (size_t)(2 * gn)));
            DPR_HIP(hipMalloc(&d_jobs, sizeof(int4) * jobs.size()));
            DPR_HIP(hipMalloc(&d_out, sizeof(double) * (size_t)(outsz > 0 ? outsz : 1)));
            DPR_HIP(hipMemcpyAsync(d_i64, h_moff.data(), sizeof(int64_t) * (size_t)gn, hipMemcpyHostToDevice, s));
            DPR_HIP(hipMemcpyAsync(d_i64 + gn, h_coff.data(), sizeof(int64_t) * (size_t)gn, hipMemcpyHostToDevice, s));
            DPR_HIP(hipMemcpyAsync(d_i64 + 2 * gn, h_out.data(), sizeof(int64_t) * (size_t)gn, hipMemcpyHostToDevice, s));
            DPR_HIP(hipMemcpyAsync(d_i32, h_m.data(), sizeof(int32_t) * (size_t)gn, hipMemcpyHostToDevice, s));
            DPR_HIP(hipMemcpyAsync(d_i32 + gn, h_ld.data(), sizeof(int32_t) * (size_t)gn, hipMemcpyHostToDevice, s));
            DPR_HIP(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(int4) * jobs.size(), hipMemcpyHostToDevice, s));
            DPR_HIP(hipMemcpyAsync(d_cl + g0, cl.data() + g0, sizeof(DcCluster) * (size_t)gn, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(dc_cols_kernel, dim3((unsigned)gn), dim3(64), 0, s, p, d_cl + g0, d_members, d_cols, d_clx);
            PairJobs J;
            J.jobs = d_jobs; J.members = d_members; J.cols = d_cols;
            J.cl_moff = d_i64; J.cl_coff = d_i64 + gn; J.cl_out = d_i64 + 2 * gn;
            J.cl_m = d_i32; J.cl_ld = d_i32 + gn; J.out = d_out;
            if (source == DPR_SRC_MSA) { if (int r2 = msa_dist_jobs(*msa, dist_type, J, (int)jobs.size(), s)) return r2; }
            else { if (int r2 = mash_dist_jobs(*mash, J, (int)jobs.size(), s)) return r2; }
            // clusters are sorted by size: the large ones of the group first, a workgroup of 16 wavefronts each
            int64_t nbig = 0;
            while (nbig < gn && cl[(size_t)(g0 + nbig)].m > big_m) ++nbig;
            if (nbig > 0) {
                hipLaunchKernelGGL(dc_cluster_kernel<16>, dim3((unsigned)nbig), dim3(1024), 0, s, p, d_clx, d_cl + g0, d_members, d_out,
                                   d_qid, d_qfrom, d_qdis, d_status, d_trace);
                DPR_HIP(hipGetLastError());
            }
            if (gn > nbig) {
                hipLaunchKernelGGL(dc_cluster_kernel<1>, dim3((unsigned)(gn - nbig)), dim3(64), 0, s, p, d_clx, d_cl + g0 + nbig, d_members, d_out,
                                   d_qid, d_qfrom, d_qdis, d_status, d_trace);
                DPR_HIP(hipGetLastError());
            }
            DPR_HIP(hipStreamSynchronize(s));   // host vectors and group buffers are reused
            cleanup_group();
            g0 = g1;
        }
        int32_t st = 0;
        DPR_HIP(hipMemcpy(&st, d_status, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (st != 0) {
            set_error("divide-and-conquer: no eligible edge with pendant length < 2 in a cluster (distances >= 2?)");
            return DPR_ERR_NOCAND;
        }
        return DPR_OK;
    };
    rc = run();
    cleanup_group();
    void* ptrs[] = { d_members, d_cols, d_clx, d_qid, d_qfrom, d_qdis, d_status, d_cl };
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    return rc;
}

}  // namespace dpr
