// The reference's operator interface for the hot path over the C ABI (see dipper_host.hpp).
#include "dipper_host.hpp"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <iostream>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace dipper {

DeviceContext::DeviceContext(int device)
{
    gpuCheck(dpr_create(&ctx, device), "dpr_create");
    // one of several ranks (startRanks): join the others before any input reaches the device
    const RankInfo& ri = rankInfo();
    if (ri.world > 1) gpuCheck(dpr_comm_init_shared(ctx, ri.rank, ri.world, ri.region, DPR_COMM_SHARED_BYTES, ri.transport), "dpr_comm_init_shared");
}
DeviceContext::~DeviceContext() { if (ctx) dpr_destroy(ctx); }

void printRankSummary(dpr_ctx* ctx)
{
    if (rankInfo().world <= 1 || !ctx) return;
    int transport = 0, rank = 0, nranks = 1;
    int64_t coll = 0;
    dpr_comm_stats(ctx, &transport, &coll);
    dpr_comm_info(ctx, &rank, &nranks);
    static const char* const names[] = { "none", "rccl", "ipc", "local" };
    std::cerr << "Ranks: " << nranks << " (transport " << names[transport & 3] << ", " << coll << " device collectives)\n";
}

struct AsyncDeviceContext::Impl {
    std::thread th, warm;
    DeviceContext* dev = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    size_t reserve_n = 0;
    bool closing = false;
    double create_ms = 0;
};
double AsyncDeviceContext::createMs() const { return impl->create_ms; }
AsyncDeviceContext::AsyncDeviceContext(int device) : impl(new Impl)
{
    Impl* q = impl;
    impl->th = std::thread([q, device] {
        const auto tc0 = std::chrono::steady_clock::now();
        q->dev = new DeviceContext(device);
        q->create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count();
        // (helper of the helper: the one-time graph set-up of the HIP runtime runs beside the allocation below, the uploads
        //  and the distance kernels; joined when the context goes away)
        q->warm = std::thread([ctx = q->dev->ctx] { (void)dpr_warm_graphs(ctx); });
        std::unique_lock<std::mutex> lk(q->mu);
        q->cv.wait(lk, [q] { return q->reserve_n != 0 || q->closing; });
        const size_t n = q->reserve_n;
        lk.unlock();
        if (n) (void)dpr_reserve_nj(q->dev->ctx, (int64_t)n);      // best effort: dpr_dist_matrix allocates what is missing
    });
}
void AsyncDeviceContext::reserveNJ(size_t n)
{
    { std::lock_guard<std::mutex> lk(impl->mu); impl->reserve_n = n; }
    impl->cv.notify_all();
}
DeviceContext& AsyncDeviceContext::get()
{
    { std::lock_guard<std::mutex> lk(impl->mu); impl->closing = true; }
    impl->cv.notify_all();
    if (impl->th.joinable()) impl->th.join();
    // the graph warm-up (~30 ms, started when the context came up) is long over when the input has been read; joined here so
    // that every dpr_* call that follows is the only one on this context (dpr_warm_graphs is the one entry point that may
    // run beside others -- it touches a private stream only -- and here it ran beside dpr_reserve_nj alone)
    if (impl->warm.joinable()) impl->warm.join();
    return *impl->dev;
}
AsyncDeviceContext::~AsyncDeviceContext()
{
    { std::lock_guard<std::mutex> lk(impl->mu); impl->closing = true; }
    impl->cv.notify_all();
    if (impl->th.joinable()) impl->th.join();
    if (impl->warm.joinable()) impl->warm.join();
    delete impl->dev;
    delete impl;
}

void MSADeviceArrays::allocateDeviceArrays(DeviceContext& dev, const PackedSequences& packed)
{
    numSequences = packed.numSequences;
    if (numSequences < 2) die("ERROR: need at least two sequences");
    seqLen = packed.seqLen;
    gpuCheck(dpr_set_msa(dev.ctx, packed.flat.data(), (int64_t)numSequences, (int64_t)seqLen), "dpr_set_msa");
}

void MashDeviceArrays::allocateDeviceArrays(DeviceContext& dev, const PackedSequences& packed)
{
    numSequences = packed.numSequences;
    if (numSequences < 2) die("ERROR: need at least two sequences");
    gpuCheck(dpr_set_reads(dev.ctx, packed.flat.data(), packed.off.data(), packed.lens.data(), (int64_t)numSequences), "dpr_set_reads");
}

void packAligned(const std::vector<std::string>& seqs, const std::vector<int>& ids, std::vector<uint64_t>& flat, int& seqLen)
{
    const size_t numSequences = seqs.size();
    // seqLen = length of the sequence in slot 0 (src/MSA.cu:19)
    size_t slot0 = 0;
    for (size_t i = 0; i < numSequences; ++i) if (ids[i] == 0) slot0 = i;
    seqLen = (int)seqs[slot0].size();
    const size_t W = ((size_t)seqLen + 15) / 16;
    flat.assign(numSequences * W, 0);
    unsigned nt = hostThreads(32);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&, t] {
            std::vector<uint64_t> tmp;
            for (size_t i = t; i < numSequences; i += nt) {
                const std::string& s = seqs[i];
                tmp.assign((s.size() + 15) / 16 + 1, 0);
                dpr_pack4(s.data(), s.size(), tmp.data());
                // positions beyond a shorter sequence read as code 0 in the reference's flat buffer
                // only by accident (neighbouring data); here they are padded with the invalid code 4.
                uint64_t* dst = flat.data() + (size_t)ids[i] * W;
                const size_t have = (s.size() + 15) / 16;
                for (size_t w = 0; w < W; ++w) dst[w] = w < have ? tmp[w] : 0x4444444444444444ull;
                if (s.size() < (size_t)seqLen && have > 0 && have <= W) {
                    // tail of the last present word: mark the missing bases invalid
                    const size_t r = s.size() % 16;
                    if (r) dst[have - 1] |= 0x4444444444444444ull << (4 * r);
                }
            }
        });
    for (auto& th : pool) th.join();
}

// replaces the tbb::parallel_for packing loop (src/tree_generation.cu:352-362) + MSADeviceArrays::
// allocateDeviceArrays (src/MSA.cu:14-72)
void MSADeviceArrays::allocateDeviceArrays(DeviceContext& dev, const std::vector<std::string>& seqs,
                                           const std::vector<int>& ids)
{
    numSequences = seqs.size();
    if (numSequences < 2) die("ERROR: need at least two sequences");
    std::vector<uint64_t> flat;
    packAligned(seqs, ids, flat, seqLen);
    gpuCheck(dpr_set_msa(dev.ctx, flat.data(), (int64_t)numSequences, (int64_t)seqLen), "dpr_set_msa");
}

void NJDeviceArrays::getDismatrix(DeviceContext& dev, int numSequences, Param& params, MatrixReader* matrixReader)
{
    d_numSequences = numSequences;
    if (params.in == "d") {
        gpuCheck(dpr_set_matrix_lower(dev.ctx, matrixReader->lower.data(), numSequences), "dpr_set_matrix_lower");
        gpuCheck(dpr_dist_matrix(dev.ctx, DPR_SRC_MATRIX, 0, 0), "dpr_dist_matrix");
    } else if (params.in == "m") {
        gpuCheck(dpr_dist_matrix(dev.ctx, DPR_SRC_MSA, (int)params.distanceType, 0), "dpr_dist_matrix");
    } else {
        gpuCheck(dpr_dist_matrix(dev.ctx, DPR_SRC_MASH, 0, (int)params.kmerSize), "dpr_dist_matrix");
    }
    if (rankInfo().world > 1) {
        char plan[256] = "";
        dpr_get_nj_multi_info(dev.ctx, plan, (int)sizeof plan);
        std::cerr << "NJ over " << rankInfo().world << " ranks: " << plan << "\n";
    }
}

void NJDeviceArrays::findNeighbourJoiningTree(DeviceContext& dev, std::vector<std::string>& name, std::ostream& output_)
{
    const int N = d_numSequences;
    std::vector<int32_t> mx((size_t)std::max(N - 2, 1)), my((size_t)std::max(N - 2, 1));
    std::vector<double> bx((size_t)std::max(N - 2, 1)), by((size_t)std::max(N - 2, 1));
    double last = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t done = dpr_nj_run(dev.ctx, -1, mx.data(), my.data(), bx.data(), by.data(), &last);
    if (done < 0) gpuCheck((int)done, "dpr_nj_run");
    const auto t1 = std::chrono::steady_clock::now();
    writeNewickFromMerges(output_, name, mx, my, bx, by, last);
    if (cliLog()) {
        double dist_ms = 0, nj_ms = 0;
        dpr_get_timing(dev.ctx, &dist_ms, &nj_ms);
        std::cerr << "  device: distances " << dist_ms << " ms, NJ " << nj_ms << " ms; dpr_nj_run call "
                  << std::chrono::duration<double, std::milli>(t1 - t0).count() << " ms; Newick text "
                  << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count() << " ms\n";
    }
}

void packUnaligned(const std::vector<std::string>& seqs, const std::vector<int>& ids, std::vector<uint64_t>& flat,
                   std::vector<uint64_t>& off, std::vector<uint64_t>& lens)
{
    const size_t numSequences = seqs.size();
    lens.assign(numSequences, 0); off.assign(numSequences, 0);
    std::vector<uint64_t> nw(numSequences);
    for (size_t i = 0; i < numSequences; ++i) {
        lens[(size_t)ids[i]] = seqs[i].size();
        nw[(size_t)ids[i]] = (seqs[i].size() + 31) / 32;
    }
    uint64_t total = 0;
    for (size_t s = 0; s < numSequences; ++s) { off[s] = total; total += nw[s]; }  // exclusive scan (src/mash.cu:109-119)
    flat.assign(total + 1, 0);
    unsigned nt = hostThreads(32);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&, t] {
            for (size_t i = t; i < numSequences; i += nt)
                if (!seqs[i].empty()) dpr_pack2(seqs[i].data(), seqs[i].size(), flat.data() + off[(size_t)ids[i]]);
        });
    for (auto& th : pool) th.join();
}

void MashDeviceArrays::allocateDeviceArrays(DeviceContext& dev, const std::vector<std::string>& seqs,
                                            const std::vector<int>& ids)
{
    numSequences = seqs.size();
    if (numSequences < 2) die("ERROR: need at least two sequences");
    std::vector<uint64_t> lens, off, flat;
    packUnaligned(seqs, ids, flat, off, lens);
    gpuCheck(dpr_set_reads(dev.ctx, flat.data(), off.data(), lens.data(), (int64_t)numSequences), "dpr_set_reads");
}

void MashDeviceArrays::sketchConstructionOnGpu(DeviceContext& dev, Param& params)
{
    gpuCheck(dpr_sketch(dev.ctx, (int)params.kmerSize, (int)params.sketchSize, nullptr), "dpr_sketch");
}

// ---- Newick import ---------------------------------------------------------------------------------
Tree::Tree(const std::string& newick_in, size_t totalLeaves)
{
    // (the text without line breaks; copied only if it has any.  500 000 tips: 17 MB, a million nodes -- the vector is sized once, a
    //  node's children get room for two, nodes are moved into place)
    std::string stripped;
    if (newick_in.find_first_of("\n\r") != std::string::npos) {
        stripped.reserve(newick_in.size());
        for (char c : newick_in) if (c != '\n' && c != '\r') stripped.push_back(c);
    }
    const std::string& s = stripped.empty() ? newick_in : stripped;
    nodes.reserve(2 * totalLeaves + 2);
    size_t nextInternal = totalLeaves, nextLeaf = 0;
    std::vector<int> stack;
    int last = -1;  // node whose label / length is being read
    auto parse_len = [&](size_t& i) {
        std::string num;
        size_t j = i + 1;
        while (j < s.size() && s[j] != ',' && s[j] != ')' && s[j] != '(' && s[j] != ';') {
            const char c = s[j];
            if (std::isdigit((unsigned char)c) || c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+') num.push_back(c);
            ++j;
        }
        i = j - 1;
        return num.empty() ? 0.0 : (double)std::strtof(num.c_str(), nullptr);
    };
    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        if (c == '(') {
            Node nd;
            nd.idx = (int)nextInternal++;
            nd.name = "node_" + std::to_string(nd.idx);
            nd.parent = stack.empty() ? -1 : stack.back();
            nd.children.reserve(2);
            const int par = nd.parent;
            nodes.push_back(std::move(nd));
            const int id = (int)nodes.size() - 1;
            if (par >= 0) nodes[(size_t)par].children.push_back(id); else root = id;
            stack.push_back(id);
            last = -1;
        } else if (c == ')') {
            if (stack.empty()) die("ERROR: incorrect Newick format!");
            last = stack.back();
            stack.pop_back();
            // an internal label / support value after ')' is ignored, as in the reference
            size_t j = i + 1;
            while (j < s.size() && s[j] != ':' && s[j] != ',' && s[j] != ')' && s[j] != ';') ++j;
            i = j - 1;
        } else if (c == ',') {
            last = -1;
        } else if (c == ':') {
            const double v = parse_len(i);
            if (last >= 0) nodes[(size_t)last].bl = v;
        } else if (c == ';') {
            break;
        } else if (std::isspace((unsigned char)c)) {
            continue;
        } else {
            // leaf label (quoted labels keep their inner text)
            std::string name;
            size_t j = i;
            if (c == '\'') {
                ++j;
                while (j < s.size() && s[j] != '\'') name.push_back(s[j++]);
                ++j;
            } else {
                while (j < s.size() && s[j] != ':' && s[j] != ',' && s[j] != ')' && s[j] != '(' && s[j] != ';') name.push_back(s[j++]);
            }
            i = j - 1;
            if (stack.empty()) die("ERROR: incorrect Newick format!");
            Node nd;
            nd.idx = (int)nextLeaf++;
            nd.name = std::move(name);
            nd.parent = stack.back();
            const int par = nd.parent;
            nodes.push_back(std::move(nd));
            last = (int)nodes.size() - 1;
            nodes[(size_t)par].children.push_back(last);
        }
    }
    if (!stack.empty()) die("ERROR: incorrect Newick format!");
    if (root < 0) die("ERROR: Tree found empty!");
    nodes[(size_t)root].bl = 0;
    m_numLeaves = nextLeaf;
}

int Tree::findLeaf(const std::string& name) const
{
    for (size_t i = 0; i < nodes.size(); ++i)
        if (nodes[i].children.empty() && nodes[i].name == name) return (int)i;
    return -1;
}

// ---- placement ----------------------------------------------------------------------------------------
void KPlacementDeviceArrays::allocateDeviceArrays(size_t num, int backbone)
{
    numSequences = (int)num;
    backboneSize = backbone;
    bd = 2;
    h_head.assign(num * 2, -1);
    h_e.assign(num * 8, -1);
    h_nxt.assign(num * 8, -1);
    h_belong.assign(num * 8, -1);
    h_len.assign(num * 8, 2.0);
}

void KPlacementDeviceArrays::initializeDeviceArrays(const Tree& t)
{
    // post-order DFS; per tree edge two directed slots (child->parent, parent->child), each pushed
    // at the front of its source's list (src/placement_close_k.cu:160-183)
    size_t edgeCount = 0;
    struct Frame { int node; size_t next; };
    std::vector<Frame> st;
    st.push_back(Frame{ t.root, 0 });
    while (!st.empty()) {
        Frame& f = st.back();
        const Node& nd = t.nodes[(size_t)f.node];
        if (f.next < nd.children.size()) {
            const int c = nd.children[f.next++];
            st.push_back(Frame{ c, 0 });
            continue;
        }
        if (nd.parent >= 0) {
            const int x = nd.idx, y = t.nodes[(size_t)nd.parent].idx;
            if (edgeCount + 2 > h_e.size()) die("ERROR: backbone tree does not fit the allocated arrays");
            h_e[edgeCount] = y; h_len[edgeCount] = nd.bl; h_belong[edgeCount] = x;
            h_nxt[edgeCount] = h_head[(size_t)x]; h_head[(size_t)x] = (int32_t)edgeCount; edgeCount++;
            h_e[edgeCount] = x; h_len[edgeCount] = nd.bl; h_belong[edgeCount] = y;
            h_nxt[edgeCount] = h_head[(size_t)y]; h_head[(size_t)y] = (int32_t)edgeCount; edgeCount++;
        }
        st.pop_back();
    }
}

static int sourceOf(const Param& params)
{
    return params.in == "r" ? DPR_SRC_MASH : (params.in == "m" ? DPR_SRC_MSA : DPR_SRC_MATRIX);
}

// the reference's two timing lines (src/placement_close_k.cu:852-853,985-986).  With Mash input the distance rows of the
// next batch are computed beside the tree kernels: the first line is then the time the tree kernels waited for rows
// (the two lines still add up to the run) and a third line gives the batches' own, overlapped duration.
static void printPlaceTiming(DeviceContext& dev)
{
    double dist_ms = 0, tree_ms = 0, busy_ms = 0;
    int overlapped = 0;
    dpr_get_place_timing(dev.ctx, &dist_ms, &tree_ms);
    dpr_get_place_overlap(dev.ctx, &overlapped, &busy_ms);
    std::cerr << "Distance Operation Time " << (long long)dist_ms << " ms\n";
    std::cerr << "Tree Operation Time " << (long long)tree_ms << " ms\n";
    if (overlapped) {
        int64_t batches = 0, beside = 0;
        dpr_get_place_policy(dev.ctx, &batches, &beside);
        std::cerr << "Distance batches overlapped with tree operations: " << beside << " of " << batches << ", " << (long long)busy_ms << " ms in flight\n";
    }
}

void KPlacementDeviceArrays::findPlacementTree(DeviceContext& dev, Param& params)
{
    if (exact) {
        gpuCheck(dpr_place_exact_run(dev.ctx, sourceOf(params), (int)params.distanceType, (int)params.kmerSize, numSequences,
                                     h_head.data(), h_e.data(), h_nxt.data(), h_belong.data(), h_len.data()),
                 "dpr_place_exact_run");
        double dist_ms = 0, tree_ms = 0;
        dpr_get_timing(dev.ctx, &dist_ms, &tree_ms);    // (the exact mode's distance rows are produced inside its per-tip loop)
        std::cerr << "Distance + Tree Operation Time " << (long long)tree_ms << " ms\n";
        return;
    }
    gpuCheck(dpr_place_run(dev.ctx, sourceOf(params), (int)params.distanceType, (int)params.kmerSize, 2, numSequences,
                           h_head.data(), h_e.data(), h_nxt.data(), h_belong.data(), h_len.data()), "dpr_place_run");
    printPlaceTiming(dev);
}

void KPlacementDeviceArrays::addQuery(DeviceContext& dev, Param& params)
{
    gpuCheck(dpr_place_run(dev.ctx, sourceOf(params), (int)params.distanceType, (int)params.kmerSize, backboneSize,
                           numSequences, h_head.data(), h_e.data(), h_nxt.data(), h_belong.data(), h_len.data()),
             "dpr_place_run");
    printPlaceTiming(dev);
}

void KPlacementDeviceArraysDC::allocateDeviceArraysDC(size_t num, size_t totalNum)
{
    allocateDeviceArrays(totalNum);          // arrays sized by the total tip count, node ids start there
    backboneSize = (int)num;
    totalNumSequences = (int)totalNum;
    clusterID.assign(totalNum, -1);
}

void KPlacementDeviceArraysDC::findTreeDC(DeviceContext& dev, Param& params)
{
    gpuCheck(dpr_dc_run(dev.ctx, sourceOf(params), (int)params.distanceType, (int)params.kmerSize, totalNumSequences,
                        backboneSize, 0, h_head.data(), h_e.data(), h_nxt.data(), h_belong.data(), h_len.data(),
                        clusterID.data()), "dpr_dc_run");
    int64_t counts[5] = { 0, 0, 0, 0, 0 };
    double ms[3] = { 0, 0, 0 };
    dpr_get_dc_stats(dev.ctx, counts, ms);
    std::cerr << "Finished backbone construction in: " << (long long)ms[0] << " ms\n";
    std::cerr << "Finished clustering in: " << (long long)ms[1] << " ms\n";
    std::cerr << "Finished cluster trees in: " << (long long)ms[2] << " ms (" << counts[0] << " clusters, largest "
              << counts[1] << ")\n";
}

// printTree (src/placement_close_k.cu:568-643), iterative: root = node numSequences+bd-2, children in
// adjacency-list order, the edge back to the parent skipped; a node with a single adjacency entry is a leaf.
void KPlacementDeviceArrays::printTree(const std::vector<std::string>& name, std::ostream& output_)
{
    // (a node's adjacency entries other than the one it was entered through, gathered once into one array: no allocation per node)
    struct Frame { int node, from; size_t first, count, next; };
    std::vector<int> pos;
    pos.reserve((size_t)numSequences * 4);
    auto make = [&](int node, int from) {
        Frame f{ node, from, pos.size(), 0, 0 };
        for (int i = h_head[(size_t)node]; i != -1; i = h_nxt[(size_t)i])
            if (h_e[(size_t)i] != from) pos.push_back(i);
        f.count = pos.size() - f.first;
        return f;
    };
    auto is_internal = [&](int node) { return h_nxt[(size_t)h_head[(size_t)node]] != -1; };
    const int root = numSequences + bd - 2;
    std::vector<Frame> st;
    if (!is_internal(root)) { output_ << name[(size_t)root] << ";\n"; return; }
    TextBuf out;
    out.s.reserve((size_t)numSequences * 40);
    out.put('(');
    st.push_back(make(root, -1));
    while (!st.empty()) {
        Frame& f = st.back();
        if (f.next == f.count) {
            pos.resize(f.first);             // (frames end in the order they began: the array is a stack)
            st.pop_back();
            if (st.empty()) break;
            Frame& up = st.back();
            out.put(':'); out.putLength(h_len[(size_t)pos[up.first + up.next - 1]]); out.put(up.next == up.count ? ')' : ',');
            continue;
        }
        const int slot = pos[f.first + f.next++];
        const int child = h_e[(size_t)slot];
        if (is_internal(child)) {
            out.put('(');
            const int parent = f.node;       // (push_back may move the frames)
            st.push_back(make(child, parent));
        } else {
            out.put(name[(size_t)child]); out.put(':'); out.putLength(h_len[(size_t)slot]); out.put(f.next == f.count ? ')' : ',');
        }
    }
    out.put(";\n");
    output_.write(out.s.data(), (std::streamsize)out.s.size());
}

}  // namespace dipper
