// The reference's operator interface for the hot path over the C ABI (see dipper_host.hpp).
#include "dipper_host.hpp"

#include <algorithm>
#include <iostream>
#include <thread>

namespace dipper {

DeviceContext::DeviceContext(int device) { gpuCheck(dpr_create(&ctx, device), "dpr_create"); }
DeviceContext::~DeviceContext() { if (ctx) dpr_destroy(ctx); }

// replaces the tbb::parallel_for packing loop (src/tree_generation.cu:352-362) + MSADeviceArrays::
// allocateDeviceArrays (src/MSA.cu:14-72)
void MSADeviceArrays::allocateDeviceArrays(DeviceContext& dev, const std::vector<std::string>& seqs,
                                           const std::vector<int>& ids)
{
    numSequences = seqs.size();
    if (numSequences < 2) die("ERROR: need at least two sequences");
    // seqLen = length of the sequence in slot 0 (src/MSA.cu:19)
    size_t slot0 = 0;
    for (size_t i = 0; i < numSequences; ++i) if (ids[i] == 0) slot0 = i;
    seqLen = (int)seqs[slot0].size();
    const size_t W = ((size_t)seqLen + 15) / 16;
    std::vector<uint64_t> flat(numSequences * W, 0);
    unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
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
}

void NJDeviceArrays::findNeighbourJoiningTree(DeviceContext& dev, std::vector<std::string>& name, std::ostream& output_)
{
    const int N = d_numSequences;
    std::vector<int32_t> mx((size_t)std::max(N - 2, 1)), my((size_t)std::max(N - 2, 1));
    std::vector<double> bx((size_t)std::max(N - 2, 1)), by((size_t)std::max(N - 2, 1));
    double last = 0.0;
    const int64_t done = dpr_nj_run(dev.ctx, -1, mx.data(), my.data(), bx.data(), by.data(), &last);
    if (done < 0) gpuCheck((int)done, "dpr_nj_run");
    writeNewickFromMerges(output_, name, mx, my, bx, by, last);
}

}  // namespace dipper
