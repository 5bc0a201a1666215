// Host side of the `dipper` command line (plain C++17, no Boost/TBB): input readers, encoders,
// the reference's operator interface for the hot path re-expressed over the C ABI of
// include/dipper_hip.h, and Newick output.
//
// The struct and method names follow src/mash_placement.cuh of the reference so that a reader of
// the reference finds the same vocabulary; the granularity differs (whole matrix / whole run per
// call instead of one row per call), see INTEGRATION.md.
#pragma once
#include <sstream>
#include <cstring>
#include <cmath>
#include <charconv>
#include <ostream>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iosfwd>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dipper_hip.h"

namespace dipper {

struct Param {  // src/mash_placement.cuh:16-32
    uint64_t kmerSize = 15, sketchSize = 1000, threshold = 1, distanceType = 1;
    std::string in = "r", out = "t";
    uint64_t batchSize = 0, backboneSize = 0;
};

[[noreturn]] void die(const std::string& msg);   // prints msg to stderr, exit(1)
void gpuCheck(int rc, const char* what);         // "Gpu_ERROR"-style exit on rc < 0

// FASTA (plain or gz), klib-kseq semantics: name = header up to the first whitespace
// (src/tree_generation.cu:132-154, src/kseq.h).
void readSequences(const std::string& path, std::vector<std::string>& seqs, std::vector<std::string>& names);

// Host threads this process may really use: min(hardware_concurrency, affinity mask, cgroup cpu.max quota).  A GPU box
// reports 256 logical CPUs but grants 16: a pool sized by hardware_concurrency burns the quota of a scheduling period
// in a quarter of it and is then throttled as a whole.
unsigned hostThreads(unsigned cap = 64);

// Fast path of the sequence inputs (-i m / -i r without --add): the records are indexed in the mapped (or inflated)
// text by all host threads and packed straight into the flat 4-bit / 2-bit arrays of the device interface -- no
// per-sequence std::string copies.  Same records, names and codes as readSequences + dpr_pack4 / dpr_pack2
// (klib-kseq semantics); ok = false: the text needs the serial parser (FASTQ), use readSequences.
// storage that is NOT zero-filled when it is sized: the 150 MB of packed words of a 30 000 x 10 000 alignment are written
// completely by the packing threads, and a value-initialising resize() was 70 ms of SERIAL page faults in front of them
template <class T> struct NoInitAlloc {
    using value_type = T;
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
    void deallocate(T* p, size_t) { ::operator delete(p); }
    template <class U, class... A> void construct(U* p, A&&... a) { if (sizeof...(A) > 0) ::new ((void*)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const NoInitAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
using WordVec = std::vector<uint64_t, NoInitAlloc<uint64_t>>;

struct PackedSequences {
    bool ok = false;
    size_t numSequences = 0;
    std::vector<std::string> names;     // by SLOT (ids applied)
    // aligned (4-bit): flat [n][W], seqLen = length of the sequence in slot 0 (src/MSA.cu:19)
    int seqLen = 0;
    // unaligned (2-bit): flat words at off[slot], lens[slot] bases
    WordVec flat;
    std::vector<uint64_t> off, lens;
};
// on_count(n) is called as soon as the number of records is known (the device thread reserves its matrices then).
// ids_of (optional) replaces the seeded shuffle: called once with the record names IN INPUT ORDER, it returns ids[r] = slot of
// input record r (a permutation) -- the --add path puts the backbone's tips into the slots their tree indices name
// (src/tree_generation.cu:271-282).
void readSequencesPacked(const std::string& path, bool aligned, long long seed, PackedSequences& out,
                         void (*on_count)(size_t n, void* user) = nullptr, void* user = nullptr,
                         const std::function<std::vector<int>(const std::vector<std::string>&)>* ids_of = nullptr);

// the per-sequence encoders of the general path (seqs[i] goes to slot ids[i]); the fast path must produce the same arrays
void packAligned(const std::vector<std::string>& seqs, const std::vector<int>& ids, std::vector<uint64_t>& flat, int& seqLen);
void packUnaligned(const std::vector<std::string>& seqs, const std::vector<int>& ids, std::vector<uint64_t>& flat,
                   std::vector<uint64_t>& off, std::vector<uint64_t>& lens);

// Input-order shuffle of the reference (src/tree_generation.cu:341-344,470-473): ids[i] = slot of
// input sequence i.  seed < 0 keeps the input order.
std::vector<int> shuffledIds(size_t n, long long seed);

// PHYLIP distance matrix (lower-triangular or square), src/matrix_reader.cu:23-45 +
// src/tree_generation.cu:608-611: values parsed with float precision (stof).
struct MatrixReader {
    int numSequences = 0;
    std::vector<std::string> name;
    std::vector<double> lower;  // row i has i entries
    void read(const std::string& path);
};

// ---- several GPUs: one PROCESS per rank (no reference counterpart: src/tree_generation.cu:240-245 selects one device) ---------
// `dipper --gpus G`: once the input has been read and BEFORE the first GPU call the command forks G rank processes around one
// anonymous shared mapping (the packed input is shared copy-on-write: read once, packed once); every rank creates its context on
// its own device and joins the others through that region (dpr_comm_init_shared: RCCL over xGMI for one rank per GPU, device
// windows over hipIpc for ranks that share a device).  The mode's multi-rank plan then runs inside the library: NJ with
// replicated / unit-sharded / row-sharded matrices, placement and --add with the distance rows of a batch sharded + one
// all-gather per batch, divide-and-conquer with query shares and clusters dealt to the ranks.  Every rank ends with the whole
// tree; rank 0 writes it.  The launcher (the original process) never touches the GPU: it waits for its ranks, and if one fails
// it raises the region's failure word (the others then leave their collectives with an error instead of hanging) and exits 1.
// `--rank R --world G --rendezvous NAME`: the same for ranks started by somebody else (mpirun, srun, a test): the region is
// the POSIX shared memory object NAME.
struct RankInfo {
    int rank = 0, world = 1;
    void* region = nullptr;      // DPR_COMM_SHARED_BYTES of shared memory
    int device = 0;              // this rank's GPU
    int transport = 0;           // 0 auto, 1 rccl, 2 ipc
};
RankInfo& rankInfo();
struct RankOptions {
    int gpus = 1;                // --gpus
    std::vector<int> devices;    // --devices a,b,..  (default: --device + rank)
    int transport = 0;           // --transport auto|rccl|ipc
    int ext_rank = -1, ext_world = 0;      // --rank / --world: ranks started from outside
    std::string rendezvous;      // --rendezvous
    bool multi() const { return gpus > 1 || ext_world > 1; }
};
// Returns in every RANK with rankInfo() filled in (ranks > 0: the progress lines on std::cerr are switched off; errors still
// reach stderr).  In the launcher it does not return.  Must be called with no other thread alive and before any GPU call.
void startRanks(const RankOptions& o, int base_device);
// rank 0's closing line of a multi-rank run: "Ranks: G (transport, N device collectives; NJ plan ...)"; nothing with one rank
void printRankSummary(dpr_ctx* ctx);

struct DeviceContext {  // replaces cudaSetDevice (src/tree_generation.cu:240-245)
    dpr_ctx* ctx = nullptr;
    explicit DeviceContext(int device);
    ~DeviceContext();
};
// The HIP runtime start-up (~0.2 s) overlaps the reading of the input: the context is created on a
// helper thread and joined by get().
struct AsyncDeviceContext {
    explicit AsyncDeviceContext(int device);
    ~AsyncDeviceContext();
    DeviceContext& get();
    // ask the helper thread to allocate the NJ matrices for n tips (dpr_reserve_nj) once the context exists; returns at once
    void reserveNJ(size_t n);
    // milliseconds dpr_create took on the helper thread (HIP runtime start-up + code object load); valid after get()
    double createMs() const;
private:
    struct Impl;
    Impl* impl;
};

struct MSADeviceArrays {  // src/mash_placement.cuh:87-98
    size_t numSequences = 0;
    int seqLen = 0;
    // seqs[i] goes to slot ids[i]; packs with the 4-bit encoder in parallel and uploads
    void allocateDeviceArrays(DeviceContext& dev, const std::vector<std::string>& seqs, const std::vector<int>& ids);
    void allocateDeviceArrays(DeviceContext& dev, const PackedSequences& packed);      // already packed by readSequencesPacked
};

struct NJDeviceArrays {  // src/mash_placement.cuh:199-212
    int d_numSequences = 0;
    void getDismatrix(DeviceContext& dev, int numSequences, Param& params, MatrixReader* matrixReader);
    void findNeighbourJoiningTree(DeviceContext& dev, std::vector<std::string>& name, std::ostream& output_);
};

struct MashDeviceArrays {  // src/mash_placement.cuh:34-50
    size_t numSequences = 0;
    // 2-bit packing in parallel (src/tree_generation.cu:480-490) + upload; seqs[i] goes to slot ids[i]
    void allocateDeviceArrays(DeviceContext& dev, const std::vector<std::string>& seqs, const std::vector<int>& ids);
    void allocateDeviceArrays(DeviceContext& dev, const PackedSequences& packed);
    void sketchConstructionOnGpu(DeviceContext& dev, Param& params);
};

// Newick import with the reference's id assignment (src/tree.cpp:216-361): leaves get idx 0,1,.. in
// order of appearance, internal nodes totalLeaves, totalLeaves+1, .. in order of '(' and the name
// "node_<idx>"; branch lengths parsed with float precision (stof); the root's length is set to 0.
struct Node {
    std::string name;
    int idx = -1;
    double bl = 0.0;
    int parent = -1;             // index into Tree::nodes
    std::vector<int> children;   // indices into Tree::nodes
};
struct Tree {
    std::vector<Node> nodes;     // creation order (pre-order)
    int root = -1;
    size_t m_numLeaves = 0;
    Tree(const std::string& newick, size_t totalLeaves);
    int findLeaf(const std::string& name) const;   // node index or -1
};

// k-closest placement (src/mash_placement.cuh:167-197); with exact = true the same host object drives
// the exact mode of PlacementDeviceArrays (src/mash_placement.cuh:137-165, src/placement.cu): identical
// adjacency arrays and printTree (src/placement.cu:454-505 == src/placement_close_k.cu:568-643)
struct KPlacementDeviceArrays {
    int numSequences = 0, backboneSize = -1, bd = 2;
    bool exact = false;
    std::vector<int32_t> h_head, h_e, h_nxt, h_belong;
    std::vector<double> h_len;
    void allocateDeviceArrays(size_t num, int backboneSize = -1);
    void initializeDeviceArrays(const Tree& t);   // backbone -> forward-star adjacency (src/placement_close_k.cu:126-264)
    void findPlacementTree(DeviceContext& dev, Param& params);
    void addQuery(DeviceContext& dev, Param& params);
    void printTree(const std::vector<std::string>& name, std::ostream& output_);
};

// Divide-and-conquer mode (src/mash_placement.cuh KPlacementDeviceArraysDC; findBackboneTreeDC /
// findClustersDC / findClusterTreeDC / printTreeDC, src/divide_and_conquer/placement_close_k.cu:731-1535,
// 651-711).  One call of the C ABI runs the three phases; the tree is printed from node totalNumSequences.
struct KPlacementDeviceArraysDC : KPlacementDeviceArrays {
    int totalNumSequences = 0;
    std::vector<int32_t> clusterID;
    void allocateDeviceArraysDC(size_t num, size_t totalNum);
    void findTreeDC(DeviceContext& dev, Param& params);
    void printTreeDC(const std::vector<std::string>& name, std::ostream& output_) { printTree(name, output_); }
};

// THE number formatter of every Newick writer of this build: the reference streams its doubles with the default
// ostream settings (6 significant digits, %g style; src/neighborJoining.cu:252-270, src/placement_close_k.cu:568-643)
// a branch length as `os << double` prints it (printf %g, precision 6 -- the reference's std::cout << double), without the stream's
// formatting machinery: std::to_chars with that format is specified to give printf's characters (550 000 tips: 195 -> ~40 ms)
inline size_t fmtLength(char* b, size_t cap, double v)
{
    if (std::isfinite(v)) return (size_t)(std::to_chars(b, b + cap, v, std::chars_format::general, 6).ptr - b);
    std::ostringstream o;
    o << v;
    const std::string t = o.str();
    std::memcpy(b, t.data(), t.size() < cap ? t.size() : cap);
    return t.size() < cap ? t.size() : cap;
}
inline void putLength(std::ostream& os, double v) { char b[48]; os.write(b, (std::streamsize)fmtLength(b, sizeof b, v)); }
// Newick text assembled in memory and written once
struct TextBuf {
    std::string s;
    void put(char c) { s.push_back(c); }
    void put(const std::string& t) { s.append(t); }
    void putLength(double v) { char b[48]; s.append(b, fmtLength(b, sizeof b, v)); }
};
// extra timing lines of the command on stderr: the category `cli` of the library's one logging variable, DPR_LOG
inline bool cliLog()
{
    const char* e = std::getenv("DPR_LOG");
    if (!e) return false;
    const std::string s = std::string(",") + e + ",";
    return s.find(",cli,") != std::string::npos || s.find(",cli=") != std::string::npos;
}

// Newick text of an NJ merge log (bookkeeping + print of src/neighborJoining.cu:233-270), iterative.
void writeNewickFromMerges(std::ostream& os, const std::vector<std::string>& name, const std::vector<int32_t>& mx,
                           const std::vector<int32_t>& my, const std::vector<double>& bx,
                           const std::vector<double>& by, double last_d);

}  // namespace dipper
