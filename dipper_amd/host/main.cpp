// dipper -- command line of the MI355X-native distance-based phylogeny engine.
// Keeps the flags, formats, banners and exit codes of the reference's main()
// (src/tree_generation.cu:33-99,159-646); host orchestration is plain C++ over the C ABI.
#include "dipper_host.hpp"

#include <thread>
#include <string_view>
#include <algorithm>

#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <unordered_map>

using namespace dipper;

static const char* kHelp =
    "DIPPER Command Line Arguments:\n\n"
    "Required Options:\n"
    "  -i [ --input-format ] arg   Input format:\n"
    "                                d - distance matrix in PHYLIP format\n"
    "                                r - unaligned sequences in FASTA format\n"
    "                                m - aligned sequences in FASTA format\n"
    "  -I [ --input-file ] arg     Input file path:\n"
    "                                PHYLIP format for distance matrix\n"
    "                                FASTA format for aligned or unaligned sequences\n"
    "  -O [ --output-file ] arg    Output file path\n\n"
    "Optional Options:\n"
    "  -o [ --output-format ] arg  Output format:\n"
    "                                t - phylogenetic tree in Newick format (default)\n"
    "                                d - distance matrix in PHYLIP format (coming soon)\n"
    "  -m [ --algorithm ] arg      Algorithm selection:\n"
    "                                0 - default mode\n"
    "                                1 - force placement\n"
    "                                2 - force conventional NJ\n"
    "                                3 - force divide-and-conquer\n"
    "  -p [ --placement-mode ] arg Placement mode:\n"
    "                                0 - exact mode\n"
    "                                1 - k-closest mode (default)\n"
    "  -k [ --kmer-size ] arg      K-mer size:\n"
    "                                Valid range: 2-15 (default: 15)\n"
    "  -s [ --sketch-size ] arg    Sketch size (default: 1000)\n"
    "  -d [ --distance-type ] arg  Distance type to calculate:\n"
    "                                1 - uncorrected\n"
    "                                2 - JC (default)\n"
    "                                3 - Tajima-Nei\n"
    "                                4 - K2P\n"
    "                                5 - Tamura\n"
    "                                6 - Jinnei\n"
    "  -a [ --add ]                Add query to backbone using k-closest placement\n"
    "  -t [ --input-tree ] arg     Input backbone tree (Newick format), required with --add option\n"
    "  -h [ --help ]               Print this help message\n\n"
    "MI355X build options:\n"
    "  --seed arg                  Seed of the input-order shuffle (reference: time(NULL));\n"
    "                              default 1, negative keeps the input order\n"
    "  --device arg                GPU index (default 0; the reference hard-codes 1)\n"
    "  --gpus arg                  Number of GPUs: one rank process per GPU, started by this command\n"
    "                              once the input is read (devices --device, --device+1, ...)\n"
    "  --devices arg               Comma-separated GPU index of every rank (the same index twice:\n"
    "                              ranks share that GPU -- rehearsal of the multi-GPU paths)\n"
    "  --transport arg             auto (default) | rccl | ipc: how the ranks exchange data\n"
    "  --rank arg --world arg --rendezvous arg\n"
    "                              this process is rank `rank` of `world` ranks started from outside;\n"
    "                              they meet in the POSIX shared memory object `rendezvous`\n";

struct Opt { const char* lng; char sht; bool has_arg; };
static const Opt kOpts[] = {
    { "input-format", 'i', true }, { "input-file", 'I', true }, { "output-file", 'O', true },
    { "output-format", 'o', true }, { "algorithm", 'm', true }, { "placement-mode", 'p', true },
    { "kmer-size", 'k', true }, { "sketch-size", 's', true }, { "distance-type", 'd', true },
    { "add", 'a', false }, { "input-tree", 't', true }, { "help", 'h', false },
    { "seed", 0, true }, { "device", 0, true }, { "gpus", 0, true }, { "devices", 0, true }, { "transport", 0, true },
    { "rank", 0, true }, { "world", 0, true }, { "rendezvous", 0, true }, { "dump-tree", 0, true }, { "dump-fasta", 0, false }, { "dump-lengths", 0, false }, { "dump-packed", 0, true },
};

static void usageError(const std::string& what)
{
    std::cerr << "\033[31m" << what << "\033[0m" << std::endl;
    std::cerr << kHelp << std::endl;
    std::exit(1);
}

static std::map<std::string, std::string> parseArguments(int argc, char** argv)
{
    std::map<std::string, std::string> vm;
    for (int a = 1; a < argc; ++a) {
        std::string tok = argv[a];
        const Opt* opt = nullptr;
        std::string val;
        bool have_val = false;
        if (tok.rfind("--", 0) == 0) {
            std::string nm = tok.substr(2);
            const size_t eq = nm.find('=');
            if (eq != std::string::npos) { val = nm.substr(eq + 1); nm = nm.substr(0, eq); have_val = true; }
            for (const Opt& o : kOpts) if (nm == o.lng) opt = &o;
            if (!opt) usageError("unrecognised option '" + tok + "'");
        } else if (tok.size() >= 2 && tok[0] == '-') {
            for (const Opt& o : kOpts) if (o.sht && tok[1] == o.sht) opt = &o;
            if (!opt) usageError("unrecognised option '" + tok + "'");
            if (tok.size() > 2) { val = tok.substr(2); have_val = true; }
        } else {
            usageError("too many positional options have been specified on the command line");
        }
        if (opt->has_arg) {
            if (!have_val) {
                if (a + 1 >= argc) usageError(std::string("the required argument for option '--") + opt->lng + "' is missing");
                val = argv[++a];
            }
            vm[opt->lng] = val;
        } else {
            vm[opt->lng] = "1";
        }
    }
    return vm;
}

static uint64_t stoiOr(const std::map<std::string, std::string>& vm, const char* key, uint64_t dflt)
{
    // the reference parses every numeric flag with stoi inside try{}catch{} (src/tree_generation.cu:191-208)
    auto it = vm.find(key);
    if (it == vm.end()) return dflt;
    try { return (uint64_t)std::stoi(it->second); } catch (...) { return dflt; }
}

static std::string strOr(const std::map<std::string, std::string>& vm, const char* key, const std::string& dflt)
{
    auto it = vm.find(key);
    return it == vm.end() ? dflt : it->second;
}

int main(int argc, char** argv)
{
    // This process lives for one tree: what the HIP runtime sets up and the kernel driver tears down around it counts.  Two of
    // ROCr's / HIP's own switches, set for THIS process only and only when the caller has not set them: no SDMA queues (the few
    // copies of a run go through shader copies) and at most two hardware queues (the runs use at most two streams at a time):
    // the whole command, 30 000 x 10 000 back to back, 987 -> 920 ms (profiles/r3/cli_env_sweep.jsonl; a caller who wants the
    // runtime's defaults sets the two variables himself).
    setenv("HSA_ENABLE_SDMA", "0", 0);
    setenv("GPU_MAX_HW_QUEUES", "2", 0);
    auto inputStart = std::chrono::high_resolution_clock::now();
    auto vm = parseArguments(argc, argv);
    if (vm.count("help")) { std::cerr << kHelp << std::endl; return 0; }
    if (vm.count("dump-tree")) {
        // developer aid (no GPU): parse --input-tree with totalLeaves = arg and print the node table
        std::ifstream tf(strOr(vm, "input-tree", ""));
        if (!tf) { std::cerr << "ERROR: Unable to open input tree file: " << strOr(vm, "input-tree", "") << "\n"; return 1; }
        std::string nwk;
        std::getline(tf, nwk);
        Tree t(nwk, (size_t)stoiOr(vm, "dump-tree", 0));
        std::printf("%zu %zu %d\n", t.nodes.size(), t.m_numLeaves, t.nodes[(size_t)t.root].idx);
        for (const Node& nd : t.nodes)
            std::printf("%d %d %.17g %d %s\n", nd.idx, nd.parent >= 0 ? t.nodes[(size_t)nd.parent].idx : -1, nd.bl,
                        nd.children.empty() ? 1 : 0, nd.name.c_str());
        return 0;
    }
    if (vm.count("dump-lengths")) {
        // developer aid (no GPU): every branch-length token of --input-tree, parsed as a double and written back
        // through the number formatter of this build's Newick writers, one per line
        std::ifstream tf(strOr(vm, "input-tree", ""));
        if (!tf) { std::cerr << "ERROR: Unable to open input tree file: " << strOr(vm, "input-tree", "") << "\n"; return 1; }
        std::string nwk;
        std::getline(tf, nwk);
        for (size_t i = 0; i < nwk.size(); ++i) {
            if (nwk[i] != ':') continue;
            char* e = nullptr;
            const double v = std::strtod(nwk.c_str() + i + 1, &e);
            putLength(std::cout, v);
            std::cout << "\n";
        }
        return 0;
    }
    if (vm.count("dump-packed")) {
        // developer aid (no GPU): --dump-packed m|r compares the fast input path (records indexed in the mapped text and
        // packed straight into the flat arrays) with the general one (readSequences + per-sequence encoders)
        const bool aligned = vm["dump-packed"] == "m";
        long long sd = 1;
        try { if (vm.count("seed")) sd = std::stoll(vm["seed"]); } catch (...) {}
        PackedSequences fast;
        readSequencesPacked(strOr(vm, "input-file", ""), aligned, sd, fast);
        if (!fast.ok) { std::printf("SERIAL-ONLY\n"); return 0; }
        std::vector<std::string> seqs, names_;
        readSequences(strOr(vm, "input-file", ""), seqs, names_);
        if (seqs.empty()) { std::printf("0 records\n%s\n", fast.numSequences == 0 ? "IDENTICAL" : "DIFFERENT"); return fast.numSequences == 0 ? 0 : 2; }
        const std::vector<int> ids = shuffledIds(seqs.size(), sd);
        std::vector<std::string> names(seqs.size());
        for (size_t i = 0; i < seqs.size(); ++i) names[(size_t)ids[i]] = names_[i];
        bool same = fast.numSequences == seqs.size() && fast.names == names;
        if (aligned) {
            std::vector<uint64_t> flat; int seqLen = 0;
            packAligned(seqs, ids, flat, seqLen);
            same = same && seqLen == fast.seqLen && flat.size() == fast.flat.size() && std::equal(flat.begin(), flat.end(), fast.flat.begin());
            std::printf("%zu %d %zu\n", seqs.size(), seqLen, flat.size());
        } else {
            std::vector<uint64_t> flat, off, lens;
            packUnaligned(seqs, ids, flat, off, lens);
            same = same && flat.size() == fast.flat.size() && std::equal(flat.begin(), flat.end(), fast.flat.begin()) && off == fast.off && lens == fast.lens;
            std::printf("%zu %zu\n", seqs.size(), flat.size());
        }
        std::printf(same ? "IDENTICAL\n" : "DIFFERENT\n");
        return same ? 0 : 2;
    }
    if (vm.count("dump-fasta")) {
        // developer aid (no GPU): read --input-file and print name, length and FNV-1a of every record
        std::vector<std::string> seqs, names;
        readSequences(strOr(vm, "input-file", ""), seqs, names);
        std::printf("%zu\n", seqs.size());
        for (size_t i = 0; i < seqs.size(); ++i) {
            uint64_t h = 1469598103934665603ull;
            for (unsigned char ch : seqs[i]) { h ^= ch; h *= 1099511628211ull; }
            std::printf("%s %zu %016llx\n", names[i].c_str(), seqs[i].size(), (unsigned long long)h);
        }
        return 0;
    }
    for (const char* req : { "input-format", "input-file", "output-file" })
        if (!vm.count(req)) usageError(std::string("the option '--") + req + "' is required but missing");
    if (vm.count("add") && !vm.count("input-tree"))
        usageError("Backbone tree (--input-tree/-t) is required with --add option");

    Param params;
    params.kmerSize = stoiOr(vm, "kmer-size", 15);
    params.sketchSize = stoiOr(vm, "sketch-size", 1000);
    params.distanceType = stoiOr(vm, "distance-type", 1);  // code default is 1 although the help says JC
    params.in = strOr(vm, "input-format", "r");
    params.out = strOr(vm, "output-format", "t");
    const std::string algo = strOr(vm, "algorithm", "0");
    const std::string placemode = strOr(vm, "placement-mode", "1");  // the reference reads --algorithm here
    const bool add = vm.count("add") != 0;
    long long seed = 1;
    try { if (vm.count("seed")) seed = std::stoll(vm["seed"]); } catch (...) {}
    int device = (int)stoiOr(vm, "device", 0);
    const std::string inputFile = vm["input-file"], outputFile = vm["output-file"];
    // several GPUs: one rank process per GPU, started once the input is read and before the first GPU call (startRanks)
    RankOptions ranks;
    ranks.gpus = (int)stoiOr(vm, "gpus", 1);
    if (ranks.gpus < 1) usageError("--gpus must be at least 1");
    if (vm.count("devices")) {
        std::stringstream ss(vm["devices"]);
        std::string tok;
        while (std::getline(ss, tok, ',')) {
            try { ranks.devices.push_back(std::stoi(tok)); } catch (...) { usageError("--devices: a comma-separated list of GPU indices"); }
        }
        if (!vm.count("gpus") && !vm.count("world")) ranks.gpus = (int)ranks.devices.size();
    }
    {
        const std::string tr = strOr(vm, "transport", "auto");
        if (tr == "auto") ranks.transport = 0; else if (tr == "rccl") ranks.transport = 1; else if (tr == "ipc") ranks.transport = 2;
        else usageError("--transport: auto, rccl or ipc");
    }
    if (vm.count("world")) {
        ranks.ext_world = (int)stoiOr(vm, "world", 1);
        ranks.ext_rank = vm.count("rank") ? (int)stoiOr(vm, "rank", 0) : -1;
        ranks.rendezvous = strOr(vm, "rendezvous", "");
        ranks.gpus = 1;
        if (ranks.ext_world > 1 && (ranks.ext_rank < 0 || ranks.ext_rank >= ranks.ext_world || ranks.rendezvous.empty()))
            usageError("--world needs --rank (0 <= rank < world) and --rendezvous");
    }
    const bool multi = ranks.multi();
    if (!multi && ranks.devices.size() == 1) device = ranks.devices[0];      // (`--devices 3` alone: one rank on GPU 3)

    const int placement_thr = 30000, dc_thr = 1000000;  // src/tree_generation.cu:247-248
    auto ms_since = [](std::chrono::high_resolution_clock::time_point t0) {
        return (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    };
    auto open_out = [&]() {
        // (several ranks: every rank ends with the whole tree, rank 0 writes it)
        auto os = std::make_unique<std::ofstream>(rankInfo().rank == 0 ? outputFile.c_str() : "/dev/null");
        if (!*os) die("ERROR: cannot open output file: " + outputFile);
        return os;
    };
    // mode selection of the reference (src/tree_generation.cu:377,422,450): 1 = placement, 3 = DC, 2 = NJ
    auto pick_mode = [&](long long n) {
        if (algo == "1" || (algo == "0" && n >= placement_thr && n < dc_thr)) return 1;
        if (algo == "3" || (algo == "0" && n >= dc_thr)) return 3;
        return 2;
    };
    // exact placement mode: `-p 0`, or -- the reference's effective behaviour, which reads the placement mode
    // from --algorithm (src/tree_generation.cu:222-224) -- an explicit `-m 0` that lands in placement
    const bool exact_mode = placemode == "0" || (vm.count("algorithm") && algo == "0");

    if (add) {
        // src/tree_generation.cu:252-332
        std::ifstream treeFileStream(strOr(vm, "input-tree", ""));
        if (!treeFileStream) { std::cerr << "ERROR: Unable to open input tree file: " << strOr(vm, "input-tree", "") << "\n"; return 1; }
        if (!(params.out == "t" && (params.in == "r" || params.in == "m"))) {
            std::cerr << "Adding new sequnces only supported with input aligned and unaligned sequences\n";
            return 1;
        }
        std::string newickTree;
        std::getline(treeFileStream, newickTree);
        // The slot of a sequence is the index its name has in the backbone tree (Tree::Tree numbers the leaves in order of
        // appearance, src/tree.cpp:341), queries follow in input order (src/tree_generation.cu:271-282).  Round 4: the
        // records go through the packed reader of the other modes (indexed in the mapped text, packed in parallel straight
        // into the device interface's arrays, HIP start-up on a helper thread meanwhile) -- the serial parser + per-sequence
        // encoders cost 1.6 s of a 4.1 s command for 550 000 x 1 000 sites; FASTQ input still takes that route.
        std::unique_ptr<Tree> tp;
        size_t backboneSize = 0;
        auto slots_of = [&](const std::vector<std::string>& namesDump, std::vector<std::string>* names_out) -> std::vector<int> {
            std::cerr << "Read " << namesDump.size() << " sequences from input file.\n";
            if (cliLog()) std::cerr << "  records indexed at " << ms_since(inputStart) << " ms\n";
            if (namesDump.empty()) die("No sequences found in the input file.");
            tp.reset(new Tree(newickTree, namesDump.size()));
            const Tree& t = *tp;
            std::cerr << "Tree loaded successfully with " << t.nodes.size() << " nodes and root " << t.nodes[(size_t)t.root].name << ".\n";
            if (cliLog()) std::cerr << "  backbone tree parsed at " << ms_since(inputStart) << " ms\n";
            backboneSize = t.m_numLeaves;
            const size_t numSequences = namesDump.size();
            // name -> leaf index: an open-addressing table over the tree's own names (later duplicates win, as with a map), filled
            // serially, then looked up by all host threads -- a node-based map cost 140 ms for 500 000 tips + 550 000 records
            auto hash_of = [](const std::string& v) {
                uint64_t h = 1469598103934665603ull;
                for (unsigned char ch : v) { h ^= ch; h *= 1099511628211ull; }
                return h ^ (h >> 29);
            };
            size_t cap = 16;
            while (cap < 2 * t.nodes.size()) cap <<= 1;
            std::vector<int32_t> table(cap, -1);
            for (size_t k = 0; k < t.nodes.size(); ++k) {          // (the table holds positions in t.nodes)
                const Node& nd = t.nodes[k];
                if (!nd.children.empty()) continue;
                size_t h = (size_t)hash_of(nd.name) & (cap - 1);
                while (table[h] >= 0 && t.nodes[(size_t)table[h]].name != nd.name) h = (h + 1) & (cap - 1);
                table[h] = (int32_t)k;
            }
            std::vector<int> ids(numSequences);
            {
                const unsigned nt = hostThreads(32);
                std::vector<std::thread> pool;
                for (unsigned w = 0; w < nt; ++w)
                    pool.emplace_back([&, w] {
                        for (size_t i = numSequences * w / nt; i < numSequences * (w + 1) / nt; ++i) {
                            size_t h = (size_t)hash_of(namesDump[i]) & (cap - 1);
                            while (table[h] >= 0 && t.nodes[(size_t)table[h]].name != namesDump[i]) h = (h + 1) & (cap - 1);
                            ids[i] = table[h] >= 0 ? t.nodes[(size_t)table[h]].idx : -1;
                        }
                    });
                for (std::thread& th : pool) th.join();
            }
            size_t found = 0, next = backboneSize;
            if (names_out) names_out->assign(backboneSize, "");
            for (size_t i = 0; i < numSequences; ++i) {
                if (ids[i] < 0) { ids[i] = (int)next++; if (names_out) names_out->push_back(namesDump[i]); }
                else { ++found; if (names_out) (*names_out)[(size_t)ids[i]] = namesDump[i]; }
            }
            if (found != backboneSize || next != numSequences) die("ERROR: every backbone tip needs exactly one sequence in the input file");
            if (backboneSize >= numSequences) die("ERROR: no query sequences to add");
            if (cliLog()) std::cerr << "  records matched to the backbone's tips at " << ms_since(inputStart) << " ms\n";
            return ids;
        };
        if (access(inputFile.c_str(), R_OK) != 0) {
            std::fprintf(stderr, "ERROR: cant open file: %s\n", inputFile.c_str());  // src/tree_generation.cu:138-141
            return 1;
        }
        // one GPU: HIP start-up runs while the input is read; several: the ranks are started once it has been read
        std::unique_ptr<AsyncDeviceContext> adev;
        if (!multi) adev.reset(new AsyncDeviceContext(device));
        const std::function<std::vector<int>(const std::vector<std::string>&)> ids_fn =
            [&](const std::vector<std::string>& nd) { return slots_of(nd, nullptr); };
        PackedSequences packed;
        readSequencesPacked(inputFile, params.in == "m", -1, packed, nullptr, nullptr, &ids_fn);
        std::vector<std::string> seqs, names;
        std::vector<int> ids;
        if (!packed.ok) {                         // FASTQ: serial parser + the per-sequence encoders
            std::vector<std::string> namesDump;
            readSequences(inputFile, seqs, namesDump);
            ids = slots_of(namesDump, &names);
        } else {
            names = std::move(packed.names);
        }
        const Tree& t = *tp;
        const size_t numSequences = packed.ok ? packed.numSequences : seqs.size();
        if (multi) { startRanks(ranks, device); adev.reset(new AsyncDeviceContext(rankInfo().device)); }
        auto output_ = open_out();
        if (cliLog()) std::cerr << "  input read, backbone tree parsed at " << ms_since(inputStart) << " ms\n";
        DeviceContext& dev = adev->get();
        if (cliLog()) std::cerr << "  device ready at " << ms_since(inputStart) << " ms\n";
        KPlacementDeviceArrays kplacementDeviceArrays;
        if (params.in == "r") {
            MashDeviceArrays mashDeviceArrays;
            std::cerr << "Allocating Mash Device Arrays" << std::endl;
            if (packed.ok) mashDeviceArrays.allocateDeviceArrays(dev, packed);
            else mashDeviceArrays.allocateDeviceArrays(dev, seqs, ids);
            std::cerr << "Sketch Construction in Progress" << std::endl;
            mashDeviceArrays.sketchConstructionOnGpu(dev, params);
        } else {
            MSADeviceArrays msaDeviceArrays;
            if (packed.ok) msaDeviceArrays.allocateDeviceArrays(dev, packed);
            else msaDeviceArrays.allocateDeviceArrays(dev, seqs, ids);
        }
        if (cliLog()) std::cerr << "  sequences on the device (sketches built) at " << ms_since(inputStart) << " ms\n";
        kplacementDeviceArrays.allocateDeviceArrays(numSequences, (int)backboneSize);
        kplacementDeviceArrays.initializeDeviceArrays(t);
        if (cliLog()) std::cerr << "  backbone state built at " << ms_since(inputStart) << " ms\n";
        kplacementDeviceArrays.addQuery(dev, params);
        if (cliLog()) std::cerr << "  addQuery done at " << ms_since(inputStart) << " ms\n";
        kplacementDeviceArrays.printTree(names, *output_);
        if (cliLog()) std::cerr << "  tree written at " << ms_since(inputStart) << " ms\n";
        printRankSummary(dev.ctx);
        return 0;
    }

    if ((params.in == "m" || params.in == "r") && params.out == "d") {
        // -o d: distance matrix in PHYLIP format (lower-triangular, tab separated -- the layout the
        // reader of -i d consumes).  "coming soon" in the reference (src/tree_generation.cu:55-58).
        const bool aligned = params.in == "m";
        std::vector<std::string> seqs, names_, names;
        readSequences(inputFile, seqs, names_);
        const size_t numSequences = seqs.size();
        if (numSequences < 2) die("ERROR: need at least two sequences in " + inputFile);
        names.resize(numSequences);
        const std::vector<int> ids = shuffledIds(numSequences, -1);   // input order
        for (size_t i = 0; i < numSequences; ++i) names[(size_t)ids[i]] = names_[i];
        auto output_ = open_out();
        DeviceContext dev(device);
        dpr_set_nj_mode(0);   // plain matrix layout, no NJ structures
        NJDeviceArrays njDeviceArrays;
        if (aligned) {
            MSADeviceArrays msaDeviceArrays;
            msaDeviceArrays.allocateDeviceArrays(dev, seqs, ids);
        } else {
            MashDeviceArrays mashDeviceArrays;
            mashDeviceArrays.allocateDeviceArrays(dev, seqs, ids);
            mashDeviceArrays.sketchConstructionOnGpu(dev, params);
        }
        njDeviceArrays.getDismatrix(dev, (int)numSequences, params, nullptr);
        std::vector<double> row(numSequences);
        char buf[64];
        *output_ << numSequences << "\n";
        for (size_t i = 0; i < numSequences; ++i) {
            gpuCheck(dpr_get_matrix_row(dev.ctx, (int64_t)i, row.data()), "dpr_get_matrix_row");
            *output_ << names[i];
            for (size_t j = 0; j < i; ++j) {
                std::snprintf(buf, sizeof(buf), "\t%.9g", row[j]);
                *output_ << buf;
            }
            *output_ << "\n";
        }
        return 0;
    }

    if ((params.in == "m" || params.in == "r") && params.out == "t") {
        const bool aligned = params.in == "m";
        std::vector<std::string> names;
        // (checked BEFORE the device thread starts: on a host without a GPU that thread ends the process with its own message,
        //  and which of the two a caller saw for a missing file depended on the race)
        if (access(inputFile.c_str(), R_OK) != 0) {
            std::fprintf(stderr, "ERROR: cant open file: %s\n", inputFile.c_str());  // src/tree_generation.cu:138-141
            return 1;
        }
        // one GPU: HIP start-up runs while the input is read; several: the ranks are started once it has been read
        std::unique_ptr<AsyncDeviceContext> adev;
        if (!multi) adev.reset(new AsyncDeviceContext(device));
        // fast path: records indexed in the mapped text and packed straight into the device interface's flat arrays;
        // as soon as the number of records is known the device thread allocates the NJ matrices (when NJ is the mode)
        struct Hook { AsyncDeviceContext* adev; decltype(pick_mode)* pick; } hook{ adev.get(), &pick_mode };
        PackedSequences packed;
        readSequencesPacked(inputFile, aligned, seed, packed, [](size_t n, void* u) {
            Hook* h = static_cast<Hook*>(u);
            if (h->adev && n >= 3 && (*h->pick)((long long)n) == 2) h->adev->reserveNJ(n);
        }, &hook);
        const long long parsed_ms = ms_since(inputStart);      // the file is read and packed (host side of the input phase)
        std::vector<std::string> seqs;
        std::vector<int> ids;
        if (!packed.ok) {                         // FASTQ: serial parser + the per-sequence encoders
            std::vector<std::string> names_;
            readSequences(inputFile, seqs, names_);
            ids = shuffledIds(seqs.size(), seed);
            names.resize(seqs.size());
            for (size_t i = 0; i < seqs.size(); ++i) names[(size_t)ids[i]] = names_[i];
        } else {
            names = std::move(packed.names);
        }
        const size_t numSequences = packed.ok ? packed.numSequences : seqs.size();
        if (numSequences < 3) die("ERROR: need at least three sequences in " + inputFile);
        if (multi) { startRanks(ranks, device); adev.reset(new AsyncDeviceContext(rankInfo().device)); }
        auto output_ = open_out();
        DeviceContext& dev = adev->get();
        MSADeviceArrays msaDeviceArrays;
        MashDeviceArrays mashDeviceArrays;
        const int mode = pick_mode((long long)numSequences);
        if (packed.ok) {
            if (aligned) msaDeviceArrays.allocateDeviceArrays(dev, packed);
            else mashDeviceArrays.allocateDeviceArrays(dev, packed);
        } else {
            if (aligned) msaDeviceArrays.allocateDeviceArrays(dev, seqs, ids);
            else mashDeviceArrays.allocateDeviceArrays(dev, seqs, ids);
        }
        std::cerr << "Input in: " << ms_since(inputStart) << " ms\n";
        // (not a line of the reference: how much of the input phase was the HIP runtime coming up on the helper thread --
        //  the input phase ends when BOTH the parsed input and the device context are there)
        std::cerr << "Device ready in: " << (long long)adev->createMs() << " ms\n";
        std::cerr << "Parsed in: " << parsed_ms << " ms\n";      // (not a line of the reference either: the host side alone)
        auto createArrayStart = std::chrono::high_resolution_clock::now();
        if (!aligned) {
            // the reference sketches only in placement/DC mode (-i r + NJ reads unsketched
            // memory there); the intended behaviour is to sketch first
            std::cerr << "Allocated in: " << ms_since(createArrayStart) << " ms\n";
            auto t0 = std::chrono::high_resolution_clock::now();
            mashDeviceArrays.sketchConstructionOnGpu(dev, params);
            std::cerr << "Sketch Created in: " << ms_since(t0) << " ms\n";
        }
        if (mode == 3) {
            // src/tree_generation.cu:422-449 (aligned), :541-575 (unaligned): backbone = batch = N/20
            std::cerr << "Using divide-and-conquer mode\n";
            const size_t backboneSize = numSequences / 20;
            if (backboneSize < 3) die("ERROR: divide-and-conquer mode needs at least 60 sequences (backbone = N/20)");
            params.batchSize = params.backboneSize = backboneSize;
            KPlacementDeviceArraysDC kplacementDeviceArraysDC;
            kplacementDeviceArraysDC.allocateDeviceArraysDC(backboneSize, numSequences);
            if (aligned) std::cerr << "Allocated in: " << ms_since(createArrayStart) << " ms\n";
            auto t0 = std::chrono::high_resolution_clock::now();
            kplacementDeviceArraysDC.findTreeDC(dev, params);
            const long long tree_ms = ms_since(t0);
            kplacementDeviceArraysDC.printTreeDC(names, *output_);
            std::cerr << "Tree Created in: " << tree_ms << " ms\n";
        } else if (mode == 1) {
            std::cerr << "Using ";
            std::cerr << (exact_mode ? " exact placement mode\n" : "k-closest placement mode\n");   // src/tree_generation.cu:378-402
            KPlacementDeviceArrays kplacementDeviceArrays;
            kplacementDeviceArrays.exact = exact_mode;
            kplacementDeviceArrays.allocateDeviceArrays(numSequences);
            if (aligned) std::cerr << "Allocated in: " << ms_since(createArrayStart) << " ms\n";
            auto t0 = std::chrono::high_resolution_clock::now();
            kplacementDeviceArrays.findPlacementTree(dev, params);
            const long long tree_ms = ms_since(t0);
            kplacementDeviceArrays.printTree(names, *output_);
            std::cerr << "Tree Created in: " << tree_ms << " ms\n";
        } else {
            std::cerr << "Using conventional NJ\n";
            if (numSequences >= 40000)
                std::cerr << "Warning: forcing conventional NJ on large datasets might result in unexpected behavior\n";
            auto t0 = std::chrono::high_resolution_clock::now();
            NJDeviceArrays njDeviceArrays;
            njDeviceArrays.getDismatrix(dev, (int)numSequences, params, nullptr);
            if (cliLog()) std::cerr << "  getDismatrix call " << ms_since(t0) << " ms\n";
            njDeviceArrays.findNeighbourJoiningTree(dev, names, *output_);
            std::cerr << "Tree Created in: " << ms_since(t0) << " ms\n";
        }
        printRankSummary(dev.ctx);
        // the tree is written: close the output and leave without running the static destructors of the HIP runtime
        output_.reset();
        if (cliLog()) std::cerr << "Main in: " << ms_since(inputStart) << " ms\n";
        std::cerr.flush();
        std::fflush(nullptr);
        // ... but the device buffers and streams ARE released here, by the process itself: left to the kernel driver, the ~15 GB of
        // live allocations of a 30 000-tip run delay the runtime start-up of the NEXT process by 0.35-0.45 s in four runs out of
        // ten on some hosts (30 runs each, back to back: none with this call, 12 without; the call itself costs nothing
        // measurable -- profiles/r3/cli_exit_sweep.jsonl).
        dpr_destroy(dev.ctx); dev.ctx = nullptr;
        // (DPR_CLI_NORMAL_EXIT=1: return through main and the exit handlers -- a profiler that writes its output at exit,
        //  rocprofv3 for one, sees nothing of a process that leaves through _exit)
        if (std::getenv("DPR_CLI_NORMAL_EXIT")) return 0;
        _exit(0);
    } else if (params.in == "d" && params.out == "t") {
        MatrixReader matrixReader;
        matrixReader.read(inputFile);
        const int numSequences = matrixReader.numSequences;
        if (numSequences < 3) die("ERROR: need at least three taxa in " + inputFile);
        const int mode = pick_mode(numSequences);
        if (mode == 3) { std::cerr << "Divide-and-conquer mode not supported with input matrix\n"; return 1; }
        if (multi) startRanks(ranks, device);
        auto output_ = open_out();
        DeviceContext dev(multi ? rankInfo().device : device);
        if (mode == 1) {
            std::cerr << "Using " << (exact_mode ? " exact placement mode\n" : "k-closest placement mode\n");
            gpuCheck(dpr_set_matrix_lower(dev.ctx, matrixReader.lower.data(), numSequences), "dpr_set_matrix_lower");
            KPlacementDeviceArrays kplacementDeviceArrays;
            kplacementDeviceArrays.exact = exact_mode;
            kplacementDeviceArrays.allocateDeviceArrays((size_t)numSequences);
            kplacementDeviceArrays.findPlacementTree(dev, params);
            kplacementDeviceArrays.printTree(matrixReader.name, *output_);
        } else {
            std::cerr << "Using conventional NJ\n";
            if (numSequences >= 40000)
                std::cerr << "Warning: forcing conventional NJ on large datasets might result in unexpected behavior\n";
            NJDeviceArrays njDeviceArrays;
            njDeviceArrays.getDismatrix(dev, numSequences, params, &matrixReader);
            njDeviceArrays.findNeighbourJoiningTree(dev, matrixReader.name, *output_);
        }
        printRankSummary(dev.ctx);
    } else {
        std::printf("Invalid input-output combinations!!!!!\n");
        return 1;
    }
    return 0;
}
