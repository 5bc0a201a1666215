// dipper -- command line of the MI355X-native distance-based phylogeny engine.
// Keeps the flags, formats, banners and exit codes of the reference's main()
// (src/tree_generation.cu:33-99,159-646); host orchestration is plain C++ over the C ABI.
#include "dipper_host.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>

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
    "  --device arg                GPU index (default 0; the reference hard-codes 1)\n";

struct Opt { const char* lng; char sht; bool has_arg; };
static const Opt kOpts[] = {
    { "input-format", 'i', true }, { "input-file", 'I', true }, { "output-file", 'O', true },
    { "output-format", 'o', true }, { "algorithm", 'm', true }, { "placement-mode", 'p', true },
    { "kmer-size", 'k', true }, { "sketch-size", 's', true }, { "distance-type", 'd', true },
    { "add", 'a', false }, { "input-tree", 't', true }, { "help", 'h', false },
    { "seed", 0, true }, { "device", 0, true },
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
    auto inputStart = std::chrono::high_resolution_clock::now();
    auto vm = parseArguments(argc, argv);
    if (vm.count("help")) { std::cerr << kHelp << std::endl; return 0; }
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
    const int device = (int)stoiOr(vm, "device", 0);
    const std::string inputFile = vm["input-file"], outputFile = vm["output-file"];
    (void)placemode;

    const int placement_thr = 30000, dc_thr = 1000000;  // src/tree_generation.cu:247-248

    if (add) die("--add needs the k-closest placement path, which is not built yet in this round");

    if (params.in == "m" && params.out == "t") {
        std::vector<std::string> seqs, names_, names;
        readSequences(inputFile, seqs, names_);
        const size_t numSequences = seqs.size();
        if (numSequences < 2) die("ERROR: need at least two sequences in " + inputFile);
        names.resize(numSequences);
        const std::vector<int> ids = shuffledIds(numSequences, seed);
        for (size_t i = 0; i < numSequences; ++i) names[(size_t)ids[i]] = names_[i];
        std::ofstream output_(outputFile.c_str());
        if (!output_) die("ERROR: cannot open output file: " + outputFile);
        DeviceContext dev(device);
        MSADeviceArrays msaDeviceArrays;
        msaDeviceArrays.allocateDeviceArrays(dev, seqs, ids);
        auto inputEnd = std::chrono::high_resolution_clock::now();
        std::cerr << "Input in: " << std::chrono::duration_cast<std::chrono::milliseconds>(inputEnd - inputStart).count() << " ms\n";
        const bool wantPlacement = algo == "1" || (algo == "0" && (int)numSequences >= placement_thr && (int)numSequences < dc_thr);
        const bool wantDC = algo == "3" || (algo == "0" && (int)numSequences >= dc_thr);
        if (wantPlacement) die("k-closest placement mode is not built yet in this round (use -m 2 for conventional NJ)");
        if (wantDC) die("divide-and-conquer mode is not built yet in this round (use -m 2 for conventional NJ)");
        std::cerr << "Using conventional NJ\n";
        if (numSequences >= 40000)
            std::cerr << "Warning: forcing conventional NJ on large datasets might result in unexpected behavior\n";
        auto t0 = std::chrono::high_resolution_clock::now();
        NJDeviceArrays njDeviceArrays;
        njDeviceArrays.getDismatrix(dev, (int)numSequences, params, nullptr);
        njDeviceArrays.findNeighbourJoiningTree(dev, names, output_);
        auto t1 = std::chrono::high_resolution_clock::now();
        std::cerr << "Tree Created in: " << std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() << " ms\n";
    } else if (params.in == "r" && params.out == "t") {
        die("unaligned input (-i r) needs the Mash path, which is not built yet in this round");
    } else if (params.in == "d" && params.out == "t") {
        MatrixReader matrixReader;
        matrixReader.read(inputFile);
        const int numSequences = matrixReader.numSequences;
        std::ofstream output_(outputFile.c_str());
        if (!output_) die("ERROR: cannot open output file: " + outputFile);
        const bool wantPlacement = algo == "1" || (algo == "0" && numSequences >= placement_thr && numSequences < dc_thr);
        const bool wantDC = algo == "3" || (algo == "0" && numSequences >= dc_thr);
        if (wantDC) { std::cerr << "Divide-and-conquer mode not supported with input matrix\n"; return 1; }
        if (wantPlacement) die("k-closest placement mode is not built yet in this round (use -m 2 for conventional NJ)");
        std::cerr << "Using conventional NJ\n";
        if (numSequences >= 40000)
            std::cerr << "Warning: forcing conventional NJ on large datasets might result in unexpected behavior\n";
        DeviceContext dev(device);
        NJDeviceArrays njDeviceArrays;
        njDeviceArrays.getDismatrix(dev, numSequences, params, &matrixReader);
        njDeviceArrays.findNeighbourJoiningTree(dev, matrixReader.name, output_);
    } else {
        std::printf("Invalid input-output combinations!!!!!\n");
        return 1;
    }
    return 0;
}
