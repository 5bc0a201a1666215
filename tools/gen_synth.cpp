// gen_synth -- seeded synthetic inputs for bench.py and the full-size tests (bench / test infrastructure, not product).
//
// Stand-in for `iqtree2 --alisim ... -t RANDOM{yh/N} -rlen lo mean hi seed 1` of the reference's experiment protocol
// (scripts/alisim.sh:14, scripts/experiment.sh:14; iqtree2 is not in the image and cannot be fetched): a random
// Yule-Harding tree with N tips, branch lengths exponential(mean) clipped to [lo, hi], a uniform ACGT root sequence of L
// sites, JC69 substitutions down the tree, optionally insertions / deletions (`--indel 0.03,0.09` of the same script:
// events per substitution, geometric lengths) for the unaligned (Mash) configurations.  It KNOWS the generating tree and
// writes it as Newick, which is what the reference's accuracy protocol compares against (scripts/nrf.sh:26: REF_TREE).
//
// Same model as tests/_util.synth_alignment / synth_reads (numpy, kept for the small seeded test cases), but native:
// 100 000 x 10 000 sites took ~45 s per rank in numpy; here one process generates every input once (all host threads)
// and the ranks of a multi-GPU bench map the files.
//
//   gen_synth --tips N --sites L [--seed S] [--mean-bl m --lo a --hi b] [--indel INS,DEL[,MEANLEN]] [--shuffle SEED]
//             [--model jc69|gtr+g+i] [--indel-gaps] [--gap-frac F[,MEANRUN]] [--threads T] [--fasta out.fa] [--packed4 out.u64] [--packed2 prefix] [--tree out.nwk]
//             [--order out.i32]
//
// Gaps in ALIGNED output (round 4; gap cells are code 4 in the 4-bit packing, src/fourBitCompressor.cpp:33-35, and '-' in the
// FASTA; without either switch every output byte is as before):
//   --indel-gaps       what the authors' indel model (scripts/alisim.sh:14: `--indel 0.03,0.09`) leaves in an alignment:
//                      deletions on branches at 0.09 per substitution, geometric lengths (`--indel`'s numbers), as runs of gap
//                      cells that every descendant inherits -- ~1e-4 of the cells at the protocol's branch lengths, shared by
//                      clades.  (Insertions, which would add a few per cent of mostly-gap columns, are not modelled.)
//   --gap-frac F[,RUN] a STRESS input, not the protocol: on top of that, per-tip runs of '-' (Poisson number, geometric length,
//                      mean RUN = 10) up to an expected fraction F of every tip -- missing data that is independent from tip to
//                      tip, as in assemblies with dropped amplicons.  It makes the not-a-base plane of the distance kernel
//                      (src/MSA.cu:103-156: `useful`, `match`) do work at any size; it also perturbs every distance by its own
//                      ~F, which the pruned NJ scan pays for (30 000 tips, F = 0.03: 80 x the units listed, NJ 5.7 s instead of
//                      0.48 s, still below the streaming loop's 6.3 s; profiles/r4/post3_cells_variants_30k.txt).
//
// Tip names are T<k+1> with k the tip's index in the generating tree's creation order (the true tree uses the same
// names).  Output order = creation order, or a seeded permutation of it (--shuffle; --order writes the permutation:
// order[pos] = k).  --packed4: [N][ceil(L/16)] little-endian uint64 in the encoding of fourBitCompressor
// (src/fourBitCompressor.cpp:5-41; aligned output only).  --packed2 prefix: prefix.flat / .off / .len = the three arrays
// of twoBitCompressor's layout (src/twoBitCompressor.cpp:5-41) that dpr_set_reads takes.
//
// Determinism: every random draw comes from a generator keyed by (seed, node id, purpose), so the output does not depend
// on the number of threads.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

struct Rng {      // splitmix64 seeding + xoshiro256**
    uint64_t s[4];
    static uint64_t mix(uint64_t& x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t a = 0, uint64_t b = 0)
    {
        uint64_t x = seed * 0xD1342543DE82EF95ull + a * 0x2545F4914F6CDD1Dull + b * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        for (auto& v : s) v = mix(x);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }       // [0, 1)
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }  // [0, n)
    double exponential(double mean) { return -mean * std::log(1.0 - uniform()); }
    uint64_t poisson(double lam)
    {
        uint64_t k = 0;
        while (lam > 0) {      // Knuth's product method in chunks (Poisson variables add)
            const double l = lam > 30.0 ? 30.0 : lam;
            lam -= l;
            const double lim = std::exp(-l);
            double p = uniform();
            while (p > lim) { ++k; p *= uniform(); }
        }
        return k;
    }
    uint64_t geometric(double p)      // 1, 2, ... with success probability p
    {
        if (p >= 1.0) return 1;
        const double u = 1.0 - uniform();
        return 1 + (uint64_t)std::floor(std::log(u) / std::log(1.0 - p));
    }
};

struct Args {
    int64_t tips = 0, sites = 0;
    uint64_t seed = 1;
    double mean_bl = 2e-5, lo = 2e-6, hi = 2e-4;
    bool indels = false;
    double ins = 0.03, del = 0.09, indel_mean = 2.0;
    double gap_frac = 0.0, gap_run = 10.0;
    bool indel_gaps = false;
    bool gtr = false;                  // --model gtr+g+i
    bool shuffle = false;
    uint64_t shuffle_seed = 0;
    int threads = 0;
    std::string fasta, packed4, packed2, tree, order;
};

[[noreturn]] void die(const std::string& m)
{
    std::fprintf(stderr, "gen_synth: %s\n", m.c_str());
    std::exit(1);
}

int host_threads()
{
    int c = (int)std::thread::hardware_concurrency();
    if (c < 1) c = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < c) c = a; }
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {      // a GPU box exposes 256 CPUs and grants 16
        char q[64]; long long period = 0;
        if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const long long lim = std::atoll(q) / period;
            if (lim >= 1 && lim < c) c = (int)lim;
        }
        std::fclose(f);
    }
    if (const char* e = std::getenv("DPR_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v < c) c = v; }
    return c;
}

struct Tree {
    std::vector<int32_t> parent, child0, child1, leaf_of_node;   // leaf_of_node: creation-order tip index or -1
    std::vector<int32_t> node_of_leaf;
    std::vector<double> bl;                                      // branch above the node
};

Tree yule(int64_t n, const Args& a)
{
    Tree t;
    t.parent.assign(1, -1); t.child0.assign(1, -1); t.child1.assign(1, -1);
    t.node_of_leaf.assign(1, 0);
    t.parent.reserve((size_t)(2 * n)); t.child0.reserve((size_t)(2 * n)); t.child1.reserve((size_t)(2 * n));
    Rng r(a.seed, 0xA11CE, 1);
    while ((int64_t)t.node_of_leaf.size() < n) {
        const size_t k = (size_t)r.below(t.node_of_leaf.size());
        const int32_t node = t.node_of_leaf[k];
        const int32_t c0 = (int32_t)t.parent.size(), c1 = c0 + 1;
        t.parent.push_back(node); t.parent.push_back(node);
        t.child0.push_back(-1); t.child0.push_back(-1); t.child1.push_back(-1); t.child1.push_back(-1);
        t.child0[(size_t)node] = c0; t.child1[(size_t)node] = c1;
        t.node_of_leaf[k] = c0;
        t.node_of_leaf.push_back(c1);
    }
    const size_t nn = t.parent.size();
    t.leaf_of_node.assign(nn, -1);
    for (size_t k = 0; k < t.node_of_leaf.size(); ++k) t.leaf_of_node[(size_t)t.node_of_leaf[k]] = (int32_t)k;
    t.bl.assign(nn, 0.0);
    for (size_t v = 1; v < nn; ++v) {
        Rng rb(a.seed, v, 2);
        double b = rb.exponential(a.mean_bl);
        t.bl[v] = b < a.lo ? a.lo : (b > a.hi ? a.hi : b);
    }
    return t;
}

using Seq = std::vector<uint8_t>;      // codes 0..3 = A C G T; 4 = gap (aligned output with --gap-frac)

// --model gtr+g+i: the substitution model of the authors' protocol (scripts/alisim.sh:14: -m "GTR+G+I"; alisim draws the
// parameters when none are given -- the values here are the explicit ones of the same script's commented variant, line 21:
// GTR{0.0132,0.105,0.0417,0.00745,1.0254}+F{0.3,0.2,0.2,0.3}+G4{0.5}+I{0.2}).  Simulated by uniformisation: events arrive at
// rate mu x (site rate) per site, an event at a site in state i moves it to j with probability Q_ij / mu (or leaves it);
// a branch of length bl carries bl expected substitutions per site, as under JC69.
struct GtrModel {
    double freq[4] = { 0.3, 0.2, 0.2, 0.3 };
    double P[4][4];                    // jump probabilities of an event (row: from), diagonal = stay
    double mu = 1.0;
    std::vector<double> cum;           // cumulative site rates (0 for invariant sites)
    double total = 0.0;
    void setup(uint64_t seed, size_t L)
    {
        const double ex[6] = { 0.013206908228919744, 0.10497798848628515, 0.04165255672197765, 0.007450050795800881, 1.0253979004402303, 1.0 };   // AC AG AT CG CT GT
        double S[4][4] = {};
        int e = 0;
        for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) { S[i][j] = S[j][i] = ex[e++]; }
        double Q[4][4], scale = 0.0;
        for (int i = 0; i < 4; ++i) {
            double row = 0.0;
            for (int j = 0; j < 4; ++j) if (j != i) { Q[i][j] = S[i][j] * freq[j]; row += Q[i][j]; }
            Q[i][i] = -row;
            scale += freq[i] * row;
        }
        mu = 0.0;
        for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) Q[i][j] /= scale; if (-Q[i][i] > mu) mu = -Q[i][i]; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) P[i][j] = i == j ? 1.0 + Q[i][i] / mu : Q[i][j] / mu;
        // +I{0.2}: a fifth of the sites never change; +G4{0.5}: the others fall into four equiprobable classes with the mean
        // rates of a gamma(0.5) distribution's quartiles; scaled so that the mean over all sites is 1
        const double pinv = 0.2, cat[4] = { 0.03338775, 0.25191592, 0.82026848, 2.89442785 };
        Rng r(seed, 0, 11);
        cum.resize(L);
        total = 0.0;
        for (size_t s2 = 0; s2 < L; ++s2) {
            const double u = r.uniform();
            const double rate = u < pinv ? 0.0 : cat[r.below(4)] / (1.0 - pinv);
            total += rate;
            cum[s2] = total;
        }
    }
    uint8_t root_base(Rng& r) const
    {
        const double u = r.uniform();
        return u < freq[0] ? 0 : (u < freq[0] + freq[1] ? 1 : (u < freq[0] + freq[1] + freq[2] ? 2 : 3));
    }
};
static GtrModel g_gtr;

void evolve(const Seq& src, Seq& dst, double bl, uint64_t node, const Args& a)
{
    dst = src;
    Rng r(a.seed, node, 3);
    if (a.gtr) {
        const uint64_t k = r.poisson(g_gtr.total * g_gtr.mu * bl);
        for (uint64_t i = 0; i < k; ++i) {
            const double u = r.uniform() * g_gtr.total;
            size_t p = (size_t)(std::upper_bound(g_gtr.cum.begin(), g_gtr.cum.end(), u) - g_gtr.cum.begin());
            if (p >= dst.size()) p = dst.size() - 1;
            const double v = r.uniform();
            if (dst[p] >= 4) continue;        // a deleted site stays deleted
            const double* row = g_gtr.P[dst[p]];
            double c = 0.0;
            for (uint8_t j = 0; j < 4; ++j) { c += row[j]; if (v < c || j == 3) { dst[p] = j; break; } }
        }
    } else {
        const uint64_t k = r.poisson((double)dst.size() * bl);
        for (uint64_t i = 0; i < k; ++i) {
            const size_t p = (size_t)r.below(dst.size());
            const uint8_t nb = (uint8_t)((dst[p] + 1 + r.below(3)) & 3);
            if (dst[p] < 4) dst[p] = nb;          // a deleted site stays deleted
        }
    }
    if (!a.indels && (a.indel_gaps || a.gap_frac > 0.0)) {
        // aligned output: a deletion on this branch is a run of gap cells that every descendant inherits (own generator:
        // the substitution stream above is the same with and without gaps)
        Rng rg(a.seed, node, 7);
        const double pg = 1.0 / (a.indel_mean > 1.0 ? a.indel_mean : 1.0);
        const uint64_t nd = rg.poisson((double)dst.size() * bl * a.del);
        for (uint64_t i = 0; i < nd; ++i) {
            const uint64_t m = rg.geometric(pg);
            const size_t p0 = (size_t)rg.below(dst.size());
            for (size_t j = p0; j < dst.size() && j < p0 + m; ++j) dst[j] = 4;
        }
    }
    if (!a.indels) return;
    const double pg = 1.0 / (a.indel_mean > 1.0 ? a.indel_mean : 1.0);
    const uint64_t nd = r.poisson((double)dst.size() * bl * a.del);
    for (uint64_t i = 0; i < nd; ++i) {
        const uint64_t m = r.geometric(pg);
        if (dst.size() > m + 32) {
            const size_t p0 = (size_t)r.below(dst.size() - m);
            dst.erase(dst.begin() + (ptrdiff_t)p0, dst.begin() + (ptrdiff_t)(p0 + m));
        }
    }
    const uint64_t ni = r.poisson((double)dst.size() * bl * a.ins);
    for (uint64_t i = 0; i < ni; ++i) {
        const uint64_t m = r.geometric(pg);
        const size_t p0 = (size_t)r.below(dst.size() + 1);
        Seq insv((size_t)m);
        for (auto& c : insv) c = (uint8_t)r.below(4);
        dst.insert(dst.begin() + (ptrdiff_t)p0, insv.begin(), insv.end());
    }
}

void pack4_row(const Seq& s, uint64_t* out, size_t W)
{
    for (size_t w = 0; w < W; ++w) {
        uint64_t v = 0;
        const size_t lo = w * 16, hi = std::min(lo + 16, s.size());
        for (size_t j = lo; j < hi; ++j) v |= (uint64_t)s[j] << (4 * (j - lo));
        out[w] = v;
    }
}

void pack2_row(const Seq& s, uint64_t* out)
{
    const size_t W = (s.size() + 31) / 32;
    for (size_t w = 0; w < W; ++w) {
        uint64_t v = 0;
        const size_t lo = w * 32, hi = std::min(lo + 32, s.size());
        for (size_t j = lo; j < hi; ++j) v |= (uint64_t)s[j] << (2 * (j - lo));
        out[w] = v;
    }
}

struct Mapped {
    void* p = nullptr;
    size_t bytes = 0;
    int fd = -1;
    void open_rw(const std::string& path, size_t n)
    {
        fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) die("cannot create " + path);
        bytes = n ? n : 1;
        if (ftruncate(fd, (off_t)bytes) != 0) die("cannot size " + path);
        p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (p == MAP_FAILED) die("cannot map " + path);
    }
    void close_it(size_t final_bytes)
    {
        if (p) munmap(p, bytes);
        if (fd >= 0) { if (final_bytes != bytes) { if (ftruncate(fd, (off_t)final_bytes) != 0) die("ftruncate"); } ::close(fd); }
        p = nullptr; fd = -1;
    }
};

void write_file(const std::string& path, const void* data, size_t bytes)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) die("cannot create " + path);
    if (bytes && std::fwrite(data, 1, bytes, f) != bytes) die("short write to " + path);
    std::fclose(f);
}

void write_tree(const Tree& t, const std::string& path)
{
    std::string out;
    out.reserve(t.parent.size() * 24);
    char buf[64];
    // iterative pre-order with explicit closing tokens
    struct Item { int32_t node; int stage; };
    std::vector<Item> st;
    st.push_back({ 0, 0 });
    while (!st.empty()) {
        Item& it = st.back();
        const int32_t v = it.node;
        if (t.child0[(size_t)v] < 0) {
            std::snprintf(buf, sizeof buf, "T%d:%.9g", t.leaf_of_node[(size_t)v] + 1, t.bl[(size_t)v]);
            out += buf;
            st.pop_back();
            continue;
        }
        if (it.stage == 0) { out += '('; it.stage = 1; st.push_back({ t.child0[(size_t)v], 0 }); }
        else if (it.stage == 1) { out += ','; it.stage = 2; st.push_back({ t.child1[(size_t)v], 0 }); }
        else {
            out += ')';
            if (v != 0) { std::snprintf(buf, sizeof buf, ":%.9g", t.bl[(size_t)v]); out += buf; }
            st.pop_back();
        }
    }
    out += ";\n";
    write_file(path, out.data(), out.size());
}

}  // namespace

int main(int argc, char** argv)
{
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) die("missing value for " + k); return argv[++i]; };
        if (k == "--tips") a.tips = std::atoll(val());
        else if (k == "--sites") a.sites = std::atoll(val());
        else if (k == "--seed") a.seed = std::strtoull(val(), nullptr, 10);
        else if (k == "--mean-bl") a.mean_bl = std::atof(val());
        else if (k == "--lo") a.lo = std::atof(val());
        else if (k == "--hi") a.hi = std::atof(val());
        else if (k == "--indel") {
            a.indels = true;
            const std::string v = val();
            double x[3] = { 0.03, 0.09, 2.0 };
            int n = 0; size_t p = 0;
            while (n < 3 && p <= v.size()) { const size_t q = v.find(',', p); x[n++] = std::atof(v.substr(p, q - p).c_str()); if (q == std::string::npos) break; p = q + 1; }
            a.ins = x[0]; a.del = x[1]; a.indel_mean = x[2];
        }
        else if (k == "--gap-frac") {
            const std::string v = val();
            a.gap_frac = std::atof(v.c_str());
            const size_t q = v.find(',');
            if (q != std::string::npos) a.gap_run = std::atof(v.substr(q + 1).c_str());
            if (!(a.gap_frac >= 0.0 && a.gap_frac < 0.9) || !(a.gap_run >= 1.0)) die("--gap-frac F[,MEANRUN]: 0 <= F < 0.9, MEANRUN >= 1");
        }
        else if (k == "--indel-gaps") a.indel_gaps = true;
        else if (k == "--model") {
            std::string v = val();
            for (auto& c : v) c = (char)std::tolower((unsigned char)c);
            if (v == "gtr+g+i") a.gtr = true;
            else if (v != "jc" && v != "jc69") die("--model jc69 | gtr+g+i");
        }
        else if (k == "--shuffle") { a.shuffle = true; a.shuffle_seed = std::strtoull(val(), nullptr, 10); }
        else if (k == "--threads") a.threads = std::atoi(val());
        else if (k == "--fasta") a.fasta = val();
        else if (k == "--packed4") a.packed4 = val();
        else if (k == "--packed2") a.packed2 = val();
        else if (k == "--tree") a.tree = val();
        else if (k == "--order") a.order = val();
        else if (k == "-h" || k == "--help") {
            std::fprintf(stderr, "usage: gen_synth --tips N --sites L [--seed S] [--mean-bl m --lo a --hi b] [--indel INS,DEL[,MEANLEN]] [--shuffle SEED]\n"
                                 "                 [--model jc69|gtr+g+i] [--indel-gaps] [--gap-frac F[,MEANRUN]] [--threads T] [--fasta f] [--packed4 f] [--packed2 prefix] [--tree f] [--order f]\n");
            return 0;
        } else die("unknown argument " + k);
    }
    if (a.tips < 2 || a.sites < 1) die("--tips >= 2 and --sites >= 1 are required");
    if (a.indels && !a.packed4.empty()) die("--packed4 needs aligned output (no --indel)");
    if (a.indels && (a.gap_frac > 0.0 || a.indel_gaps)) die("--gap-frac / --indel-gaps are for aligned output (no --indel)");
    if (a.indels && a.gtr) die("--model gtr+g+i is for aligned output (per-site rates; no --indel)");
    if (a.gtr) g_gtr.setup(a.seed, (size_t)a.sites);
    const int64_t N = a.tips, L = a.sites;
    const int T = a.threads > 0 ? a.threads : host_threads();

    const Tree t = yule(N, a);
    const size_t nn = t.parent.size();

    // output position of every tip
    std::vector<int32_t> order((size_t)N), pos((size_t)N);
    for (int64_t k = 0; k < N; ++k) order[(size_t)k] = (int32_t)k;
    if (a.shuffle) {
        Rng r(a.shuffle_seed, 0x5F0FF1E, 4);
        for (int64_t k = N - 1; k > 0; --k) std::swap(order[(size_t)k], order[(size_t)r.below((uint64_t)k + 1)]);
    }
    for (int64_t p = 0; p < N; ++p) pos[(size_t)order[(size_t)p]] = (int32_t)p;

    // subtree tip counts (children have larger ids than their parents)
    std::vector<int32_t> cnt(nn, 0);
    for (size_t v = nn; v-- > 0;) {
        if (t.child0[v] < 0) cnt[v] = 1;
        if (v > 0) cnt[(size_t)t.parent[v]] += cnt[v];
    }
    // cut: subtrees of at most N / (8 T) tips are the parallel tasks; everything above is evolved serially first
    const int32_t cap = (int32_t)std::max<int64_t>(1, N / (8 * (int64_t)T));
    std::vector<int32_t> tasks;
    std::vector<Seq> task_seq;
    {
        struct Fr { int32_t node; Seq seq; };
        std::vector<Fr> st;
        Seq root((size_t)L);
        Rng r0(a.seed, 0, 5);
        for (auto& c : root) c = a.gtr ? g_gtr.root_base(r0) : (uint8_t)r0.below(4);
        st.push_back({ 0, std::move(root) });
        while (!st.empty()) {
            Fr f = std::move(st.back());
            st.pop_back();
            if (cnt[(size_t)f.node] <= cap || t.child0[(size_t)f.node] < 0) { tasks.push_back(f.node); task_seq.push_back(std::move(f.seq)); continue; }
            for (int32_t c : { t.child0[(size_t)f.node], t.child1[(size_t)f.node] }) {
                Fr g; g.node = c;
                evolve(f.seq, g.seq, t.bl[(size_t)c], (uint64_t)c, a);
                st.push_back(std::move(g));
            }
        }
    }

    // outputs
    const bool aligned = !a.indels;
    const size_t W4 = (size_t)((L + 15) / 16);
    Mapped m4, mfa;
    std::vector<size_t> fa_off;
    std::vector<Seq> keep;                       // unaligned: tips by output position
    if (!a.packed4.empty()) m4.open_rw(a.packed4, (size_t)N * W4 * 8);
    if (!aligned) keep.resize((size_t)N);
    size_t fa_bytes = 0;
    auto header = [&](int32_t tip, char* buf) { return (size_t)std::snprintf(buf, 48, ">T%d some comment\n", tip + 1); };
    if (!a.fasta.empty() && aligned) {
        fa_off.resize((size_t)N + 1);
        char hb[48];
        size_t off = 0;
        for (int64_t p = 0; p < N; ++p) { fa_off[(size_t)p] = off; off += header(order[(size_t)p], hb) + (size_t)L + 1; }
        fa_off[(size_t)N] = off;
        fa_bytes = off;
        mfa.open_rw(a.fasta, fa_bytes);
    }
    static const char kBase[5] = { 'A', 'C', 'G', 'T', '-' };
    auto emit = [&](int32_t tip, Seq& s) {
        const size_t p = (size_t)pos[(size_t)tip];
        if (aligned && a.gap_frac > 0.0) {       // this tip's own runs of missing cells
            Rng rg(a.seed, (uint64_t)tip, 8);
            const uint64_t nr = rg.poisson(a.gap_frac * (double)s.size() / a.gap_run);
            for (uint64_t i = 0; i < nr; ++i) {
                const uint64_t m = rg.geometric(1.0 / a.gap_run);
                const size_t p0 = (size_t)rg.below(s.size());
                for (size_t j = p0; j < s.size() && j < p0 + m; ++j) s[j] = 4;
            }
        }
        if (m4.p) pack4_row(s, (uint64_t*)m4.p + p * W4, W4);
        if (mfa.p) {
            char* o = (char*)mfa.p + fa_off[p];
            o += header(tip, o);
            for (size_t j = 0; j < s.size(); ++j) o[j] = kBase[s[j]];
            o[s.size()] = '\n';
        }
        if (!aligned) keep[p] = std::move(s);
    };
    std::atomic<size_t> next_task{ 0 };
    // largest subtrees first (the task list is in DFS order; sort indices by size)
    std::vector<size_t> tord(tasks.size());
    for (size_t i = 0; i < tord.size(); ++i) tord[i] = i;
    std::sort(tord.begin(), tord.end(), [&](size_t x, size_t y) { return cnt[(size_t)tasks[x]] > cnt[(size_t)tasks[y]]; });
    auto worker = [&]() {
        struct Fr { int32_t node; Seq seq; };
        std::vector<Fr> st;
        for (;;) {
            const size_t ti = next_task.fetch_add(1);
            if (ti >= tord.size()) break;
            const size_t tk = tord[ti];
            st.push_back({ tasks[tk], std::move(task_seq[tk]) });
            while (!st.empty()) {
                Fr f = std::move(st.back());
                st.pop_back();
                if (t.child0[(size_t)f.node] < 0) { emit(t.leaf_of_node[(size_t)f.node], f.seq); continue; }
                for (int32_t c : { t.child0[(size_t)f.node], t.child1[(size_t)f.node] }) {
                    Fr g; g.node = c;
                    evolve(f.seq, g.seq, t.bl[(size_t)c], (uint64_t)c, a);
                    st.push_back(std::move(g));
                }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < T; ++i) th.emplace_back(worker);
        worker();
        for (auto& x : th) x.join();
    }
    if (m4.p) m4.close_it((size_t)N * W4 * 8);
    if (mfa.p) mfa.close_it(fa_bytes);

    if (!aligned) {
        if (!a.fasta.empty()) {
            FILE* f = std::fopen(a.fasta.c_str(), "wb");
            if (!f) die("cannot create " + a.fasta);
            std::string line;
            char hb[48];
            for (int64_t p = 0; p < N; ++p) {
                const Seq& s = keep[(size_t)p];
                const size_t h = header(order[(size_t)p], hb);
                line.assign(hb, h);
                line.resize(h + s.size() + 1);
                for (size_t j = 0; j < s.size(); ++j) line[h + j] = kBase[s[j]];
                line[h + s.size()] = '\n';
                if (std::fwrite(line.data(), 1, line.size(), f) != line.size()) die("short write to " + a.fasta);
            }
            std::fclose(f);
        }
    }
    if (!a.packed2.empty()) {
        if (aligned) die("--packed2 is written for unaligned output (--indel); aligned tips use --packed4");
        std::vector<uint64_t> off((size_t)N), len((size_t)N);
        uint64_t tot = 0;
        for (int64_t p = 0; p < N; ++p) { off[(size_t)p] = tot; len[(size_t)p] = keep[(size_t)p].size(); tot += (keep[(size_t)p].size() + 31) / 32; }
        std::vector<uint64_t> flat((size_t)(tot ? tot : 1), 0ull);
        std::atomic<int64_t> nx{ 0 };
        auto pk = [&]() { for (;;) { const int64_t p = nx.fetch_add(1); if (p >= N) break; pack2_row(keep[(size_t)p], flat.data() + off[(size_t)p]); } };
        std::vector<std::thread> th;
        for (int i = 1; i < T; ++i) th.emplace_back(pk);
        pk();
        for (auto& x : th) x.join();
        write_file(a.packed2 + ".flat", flat.data(), flat.size() * 8);
        write_file(a.packed2 + ".off", off.data(), off.size() * 8);
        write_file(a.packed2 + ".len", len.data(), len.size() * 8);
    }
    if (!a.tree.empty()) write_tree(t, a.tree);
    if (!a.order.empty()) write_file(a.order, order.data(), order.size() * 4);
    return 0;
}
