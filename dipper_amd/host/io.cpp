// Readers and writers of the dipper host side.
#include "dipper_host.hpp"

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <signal.h>
#include <cerrno>
#include <chrono>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <thread>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>

namespace dipper {

RankInfo& rankInfo()
{
    static RankInfo info;
    return info;
}
// a rank that gives up tells the others first: they leave their collectives with an error instead of waiting for it
static void rankFailed()
{
    if (rankInfo().region) dpr_shared_abort(rankInfo().region);
}

void die(const std::string& msg)
{
    std::fprintf(stderr, "%s\n", msg.c_str());      // (not std::cerr: ranks > 0 have their progress lines switched off)
    rankFailed();
    std::exit(1);
}

void gpuCheck(int rc, const char* what)
{
    if (rc < 0) {
        if (rankInfo().world > 1) std::fprintf(stderr, "Gpu_ERROR (rank %d of %d): %s failed: %s\n", rankInfo().rank, rankInfo().world, what, dpr_last_error());
        else std::fprintf(stderr, "Gpu_ERROR: %s failed: %s\n", what, dpr_last_error());
        rankFailed();
        std::exit(1);
    }
}

void startRanks(const RankOptions& o, int base_device)
{
    RankInfo& me = rankInfo();
    auto device_of = [&](int r) { return r < (int)o.devices.size() ? o.devices[(size_t)r] : base_device + r; };
    if (o.ext_world > 1) {
        // ranks started from outside: the region is a POSIX shared memory object every rank opens (the first one creates it;
        // a fresh object is zero-filled, which is the state dpr_comm_init_shared expects)
        if (o.ext_rank < 0 || o.ext_rank >= o.ext_world || o.rendezvous.empty()) die("ERROR: --world needs --rank (0 <= rank < world) and --rendezvous NAME");
        const std::string name = (o.rendezvous[0] == '/' ? "" : "/") + o.rendezvous;
        const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, DPR_COMM_SHARED_BYTES) != 0) die("ERROR: cannot open the rendezvous shared memory object " + name);
        void* p = mmap(nullptr, DPR_COMM_SHARED_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) die("ERROR: cannot map the rendezvous shared memory object " + name);
        me.rank = o.ext_rank; me.world = o.ext_world; me.region = p; me.device = device_of(o.ext_rank); me.transport = o.transport;
        if (me.rank > 0) std::cerr.setstate(std::ios_base::badbit);
        return;
    }
    if (o.gpus <= 1) return;
    const int G = o.gpus;
    if (G > 64) die("ERROR: --gpus: at most 64 ranks");
    void* region = mmap(nullptr, DPR_COMM_SHARED_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (region == MAP_FAILED) die("ERROR: cannot allocate the ranks' shared region");
    std::cerr << "Starting " << G << " ranks (devices";
    for (int r = 0; r < G; ++r) std::cerr << " " << device_of(r);
    std::cerr << ")\n";
    std::cerr.flush();
    std::fflush(nullptr);
    std::vector<pid_t> pids((size_t)G, -1);
    for (int r = 0; r < G; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            dpr_shared_abort(region);
            for (int k = 0; k < r; ++k) kill(pids[(size_t)k], SIGTERM);
            die("ERROR: fork failed");
        }
        if (pid == 0) {
            me.rank = r; me.world = G; me.region = region; me.device = device_of(r); me.transport = o.transport;
            if (r > 0) std::cerr.setstate(std::ios_base::badbit);      // one copy of the progress lines: rank 0's
            return;
        }
        pids[(size_t)r] = pid;
    }
    // the launcher: no GPU call here, ever.  First failure -> failure word -> the others get a few seconds to leave by themselves.
    int left = G, bad = 0;
    double deadline = -1.0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (left > 0) {
        int status = 0;
        const pid_t pid = waitpid(-1, &status, deadline < 0 ? 0 : WNOHANG);
        if (pid > 0) {
            int r = -1;
            for (int k = 0; k < G; ++k) if (pids[(size_t)k] == pid) r = k;
            if (r < 0) continue;
            pids[(size_t)r] = -1;
            --left;
            const bool ok = WIFEXITED(status) && WEXITSTATUS(status) == 0;
            if (!ok) {
                if (!bad) {
                    if (WIFSIGNALED(status)) std::fprintf(stderr, "ERROR: rank %d was ended by signal %d\n", r, WTERMSIG(status));
                    else std::fprintf(stderr, "ERROR: rank %d failed (exit code %d)\n", r, WIFEXITED(status) ? WEXITSTATUS(status) : -1);
                    dpr_shared_abort(region);
                    deadline = now() + 10.0;
                }
                ++bad;
            }
            continue;
        }
        if (pid < 0 && errno != EINTR) break;
        if (deadline >= 0) {
            if (now() > deadline) {
                for (int k = 0; k < G; ++k) if (pids[(size_t)k] > 0) kill(pids[(size_t)k], SIGKILL);      // (exact pids: our own children)
                deadline = now() + 1e9;
            }
            usleep(2000);
        }
    }
    std::fflush(nullptr);
    _exit(bad ? 1 : 0);
}

unsigned hostThreads(unsigned cap)
{
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const unsigned a = (unsigned)CPU_COUNT(&set);
        if (a > 0 && a < n) n = a;
    }
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {      // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64] = { 0 };
        long long period = 0;
        if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const long long quota = std::atoll(q);
            const unsigned c = (unsigned)std::max(1LL, quota / period);
            if (c < n) n = c;
        }
        std::fclose(f);
    }
    if (const char* e = std::getenv("DPR_HOST_THREADS")) {           // experiments: fewer reader threads
        const unsigned v = (unsigned)std::atoi(e);
        if (v >= 1 && v < n) n = v;
    }
    return std::max(1u, std::min(cap, n));
}

static std::string slurp_gz(const std::string& path)
{
    gzFile f = gzopen(path.c_str(), "r");
    if (!f) {
        std::fprintf(stderr, "ERROR: cant open file: %s\n", path.c_str());  // src/tree_generation.cu:138-141
        std::exit(1);
    }
    gzbuffer(f, 1 << 20);
    std::string data;
    std::vector<char> buf(1 << 22);
    int got;
    while ((got = gzread(f, buf.data(), (unsigned)buf.size())) > 0) data.append(buf.data(), (size_t)got);
    // A gzip stream that ends early (a cut download, a full disk) inflates to a prefix of the records: the reference's kseq
    // reader silently builds a tree of that prefix.  Here it is an error -- a tree of 29 of 700 sequences is not what was asked.
    int zerr = Z_OK;
    (void)gzerror(f, &zerr);
    const int cerr = gzclose(f);
    if (got < 0 || (zerr != Z_OK && zerr != Z_STREAM_END) || cerr != Z_OK) {
        std::fprintf(stderr, "ERROR: truncated or corrupt gzip input: %s\n", path.c_str());
        std::exit(1);
    }
    return data;
}

// serial record parser with klib-kseq semantics (FASTA and FASTQ)
static void parseRecords(const char* data, size_t n, std::vector<std::string>& seqs, std::vector<std::string>& names)
{
    size_t p = 0;
    // jump to the first header
    while (p < n && data[p] != '>' && data[p] != '@') ++p;
    while (p < n) {
        ++p;  // header char
        size_t e = p;
        while (e < n && !std::isspace((unsigned char)data[e])) ++e;
        names.emplace_back(data + p, e - p);
        // rest of the header line is the comment
        while (e < n && data[e] != '\n') ++e;
        p = e < n ? e + 1 : n;
        std::string seq;
        bool fastq = false;
        while (p < n) {
            const char c = data[p];
            if (c == '>' || c == '@') break;
            if (c == '+') { fastq = true; break; }
            if (c == '\n') { ++p; continue; }
            size_t le = p;
            while (le < n && data[le] != '\n') ++le;
            seq.append(data + p, le - p);
            if (seq.size() > 1 && seq.back() == '\r') seq.pop_back();
            p = le < n ? le + 1 : n;
        }
        if (fastq) {  // skip '+' line and the quality string (same length as the sequence)
            while (p < n && data[p] != '\n') ++p;
            if (p < n) ++p;
            size_t q = 0;
            while (p < n && q < seq.size()) {
                size_t le = p;
                while (le < n && data[le] != '\n') ++le;
                size_t len = le - p;
                if (len > 1 && data[le - 1] == '\r') --len;
                q += len;
                p = le < n ? le + 1 : n;
            }
            while (p < n && data[p] != '>' && data[p] != '@') ++p;
        }
        seqs.push_back(std::move(seq));
    }
}

// Plain FASTA: the records are found and copied out by all host threads (same result as the serial
// parser).  Returns false when the text needs the serial parser (FASTQ: a line starting with '+').
static bool parseFastaParallel(const char* data, size_t n, std::vector<std::string>& seqs, std::vector<std::string>& names)
{
    const unsigned nt = hostThreads(64);
    if (n < (size_t)(1 << 22) || nt < 2) return false;
    size_t first = 0;
    while (first < n && data[first] != '>' && data[first] != '@') ++first;
    if (first >= n) return false;
    // record starts: a '>' or '@' at the beginning of a line (the serial parser's rule) after the first header
    std::vector<std::vector<size_t>> starts(nt);
    std::vector<char> plus(nt, 0);
    const size_t chunk = (n - first + nt - 1) / nt;
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t] {
                const size_t lo = first + (size_t)t * chunk, hi = std::min(n, lo + chunk);
                if (lo >= hi) return;
                if (t == 0) starts[0].push_back(first);
                const char* q = data + lo;
                const char* end = data + hi;
                while (q < end) {
                    q = static_cast<const char*>(std::memchr(q, '\n', (size_t)(end - q)));
                    if (!q) break;
                    ++q;
                    if (q < data + n) {
                        if (*q == '>' || *q == '@') starts[t].push_back((size_t)(q - data));
                        else if (*q == '+') plus[t] = 1;
                    }
                }
            });
        for (auto& th : pool) th.join();
    }
    for (char c : plus) if (c) return false;
    std::vector<size_t> st;
    for (auto& v : starts) st.insert(st.end(), v.begin(), v.end());
    const size_t nrec = st.size();
    st.push_back(n);
    const size_t base = seqs.size();
    seqs.resize(base + nrec);
    names.resize(base + nrec);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&, t] {
            for (size_t r = t; r < nrec; r += nt) {
                size_t p = st[r] + 1;
                const size_t end = st[r + 1];
                size_t e = p;
                while (e < end && !std::isspace((unsigned char)data[e])) ++e;
                names[base + r].assign(data + p, e - p);
                while (e < end && data[e] != '\n') ++e;
                p = e < end ? e + 1 : end;
                std::string& seq = seqs[base + r];
                seq.reserve(end - p);
                while (p < end) {
                    if (data[p] == '\n') { ++p; continue; }
                    const char* nl = static_cast<const char*>(std::memchr(data + p, '\n', end - p));
                    const size_t le = nl ? (size_t)(nl - data) : end;
                    seq.append(data + p, le - p);
                    if (seq.size() > 1 && seq.back() == '\r') seq.pop_back();
                    p = le < end ? le + 1 : end;
                }
            }
        });
    for (auto& th : pool) th.join();
    return true;
}

void readSequences(const std::string& path, std::vector<std::string>& seqs, std::vector<std::string>& names)
{
    // uncompressed input: map the file (no copy); gzip (magic 1f 8b): inflate into memory
    int fd = open(path.c_str(), O_RDONLY);
    if (fd >= 0) {
        struct stat sb;
        unsigned char magic[2] = { 0, 0 };
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 2 && pread(fd, magic, 2, 0) == 2 &&
            !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (m != MAP_FAILED) {
                const char* data = static_cast<const char*>(m);
                if (!parseFastaParallel(data, (size_t)sb.st_size, seqs, names)) parseRecords(data, (size_t)sb.st_size, seqs, names);
                munmap(m, (size_t)sb.st_size);
                close(fd);
                return;
            }
        }
        close(fd);
    }
    const std::string data = slurp_gz(path);
    if (!parseFastaParallel(data.data(), data.size(), seqs, names)) parseRecords(data.data(), data.size(), seqs, names);
}

// ---- fast path: index + pack straight from the text -------------------------------------------------------------
namespace {
struct TextSource {      // mapped file or inflated gzip
    const char* data = nullptr;
    size_t n = 0;
    void* map = nullptr;
    size_t map_len = 0;
    std::string owned;
    ~TextSource() { if (map) munmap(map, map_len); }
    void open(const std::string& path)
    {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd >= 0) {
            struct stat sb;
            unsigned char magic[2] = { 0, 0 };
            if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 2 && pread(fd, magic, 2, 0) == 2 &&
                !(magic[0] == 0x1f && magic[1] == 0x8b)) {
                // (no MAP_POPULATE: the worker threads fault the page-cache pages in, in parallel)
                void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    (void)madvise(m, (size_t)sb.st_size, MADV_WILLNEED);
                    map = m; map_len = (size_t)sb.st_size;
                    data = static_cast<const char*>(m); n = map_len;
                    ::close(fd);
                    return;
                }
            }
            ::close(fd);
        }
        owned = slurp_gz(path);
        data = owned.data(); n = owned.size();
    }
};

template <class F> void parallelFor(size_t count, unsigned nt, F&& f)
{
    if (nt < 2 || count < 2) { for (size_t i = 0; i < count; ++i) f(i, 0u); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&, t] { for (size_t i = t; i < count; i += nt) f(i, t); });
    for (auto& th : pool) th.join();
}

// the sequence text of a record [p, end): calls seg(ptr, len) for every line's bases in order, with the serial
// parser's carriage-return rule (a trailing CR is dropped when the sequence so far is longer than one character)
template <class F> inline void forEachSegment(const char* data, size_t p, size_t end, F&& seg)
{
    size_t total = 0;
    while (p < end) {
        if (data[p] == '\n') { ++p; continue; }
        const char* nl = static_cast<const char*>(std::memchr(data + p, '\n', end - p));
        const size_t le = nl ? (size_t)(nl - data) : end;
        size_t len = le - p;
        if (len > 0 && data[le - 1] == '\r' && total + len > 1) --len;
        if (len) seg(data + p, len);
        total += len;
        p = le < end ? le + 1 : end;
    }
}

struct CodeLut {
    uint8_t c4[256], c2[256];
    CodeLut()
    {
        for (int i = 0; i < 256; ++i) { c4[i] = 4; c2[i] = 0; }      // fourBitCompressor: anything else -> 4; twoBitCompressor: -> 0 ('A')
        const char* b = "ACGT";
        for (int k = 0; k < 4; ++k) { c4[(unsigned char)b[k]] = (uint8_t)k; c2[(unsigned char)b[k]] = (uint8_t)k; }
        c4[(unsigned char)'U'] = 3; c2[(unsigned char)'U'] = 3;
    }
};
const CodeLut kLut;

// 16 bases -> their 4-bit codes in one word (base j at bits 4j), resp. 32 bits of 2-bit codes: the encoders of
// src/fourBitCompressor.cpp:5-41 / src/twoBitCompressor.cpp:5-41 on 16 characters at a time.  For A C G T U the code is
// ((c >> 1) ^ (c >> 2)) & 3 (0x41, 0x43, 0x47, 0x54, 0x55 -> 0, 1, 2, 3, 3); every other character is 4 (4-bit) or 0 (2-bit).
// (Byte by byte through the table the packing ran at 1.5 ns per base: 460 ms of one thread for 30 000 x 10 000 -- hidden behind
//  the HIP start-up with 16 threads, not with the two a rank gets when eight ranks share a host.)
#if defined(__SSE2__)
#include <emmintrin.h>
static inline __m128i codes16(const char* s, __m128i other)
{
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s));
    __m128i ok = _mm_cmpeq_epi8(v, _mm_set1_epi8('A'));
    ok = _mm_or_si128(ok, _mm_cmpeq_epi8(v, _mm_set1_epi8('C')));
    ok = _mm_or_si128(ok, _mm_cmpeq_epi8(v, _mm_set1_epi8('G')));
    ok = _mm_or_si128(ok, _mm_cmpeq_epi8(v, _mm_set1_epi8('T')));
    ok = _mm_or_si128(ok, _mm_cmpeq_epi8(v, _mm_set1_epi8('U')));
    // (16-bit shifts: what crosses a byte boundary lands in bits 6-7, which the mask drops)
    const __m128i code = _mm_and_si128(_mm_xor_si128(_mm_srli_epi16(v, 1), _mm_srli_epi16(v, 2)), _mm_set1_epi8(3));
    return _mm_or_si128(_mm_and_si128(ok, code), _mm_andnot_si128(ok, other));
}
static inline uint64_t pack4x16(const char* s)
{
    const __m128i c = codes16(s, _mm_set1_epi8(4));
    // byte pairs (b0, b1) -> b0 | b1 << 4 in the low byte of every 16-bit lane, then the eight low bytes
    const __m128i t = _mm_and_si128(_mm_or_si128(c, _mm_srli_epi16(c, 4)), _mm_set1_epi16(0x00FF));
    const __m128i p = _mm_packus_epi16(t, _mm_setzero_si128());
    return (uint64_t)_mm_cvtsi128_si64(p);
}
static inline uint32_t pack2x16(const char* s)
{
    const __m128i c = codes16(s, _mm_setzero_si128());
    const __m128i t = _mm_and_si128(_mm_or_si128(c, _mm_srli_epi16(c, 6)), _mm_set1_epi16(0x00FF));      // b0 | b1 << 2: four bits per lane
    uint64_t x = (uint64_t)_mm_cvtsi128_si64(_mm_packus_epi16(t, _mm_setzero_si128()));                   // eight bytes of four bits
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)x;
}
#define DPR_PACK_SIMD 1
#else
#define DPR_PACK_SIMD 0
#endif
}  // namespace

void readSequencesPacked(const std::string& path, bool aligned, long long seed, PackedSequences& out,
                         void (*on_count)(size_t n, void* user), void* user,
                         const std::function<std::vector<int>(const std::vector<std::string>&)>* ids_of)
{
    out = PackedSequences();
    TextSource src;
    src.open(path);
    const char* data = src.data;
    const size_t n = src.n;
    const unsigned nt = hostThreads(64);
    size_t first = 0;
    while (first < n && data[first] != '>' && data[first] != '@') ++first;
    if (first >= n) { out.ok = true; return; }
    // ---- record starts: '>' or '@' at the beginning of a line; a line starting with '+' means FASTQ -> serial parser
    std::vector<std::vector<size_t>> starts(nt);
    std::vector<char> plus(nt, 0);
    const size_t chunk = (n - first + nt - 1) / nt;
    parallelFor(nt, nt, [&](size_t t, unsigned) {
        const size_t lo = first + t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) return;
        if (t == 0) starts[0].push_back(first);
        const char* q = data + lo;
        const char* end = data + hi;
        while (q < end) {
            q = static_cast<const char*>(std::memchr(q, '\n', (size_t)(end - q)));
            if (!q) break;
            ++q;
            if (q < data + n) {
                if (*q == '>' || *q == '@') starts[t].push_back((size_t)(q - data));
                else if (*q == '+') plus[t] = 1;
            }
        }
    });
    for (char c : plus) if (c) return;      // ok stays false
    std::vector<size_t> st;
    for (auto& v : starts) st.insert(st.end(), v.begin(), v.end());
    const size_t nrec = st.size();
    st.push_back(n);
    out.numSequences = nrec;
    if (on_count) on_count(nrec, user);
    // ---- names (input order), body offsets and lengths
    std::vector<std::string> in_names(nrec);
    std::vector<size_t> body(nrec), len(nrec);
    parallelFor(nrec, nt, [&](size_t r, unsigned) {
        size_t p = st[r] + 1;
        const size_t end = st[r + 1];
        size_t e = p;
        while (e < end && !std::isspace((unsigned char)data[e])) ++e;
        in_names[r].assign(data + p, e - p);
        while (e < end && data[e] != '\n') ++e;
        body[r] = e < end ? e + 1 : end;
        size_t l = 0;
        forEachSegment(data, body[r], end, [&](const char*, size_t k) { l += k; });
        len[r] = l;
    });
    // ---- slots: the seeded shuffle, or the caller's rule (which sees the names)
    const std::vector<int> ids = ids_of ? (*ids_of)(in_names) : shuffledIds(nrec, seed);
    if (ids.size() != nrec) { std::fprintf(stderr, "ERROR: internal: slot map of %zu entries for %zu records\n", ids.size(), nrec); std::exit(1); }
    out.names.assign(nrec, std::string());
    for (size_t r = 0; r < nrec; ++r) out.names[(size_t)ids[r]] = std::move(in_names[r]);
    if (aligned) {
        size_t slot0 = 0;
        for (size_t r = 0; r < nrec; ++r) if (ids[r] == 0) slot0 = r;
        out.seqLen = (int)len[slot0];
        const size_t W = ((size_t)out.seqLen + 15) / 16;
        out.flat.resize(nrec * W);            // (every word is written below)
        parallelFor(nrec, nt, [&](size_t r, unsigned) {
            uint64_t* dst = out.flat.data() + (size_t)ids[r] * W;
            const size_t have = (len[r] + 15) / 16;
            size_t w = 0, j = 0;              // word index, base index inside the word
            uint64_t v = 0;
            forEachSegment(data, body[r], st[r + 1], [&](const char* s, size_t k) {
                size_t i = 0;
#if DPR_PACK_SIMD
                // 16 characters at a time into the running word at any bit offset (lines need not be multiples of 16)
                for (; i + 16 <= k && w < W; i += 16) {
                    const uint64_t c = pack4x16(s + i);
                    if (j == 0) dst[w++] = c;
                    else { dst[w++] = v | (c << (4 * j)); v = c >> (64 - 4 * j); }
                }
#endif
                for (; i < k && w < W; ++i) {
                    v |= (uint64_t)kLut.c4[(unsigned char)s[i]] << (4 * j);
                    if (++j == 16) { dst[w++] = v; v = 0; j = 0; }
                }
            });
            if (j && w < W) dst[w++] = v;     // last, partly filled word of the sequence (tail zero as dpr_pack4 leaves it)
            // a sequence shorter than the one in slot 0: the missing bases are marked invalid (code 4)
            for (; w < W; ++w) dst[w] = 0x4444444444444444ull;
            if (len[r] < (size_t)out.seqLen && have > 0 && have <= W) {
                const size_t rr = len[r] % 16;
                if (rr) dst[have - 1] |= 0x4444444444444444ull << (4 * rr);
            }
        });
    } else {
        out.lens.assign(nrec, 0); out.off.assign(nrec, 0);
        std::vector<uint64_t> nw(nrec);
        for (size_t r = 0; r < nrec; ++r) { out.lens[(size_t)ids[r]] = len[r]; nw[(size_t)ids[r]] = (len[r] + 31) / 32; }
        uint64_t total = 0;
        for (size_t s2 = 0; s2 < nrec; ++s2) { out.off[s2] = total; total += nw[s2]; }  // exclusive scan (src/mash.cu:109-119)
        out.flat.assign(total + 1, 0);
        parallelFor(nrec, nt, [&](size_t r, unsigned) {
            uint64_t* dst = out.flat.data() + out.off[(size_t)ids[r]];
            size_t w = 0, j = 0;
            uint64_t v = 0;
            forEachSegment(data, body[r], st[r + 1], [&](const char* s, size_t k) {
                size_t i = 0;
#if DPR_PACK_SIMD
                for (; i + 16 <= k; i += 16) {
                    const uint64_t c = pack2x16(s + i);      // 32 bits
                    v |= c << (2 * j);
                    if (j >= 16) { dst[w++] = v; v = j > 16 ? c >> (64 - 2 * j) : 0; j -= 16; }
                    else j += 16;
                }
#endif
                for (; i < k; ++i) {
                    v |= (uint64_t)kLut.c2[(unsigned char)s[i]] << (2 * j);
                    if (++j == 32) { dst[w++] = v; v = 0; j = 0; }
                }
            });
            if (j) dst[w++] = v;
        });
    }
    out.ok = true;
}

std::vector<int> shuffledIds(size_t n, long long seed)
{
    std::vector<int> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i] = (int)i;
    if (seed >= 0) {
        std::mt19937 rnd((uint32_t)seed);
        std::shuffle(ids.begin(), ids.end(), rnd);
    }
    return ids;
}

void MatrixReader::read(const std::string& path)
{
    FILE* fp = std::fopen(path.c_str(), "r");
    if (!fp) {
        std::cerr << "Cannot open file: " << path << std::endl;  // src/tree_generation.cu:596-599
        std::exit(1);
    }
    std::string data;
    {
        std::vector<char> buf(1 << 22);
        size_t got;
        while ((got = std::fread(buf.data(), 1, buf.size(), fp)) > 0) data.append(buf.data(), got);
        std::fclose(fp);
    }
    const char* s = data.c_str();
    char* endp = nullptr;
    const long n = std::strtol(s, &endp, 10);
    if (endp == s || n < 2) die("ERROR: PHYLIP header must give the number of sequences (>= 2)");
    numSequences = (int)n;
    name.assign((size_t)n, "");
    lower.assign((size_t)n * (size_t)(n - 1) / 2, 0.0);
    const char* p = endp;
    while (*p && *p != '\n') ++p;  // rest of the header line
    if (*p) ++p;
    size_t w = 0;
    for (long i = 0; i < n; ++i) {
        while (*p == '\n' || *p == '\r') ++p;  // tolerate blank lines
        if (!*p) die("ERROR: PHYLIP matrix ends before row " + std::to_string(i));
        const char* q = p;
        while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;
        name[(size_t)i].assign(p, (size_t)(q - p));
        p = q;
        for (long j = 0; j < i; ++j) {
            while (*p == ' ' || *p == '\t') ++p;  // superset of the reference: runs of separators
            if (!*p || *p == '\n' || *p == '\r') die("ERROR: PHYLIP row " + std::to_string(i) + " has fewer than " + std::to_string(i) + " values");
            char* e2 = nullptr;
            const float v = std::strtof(p, &e2);  // stof: float precision (src/matrix_reader.cu:42)
            if (e2 == p) die("ERROR: cannot parse a distance in PHYLIP row " + std::to_string(i));
            lower[w++] = (double)v;
            p = e2;
        }
        while (*p && *p != '\n') ++p;  // upper-triangle part of a square matrix is ignored
        if (*p) ++p;
    }
}

void writeNewickFromMerges(std::ostream& os, const std::vector<std::string>& name, const std::vector<int32_t>& mx,
                           const std::vector<int32_t>& my, const std::vector<double>& bx,
                           const std::vector<double>& by, double last_d)
{
    const int N = (int)name.size();
    struct Child { int node; double len; };
    std::vector<Child> kids((size_t)(2 * N) * 2, Child{ -1, 0.0 });  // two children per internal node
    std::vector<int> realID((size_t)N);
    for (int i = 0; i < N; ++i) realID[(size_t)i] = i;
    int ID = N;
    for (int it = 0; it < N - 2; ++it) {
        const int x = mx[(size_t)it], y = my[(size_t)it];
        kids[(size_t)ID * 2] = Child{ realID[(size_t)x], bx[(size_t)it] };
        kids[(size_t)ID * 2 + 1] = Child{ realID[(size_t)y], by[(size_t)it] };
        realID[(size_t)x] = ID++;
        realID[(size_t)y] = realID[(size_t)(N - it - 1)];
    }
    const int root = 2 * N - 2;
    kids[(size_t)root * 2] = Child{ realID[0], last_d * 0.5 };
    kids[(size_t)root * 2 + 1] = Child{ realID[1], last_d * 0.5 };
    // iterative pre-order print: "(" child ":" len "," child ":" len ")"
    struct Frame { int node; int next; };
    std::vector<Frame> st;
    st.push_back(Frame{ root, 0 });
    TextBuf out;
    out.s.reserve((size_t)N * 40);
    out.put('(');
    while (!st.empty()) {
        Frame& f = st.back();
        if (f.next == 2) {
            st.pop_back();
            if (st.empty()) break;
            Frame& up = st.back();
            const Child& c = kids[(size_t)up.node * 2 + (size_t)(up.next - 1)];
            out.put(':'); out.putLength(c.len); out.put(up.next == 2 ? ')' : ',');
            continue;
        }
        const Child& c = kids[(size_t)f.node * 2 + (size_t)f.next];
        f.next++;
        if (c.node >= N) {
            out.put('(');
            st.push_back(Frame{ c.node, 0 });
        } else {
            out.put(name[(size_t)c.node]); out.put(':'); out.putLength(c.len); out.put(f.next == 2 ? ')' : ',');
        }
    }
    out.put(";\n");
    os.write(out.s.data(), (std::streamsize)out.s.size());
}

}  // namespace dipper
