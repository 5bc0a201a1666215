// Readers and writers of the dipper host side.
#include "dipper_host.hpp"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>

namespace dipper {

void die(const std::string& msg)
{
    std::cerr << msg << std::endl;
    std::exit(1);
}

void gpuCheck(int rc, const char* what)
{
    if (rc < 0) {
        std::fprintf(stderr, "Gpu_ERROR: %s failed: %s\n", what, dpr_last_error());
        std::exit(1);
    }
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
    gzclose(f);
    return data;
}

void readSequences(const std::string& path, std::vector<std::string>& seqs, std::vector<std::string>& names)
{
    const std::string data = slurp_gz(path);
    const size_t n = data.size();
    size_t p = 0;
    // jump to the first header
    while (p < n && data[p] != '>' && data[p] != '@') ++p;
    while (p < n) {
        ++p;  // header char
        size_t e = p;
        while (e < n && !std::isspace((unsigned char)data[e])) ++e;
        names.emplace_back(data, p, e - p);
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
            seq.append(data, p, le - p);
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
    os << "(";
    while (!st.empty()) {
        Frame& f = st.back();
        if (f.next == 2) {
            st.pop_back();
            if (st.empty()) break;
            Frame& up = st.back();
            const Child& c = kids[(size_t)up.node * 2 + (size_t)(up.next - 1)];
            os << ":" << c.len << (up.next == 2 ? ')' : ',');
            continue;
        }
        const Child& c = kids[(size_t)f.node * 2 + (size_t)f.next];
        f.next++;
        if (c.node >= N) {
            os << "(";
            st.push_back(Frame{ c.node, 0 });
        } else {
            os << name[(size_t)c.node] << ":" << c.len << (f.next == 2 ? ')' : ',');
        }
    }
    os << ";\n";
}

}  // namespace dipper
