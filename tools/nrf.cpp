// nrf -- normalised Robinson-Foulds distance between two Newick trees over the same tip names (bench / test
// infrastructure, not product).  The reference's authors validate their trees this way against the simulated true tree
// (scripts/nrf.sh:26,36-60: MAPLE's --inputRFtrees against REF_TREE; MAPLE is a third-party tool, absent here).  Trees are
// compared UNROOTED: a tree's non-trivial bipartitions (both sides >= 2 tips) are hashed (128-bit sums of per-name
// random words; of the two sides of an edge the one with the smaller hash is the canonical one), and
//     RF = |A \ B| + |B \ A|,   nRF = RF / (|A| + |B|)      (= RF / (2 (n - 3)) for two binary trees).
// Edges of length zero are ordinary edges (no collapsing).  Usage: nrf a.nwk b.nwk  ->  one JSON line on stdout.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

[[noreturn]] void die(const std::string& m)
{
    std::fprintf(stderr, "nrf: %s\n", m.c_str());
    std::exit(1);
}

std::string slurp(const char* path)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) die(std::string("cannot open ") + path);
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    std::fclose(f);
    return s;
}

struct H128 {
    uint64_t a, b;
    bool operator<(const H128& o) const { return a != o.a ? a < o.a : b < o.b; }
    bool operator==(const H128& o) const { return a == o.a && b == o.b; }
};

uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
H128 name_hash(const char* s, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)s[i]; h *= 0x100000001b3ull; }
    return { mix64(h + 0x9E3779B97F4A7C15ull), mix64(h ^ 0xD1B54A32D192ED03ull) };
}

struct Parsed {
    std::vector<int32_t> parent;
    std::vector<H128> hsum;        // sum of the tip hashes below the node
    std::vector<int32_t> cnt;
    int64_t tips = 0;
    H128 total{ 0, 0 };
};

Parsed parse(const std::string& s)
{
    Parsed t;
    std::vector<int32_t> st;
    auto add = [&]() {
        const int32_t v = (int32_t)t.parent.size();
        t.parent.push_back(st.empty() ? -1 : st.back());
        t.hsum.push_back({ 0, 0 });
        t.cnt.push_back(0);
        return v;
    };
    size_t i = 0;
    const size_t n = s.size();
    bool after_close = false;
    while (i < n) {
        const char c = s[i];
        if (c == '(') {
            if (st.empty() && !t.parent.empty()) die("a second tree (or text) behind the first one");      // node 0 is THE root
            st.push_back(add()); ++i; after_close = false;
        }
        else if (c == ',') { ++i; after_close = false; }
        else if (c == ')') { if (st.empty()) die("unbalanced ')'"); st.pop_back(); ++i; after_close = true; }
        else if (c == ':') { ++i; while (i < n && s[i] != ',' && s[i] != ')' && s[i] != '(' && s[i] != ';') ++i; }
        else if (c == ';' || c == '\n' || c == '\r' || c == ' ' || c == '\t') { ++i; }
        else {
            size_t j = i;
            if (c == '\'') { j = s.find('\'', i + 1); if (j == std::string::npos) die("unterminated quote"); ++j; }
            else while (j < n && s[j] != ':' && s[j] != ',' && s[j] != '(' && s[j] != ')' && s[j] != ';') ++j;
            if (!after_close) {         // a tip label (labels behind ')' name internal nodes: ignored)
                // a label outside every '(' would get parent -1 and the sums below would index hsum[-1]: truncated or
                // malformed output of the command under test must end in a diagnostic, not in an out-of-bounds write
                if (st.empty()) die("label outside the tree (input without enclosing parentheses, or text before '(')");
                const int32_t v = add();
                t.hsum[(size_t)v] = name_hash(s.data() + i, j - i);
                t.cnt[(size_t)v] = 1;
                ++t.tips;
            }
            i = j;
            after_close = false;
        }
    }
    if (!st.empty()) die("unbalanced '('");
    for (size_t v = t.parent.size(); v-- > 1;) {        // children come after their parents
        const size_t p = (size_t)t.parent[v];
        t.hsum[p].a += t.hsum[v].a; t.hsum[p].b += t.hsum[v].b;
        t.cnt[p] += t.cnt[v];
    }
    if (!t.parent.empty()) t.total = t.hsum[0];
    return t;
}

std::vector<H128> splits(const Parsed& t)
{
    std::vector<H128> out;
    for (size_t v = 1; v < t.parent.size(); ++v) {
        const int64_t c = t.cnt[v];
        if (c < 2 || c > t.tips - 2) continue;
        const H128 h = t.hsum[v], o = { t.total.a - h.a, t.total.b - h.b };
        out.push_back(h < o ? h : o);
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());     // the two edges at a degree-2 root are one bipartition
    return out;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: nrf a.nwk b.nwk\n"); return 2; }
    const Parsed ta = parse(slurp(argv[1])), tb = parse(slurp(argv[2]));
    if (ta.tips != tb.tips || !(ta.total == tb.total)) die("the two trees are not over the same set of tip names");
    const std::vector<H128> A = splits(ta), B = splits(tb);
    size_t i = 0, j = 0, common = 0;
    while (i < A.size() && j < B.size()) {
        if (A[i] == B[j]) { ++common; ++i; ++j; }
        else if (A[i] < B[j]) ++i;
        else ++j;
    }
    const size_t rf = (A.size() - common) + (B.size() - common);
    const double nrf = (A.size() + B.size()) ? (double)rf / (double)(A.size() + B.size()) : 0.0;
    std::printf("{\"tips\": %lld, \"splits_a\": %zu, \"splits_b\": %zu, \"common\": %zu, \"rf\": %zu, \"nrf\": %.9g}\n",
                (long long)ta.tips, A.size(), B.size(), common, rf, nrf);
    return 0;
}
