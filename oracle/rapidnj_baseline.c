/*
 * rapidnj_baseline.c -- CPU BASELINE ONLY (bench.py's cpu_baseline leg and its CPU test).  Not an oracle:
 * nothing is checked against it and nothing under dipper_amd/ links, imports or calls it.
 *
 * north_star asks for a CPU NJ baseline "RapidNJ, since the reference ships no CPU path"
 * (scripts/experiment.sh:123 runs `rapidnj <phylip> -i pd -c 32`).  RapidNJ is not installed in the image
 * and there is no network, so this is a from-scratch implementation of its published search strategy
 * (Simonsen, Mailund, Pedersen: "Rapid Neighbour-Joining", WABI 2008) -- exact neighbour joining:
 *
 *   every node keeps the other nodes that existed when it was created, sorted by distance (S-row); the
 *   pair (a,b) is found through the row of the later-created node.  A search walks each live row in
 *   ascending distance and stops at the first entry with  d - u_row - u_max >= q_min  (u = U/(n-2),
 *   u_max over the live nodes), because no later entry of that row can beat the best q found so far.
 *   After a join the new node's row is computed, radix-sorted and stored in the slot of one child; stale
 *   entries of other rows (a deleted node, or a slot that now holds a newer node) are skipped or harmless
 *   (a slot's current distance is always a valid candidate; the pair is covered by the newer row).
 *
 * OpenMP over the rows with a shared q_min (relaxed reads only tighten or loosen the pruning, never the
 * result).  Ties may be broken differently from the reference; on tie-free input the joins are the same.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RNJ_API __attribute__((visibility("default")))

typedef struct { float key; int32_t j; } rnj_ent;   /* key <= true distance (rounded down) */

static inline float rnj_key(double d)
{
    float f = (float)d;
    if ((double)f > d) f = nextafterf(f, -INFINITY);
    return f;
}

/* ascending LSD radix sort of entries by float key (sign-magnitude -> monotone unsigned) */
static void rnj_sort(rnj_ent *a, rnj_ent *tmp, int64_t n)
{
    uint32_t *k = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n * 2);
    uint32_t *k2 = k + n;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t b;
        memcpy(&b, &a[i].key, 4);
        k[i] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    rnj_ent *src = a, *dst = tmp;
    uint32_t *ks = k, *kd = k2;
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass * 11, bits = pass == 2 ? 10 : 11;
        const uint32_t mask = (1u << bits) - 1u;
        int64_t cnt[2049];
        memset(cnt, 0, sizeof(cnt));
        for (int64_t i = 0; i < n; ++i) cnt[((ks[i] >> shift) & mask) + 1]++;
        for (uint32_t b = 0; b < mask + 1; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; ++i) {
            const int64_t p = cnt[(ks[i] >> shift) & mask]++;
            dst[p] = src[i]; kd[p] = ks[i];
        }
        rnj_ent *t = src; src = dst; dst = t;
        uint32_t *tk = ks; ks = kd; kd = tk;
    }
    if (src != a) memcpy(a, src, sizeof(rnj_ent) * (size_t)n);
    free(k);
}

/* D: N x ld, full symmetric, modified in place.  Node ids: tips 0..N-1, join t creates node N+t.
 * Outputs per join (N-2 of them): children ids and branch lengths; the last two live nodes and their
 * distance.  Returns the number of joins or <0. */
RNJ_API int64_t orc_rapidnj_run(double *D, int64_t N, int64_t ld, int threads, int32_t *child_a, int32_t *child_b,
                                double *bl_a, double *bl_b, int32_t *last_pair, double *last_d)
{
    if (N < 3) return -1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    rnj_ent **row = (rnj_ent **)calloc((size_t)N, sizeof(rnj_ent *));
    int64_t *len = (int64_t *)calloc((size_t)N, sizeof(int64_t));
    double *U = (double *)malloc(sizeof(double) * (size_t)N);
    double *u = (double *)malloc(sizeof(double) * (size_t)N);
    char *alive = (char *)malloc((size_t)N);
    int32_t *node = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);   /* node id held by a slot */
    /* initial rows: slot i lists the slots j < i (the pair is found through the higher slot) */
#pragma omp parallel
    {
        rnj_ent *tmp = (rnj_ent *)malloc(sizeof(rnj_ent) * (size_t)N);
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < N; ++i) {
            double s = 0;
            for (int64_t j = 0; j < N; ++j) if (j != i) s += D[i * ld + j];
            U[i] = s; alive[i] = 1; node[i] = (int32_t)i;
            len[i] = i;
            row[i] = (rnj_ent *)malloc(sizeof(rnj_ent) * (size_t)(i > 0 ? i : 1));
            for (int64_t j = 0; j < i; ++j) { row[i][j].key = rnj_key(D[i * ld + j]); row[i][j].j = (int32_t)j; }
            rnj_sort(row[i], tmp, i);
        }
        free(tmp);
    }
    rnj_ent *tmp = (rnj_ent *)malloc(sizeof(rnj_ent) * (size_t)N);
    int64_t n = N, joins = 0;
    while (n > 2) {
        const double r = (double)(n - 2);
        double umax = -INFINITY;
        for (int64_t i = 0; i < N; ++i) if (alive[i]) { u[i] = U[i] / r; if (u[i] > umax) umax = u[i]; }
        double qmin = INFINITY, bq = INFINITY; int64_t bi = -1, bj = -1;
#pragma omp parallel
        {
            double lq = INFINITY; int64_t li = -1, lj = -1;
#pragma omp for schedule(dynamic, 64) nowait
            for (int64_t i = 0; i < N; ++i) {
                if (!alive[i]) continue;
                const rnj_ent *S = row[i];
                const double ui = u[i];
                double bound;
#pragma omp atomic read
                bound = qmin;
                if (lq < bound) bound = lq;
                for (int64_t t = 0; t < len[i]; ++t) {
                    if ((double)S[t].key - ui - umax >= bound) break;
                    const int64_t j = S[t].j;
                    if (!alive[j] || j == i) continue;
                    const double q = D[i * ld + j] - ui - u[j];
                    if (q < lq) { lq = q; li = i; lj = j; if (q < bound) bound = q; }
                }
                if (lq < INFINITY) {
                    double cur;
#pragma omp atomic read
                    cur = qmin;
                    if (lq < cur) {
#pragma omp critical(rnj_min)
                        { if (lq < qmin) qmin = lq; }
                    }
                }
            }
#pragma omp critical(rnj_best)
            {
                if (li >= 0 && (bi < 0 || lq < bq || (lq == bq && (li < bi || (li == bi && lj < bj))))) { bq = lq; bi = li; bj = lj; }
            }
        }
        if (bi < 0) { joins = -2; break; }
        int64_t a = bi < bj ? bi : bj, b = bi < bj ? bj : bi;
        const double d = D[a * ld + b];
        double la = (d + u[a] - u[b]) * 0.5, lb = d - la;
        if (la < 0) { lb += la; la = 0; }
        if (lb < 0) { la += lb; lb = 0; }
        child_a[joins] = node[a]; child_b[joins] = node[b]; bl_a[joins] = la; bl_b[joins] = lb;
        /* new node in slot a, slot b dies */
        alive[b] = 0;
        double ua = 0;
        int64_t m_cnt = 0;
#pragma omp parallel for reduction(+ : ua) schedule(static) if (n > 4096)
        for (int64_t m = 0; m < N; ++m) {
            if (!alive[m] || m == a) continue;
            const double dam = D[a * ld + m], dbm = D[b * ld + m];
            const double v = (dam + dbm - d) * 0.5;
            U[m] += v - dam - dbm;
            D[a * ld + m] = v; D[m * ld + a] = v;
            ua += v;
        }
        U[a] = ua;
        for (int64_t m = 0; m < N; ++m)
            if (alive[m] && m != a) { tmp[m_cnt].key = rnj_key(D[a * ld + m]); tmp[m_cnt].j = (int32_t)m; ++m_cnt; }
        free(row[a]);
        row[a] = (rnj_ent *)malloc(sizeof(rnj_ent) * (size_t)(m_cnt > 0 ? m_cnt : 1));
        memcpy(row[a], tmp, sizeof(rnj_ent) * (size_t)m_cnt);
        rnj_sort(row[a], tmp, m_cnt);
        len[a] = m_cnt;
        free(row[b]); row[b] = NULL; len[b] = 0;
        node[a] = (int32_t)(N + joins);
        ++joins; --n;
    }
    if (joins >= 0) {
        int64_t p0 = -1, p1 = -1;
        for (int64_t i = 0; i < N; ++i) if (alive[i]) { if (p0 < 0) p0 = i; else p1 = i; }
        last_pair[0] = node[p0]; last_pair[1] = node[p1];
        *last_d = D[p0 * ld + p1];
    }
    for (int64_t i = 0; i < N; ++i) free(row[i]);
    free(row); free(len); free(U); free(u); free(alive); free(node); free(tmp);
    return joins;
}
