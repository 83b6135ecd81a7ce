/* Experiment (CPU model, not product): how much tree work a half traversal keeps when the first K leaves to the right of
 * every query are tested directly (a window over the Morton order) and the chain / descent only covers leaves beyond it.
 * Tree = the oracle's (orc_build_hierarchy + orc_refit), boxes FP64.  Counts per category; per-wave (64 queries) maxima. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static int ov(const double *a, const double *b)
{
    return (a[0] - b[1]) * (b[0] - a[1]) > 0 && (a[2] - b[3]) * (b[2] - a[3]) > 0 && (a[4] - b[5]) * (b[4] - a[5]) > 0;
}
/* out[0] window tests, [1] window hits, [2] hops in wave, [3] tests in wave, [4] chain steps above the wave (sum over waves),
 * [5] tests above (lane-level), [6] phase-2 visits, [7] leaf hits from the tree, [8] sum over waves of max hops in wave,
 * [9] descents started, [10] sum over waves of max per-lane private visits (tests in wave + p2 visits), [11] max stack */
void window_sim(int n, const int32_t *left, const int32_t *right, const int32_t *rf, const int32_t *rl, const double *boxes, int K, uint64_t *out)
{
    int32_t *split = malloc(sizeof(int32_t) * n), *node_of = malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n - 1; ++i) { int l = left[i]; split[i] = l >= n - 1 ? l - (n - 1) : rl[l]; node_of[split[i]] = i; }
    memset(out, 0, sizeof(uint64_t) * 16);
    int32_t stack[256];
    for (int g0 = 0; g0 < n; g0 += 64) {
        int g_last = g0 + 63 < n - 1 ? g0 + 63 : n - 1;
        uint64_t maxhops = 0, maxpriv = 0;
        /* chain above the wave */
        { int t = g_last; while (t < n - 1) { ++out[4]; t = rl[node_of[t]]; } }
        for (int j = g0; j <= g_last; ++j) {
            const double *qb = boxes + 6 * (size_t)((n - 1) + j);
            int T = j + K < n - 1 ? j + K : n - 1;
            for (int k = j + 1; k <= T; ++k) { ++out[0]; if (ov(qb, boxes + 6 * (size_t)((n - 1) + k))) ++out[1]; }
            uint64_t hops = 0, priv = 0;
            int s = j;
            while (s < n - 1) {
                int i = node_of[s], L = rl[i];
                int in_wave = s < g_last;
                if (in_wave) { ++hops; ++out[2]; }
                if (L > T) {
                    if (in_wave) { ++out[3]; ++priv; } else ++out[5];
                    int c = right[i];
                    if (ov(qb, boxes + 6 * (size_t)c)) {
                        if (c >= n - 1) ++out[7];
                        else {
                            ++out[9];
                            int sp = 0; stack[sp++] = c;
                            while (sp) {
                                int nd = stack[--sp]; ++out[6]; ++priv;
                                int rn = split[nd];
                                int cl = left[nd], cr = right[nd];
                                if (rn > T && ov(qb, boxes + 6 * (size_t)cl)) { if (cl >= n - 1) ++out[7]; else stack[sp++] = cl; }
                                if (ov(qb, boxes + 6 * (size_t)cr)) { if (cr >= n - 1) ++out[7]; else stack[sp++] = cr; }
                                if ((uint64_t)sp > out[11]) out[11] = sp;
                            }
                        }
                    }
                }
                s = L;
            }
            if (hops > maxhops) maxhops = hops;
            if (priv > maxpriv) maxpriv = priv;
        }
        out[8] += maxhops; out[10] += maxpriv;
    }
    free(split); free(node_of);
}

/* Level-synchronous phase 2 per workgroup of G queries: all (query, subtree) items the chains of G consecutive queries hit go
 * into one frontier; each level is processed 64 items per wave-step.  out[0] wave-steps (sum over workgroups of sum over levels
 * of ceil(cnt / 64)), [1] levels (sum over workgroups), [2] max frontier, [3] items (= phase-2 visits), [4] workgroups,
 * [5] sum over workgroups of levels where cnt < 32 */
void bfs_sim(int n, const int32_t *left, const int32_t *right, const int32_t *rl, const double *boxes, int G, uint64_t *out)
{
    int32_t *node_of = malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n - 1; ++i) { int l = left[i]; int sp = l >= n - 1 ? l - (n - 1) : rl[l]; node_of[sp] = i; }
    memset(out, 0, sizeof(uint64_t) * 8);
    size_t cap = 1 << 22;
    int32_t *fq = malloc(sizeof(int32_t) * cap), *fn = malloc(sizeof(int32_t) * cap), *gq = malloc(sizeof(int32_t) * cap), *gn = malloc(sizeof(int32_t) * cap);
    for (int w0 = 0; w0 < n; w0 += G) {
        size_t cnt = 0;
        for (int j = w0; j < w0 + G && j < n; ++j) {
            const double *qb = boxes + 6 * (size_t)((n - 1) + j);
            int s = j;
            while (s < n - 1) { int i = node_of[s]; int c = right[i]; if (c < n - 1 && ov(qb, boxes + 6 * (size_t)c)) { fq[cnt] = j; fn[cnt++] = c; } s = rl[i]; }
        }
        ++out[4];
        while (cnt) {
            out[0] += (cnt + 63) / 64; ++out[1]; out[3] += cnt; if (cnt > out[2]) out[2] = cnt; if (cnt < 32) ++out[5];
            size_t m = 0;
            for (size_t k = 0; k < cnt; ++k) {
                const double *qb = boxes + 6 * (size_t)((n - 1) + fq[k]);
                int nd = fn[k], cl = left[nd], cr = right[nd];
                if (cl < n - 1 && ov(qb, boxes + 6 * (size_t)cl)) { gq[m] = fq[k]; gn[m++] = cl; }
                if (cr < n - 1 && ov(qb, boxes + 6 * (size_t)cr)) { gq[m] = fq[k]; gn[m++] = cr; }
            }
            int32_t *t; t = fq; fq = gq; gq = t; t = fn; fn = gn; gn = t; cnt = m;
        }
    }
    free(node_of); free(fq); free(fn); free(gq); free(gn);
}

/* Phase-2 visits of the half traversal by size of the visited node (log2 of its leaf count), and how many DISTINCT (64-query group, node)
 * pairs they are: out[2*k] visits at nodes with 2^k <= leaves < 2^(k+1), out[2*k+1] distinct (group, node) pairs among them. k < 24. */
#include <stdio.h>
static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return x < y ? -1 : x > y; }
void p2_size_sim(int n, const int32_t *left, const int32_t *right, const int32_t *rf, const int32_t *rl, const double *boxes, uint64_t *out)
{
    int32_t *node_of = malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n - 1; ++i) { int l = left[i]; int sp = l >= n - 1 ? l - (n - 1) : rl[l]; node_of[sp] = i; }
    memset(out, 0, sizeof(uint64_t) * 48);
    size_t cap = 1 << 23, cnt = 0;
    uint64_t *vis = malloc(sizeof(uint64_t) * cap);      /* (log2size << 56) | (group << 28) | node */
    int32_t stack[256];
    for (int j = 0; j < n; ++j) {
        const double *qb = boxes + 6 * (size_t)((n - 1) + j);
        int s = j;
        while (s < n - 1) {
            int i = node_of[s]; int c = right[i];
            if (c < n - 1 && ov(qb, boxes + 6 * (size_t)c)) {
                int sp = 0; stack[sp++] = c;
                while (sp) {
                    int nd = stack[--sp];
                    int sz = rl[nd] - rf[nd] + 1, k = 0; while ((2 << k) <= sz) ++k;
                    if (cnt < cap) vis[cnt++] = ((uint64_t)k << 56) | ((uint64_t)(j >> 6) << 28) | (uint64_t)nd;
                    int cl = left[nd], cr = right[nd];
                    if (cl < n - 1 && ov(qb, boxes + 6 * (size_t)cl)) stack[sp++] = cl;
                    if (cr < n - 1 && ov(qb, boxes + 6 * (size_t)cr)) stack[sp++] = cr;
                }
            }
            s = rl[i];
        }
    }
    qsort(vis, cnt, sizeof(uint64_t), cmp_u64);
    for (size_t i = 0; i < cnt; ++i) { int k = (int)(vis[i] >> 56); ++out[2 * k]; if (i == 0 || vis[i] != vis[i - 1]) ++out[2 * k + 1]; }
    free(vis); free(node_of);
}

/* Half traversal where a hit subtree of at most F leaves is not descended but handed over as a RANGE (every leaf box in it tested
 * directly).  Per F: out[0] tree visits in phase 2, [1] range items, [2] leaf tests in ranges, [3] sum over waves (64 queries) of the
 * level-synchronous step count of the remaining descents (levels with any item), [4] leaf hits from ranges, [5] leaf hits from the tree,
 * [6] sum over waves of ceil(range items / 8) (flat steps at 8 items per wave-step) */
void flat_sim(int n, const int32_t *left, const int32_t *right, const int32_t *rf, const int32_t *rl, const double *boxes, int F, uint64_t *out)
{
    int32_t *node_of = malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n - 1; ++i) { int l = left[i]; int sp = l >= n - 1 ? l - (n - 1) : rl[l]; node_of[sp] = i; }
    memset(out, 0, sizeof(uint64_t) * 8);
    size_t cap = 1 << 20;
    int32_t *fq = malloc(sizeof(int32_t) * cap), *fn = malloc(sizeof(int32_t) * cap), *gq = malloc(sizeof(int32_t) * cap), *gn = malloc(sizeof(int32_t) * cap);
    for (int w0 = 0; w0 < n; w0 += 64) {
        size_t cnt = 0; uint64_t items = 0;
        #define HANDLE(c, qj, qb, Q, N, M) do { int c_ = (c); if (c_ >= n - 1) { ++out[5]; } else { int sz_ = rl[c_] - rf[c_] + 1; \
            if (sz_ <= F) { ++out[1]; ++items; out[2] += sz_; for (int k_ = rf[c_]; k_ <= rl[c_]; ++k_) if (ov(qb, boxes + 6 * (size_t)((n - 1) + k_))) ++out[4]; } \
            else { Q[M] = qj; N[M++] = c_; } } } while (0)
        for (int j = w0; j < w0 + 64 && j < n; ++j) {
            const double *qb = boxes + 6 * (size_t)((n - 1) + j);
            int s = j;
            while (s < n - 1) { int i = node_of[s]; int c = right[i]; if (ov(qb, boxes + 6 * (size_t)c)) HANDLE(c, j, qb, fq, fn, cnt); s = rl[i]; }
        }
        while (cnt) {
            ++out[3]; out[0] += cnt;
            size_t m = 0;
            for (size_t k = 0; k < cnt; ++k) {
                const double *qb = boxes + 6 * (size_t)((n - 1) + fq[k]);
                int nd = fn[k], cl = left[nd], cr = right[nd];
                if (ov(qb, boxes + 6 * (size_t)cl)) HANDLE(cl, fq[k], qb, gq, gn, m);
                if (ov(qb, boxes + 6 * (size_t)cr)) HANDLE(cr, fq[k], qb, gq, gn, m);
            }
            int32_t *t; t = fq; fq = gq; gq = t; t = fn; fn = gn; gn = t; cnt = m;
        }
        out[6] += (items + 7) / 8;
    }
    free(node_of); free(fq); free(fn); free(gq); free(gn);
}
