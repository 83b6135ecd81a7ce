/* CPU model (not product): the half traversal's wave-steps with BINARY records (today's k_descend_half) and with records that hold two binary levels
 * ("a wider node, which halves the wave-steps phase 2 pays ~74 instructions each for -- unbuilt and unmodelled").
 *
 * Tree = the oracle's (orc_build_hierarchy + orc_refit), boxes FP64 (the device descends conservative fp32 boxes: a few more visits).  A wave owns 64 consecutive
 * leaves, as on the device:
 *   phase 1a  per lane: hops s -> rl[node_of(s)] while s < g_last, testing the right child of node_of(s); the wave's steps = the longest lane's hops
 *   phase 1b  the shared chain from g_last upwards, one step a hop for the whole wave
 *   phase 2   lock step: every lane with a node pops it, tests both children, descends left and pushes right; idle lanes take the top of a busy lane's stack when
 *             16 or more lanes are idle (SHARE_MIN_IDLE, cd_traverse.h); the wave's steps = iterations until every stack is empty
 * WIDE: a phase-2 step fetches a node AND its two children (one 128-byte record: up to four grandchild boxes) -- the lane tests the children it would have visited in
 * its next two binary steps at once: visiting node X costs one step and yields the hit GRANDCHILDREN (a hit leaf child is a candidate at once, a hit internal child is
 * expanded in the same step).  Chain hops likewise two at a time (a record would hold its right sibling and the next one's).  Any node may head a wide record (the upper
 * bound of the gain: every binary node keeps a 128-byte record).
 * out: [0] waves [1] 1a steps [2] 1b steps [3] phase-2 steps binary [4] phase-2 steps wide [5] 1a steps wide [6] 1b steps wide [7] node visits binary (lanes x steps busy)
 *      [8] phase-2 lane-visits binary [9] phase-2 lane-visits wide [10] box tests binary [11] box tests wide [12] longest wave's phase-2 steps binary [13] ... wide */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static int ov(const double *a, const double *b)
{
    return (a[0] - b[1]) * (b[0] - a[1]) > 0 && (a[2] - b[3]) * (b[2] - a[3]) > 0 && (a[4] - b[5]) * (b[4] - a[5]) > 0;
}
#define STK 64
typedef struct { int32_t node; int q; int32_t st[STK]; int sp; } Lane;

/* one lock-step phase 2 over the lanes' initial stacks; wide != 0: a step expands two binary levels */
static uint64_t phase2(Lane *L, int nl, int n, const int32_t *left, const int32_t *right, const double *boxes, int wide, uint64_t *visits, uint64_t *tests)
{
    uint64_t steps = 0;
    for (int l = 0; l < nl; ++l) { L[l].node = -1; if (L[l].sp) L[l].node = L[l].st[--L[l].sp]; }
    for (;;) {
        /* work sharing: idle lanes (>= 16 of them) take the top of busy lanes' stacks, with the donor's query */
        int idle = 0, don = 0;
        for (int l = 0; l < 64; ++l) { if (l >= nl || L[l].node < 0) ++idle; else if (L[l].sp > 0) ++don; }
        if (don && idle >= 16) {
            int di = 0;
            for (int l = 0; l < nl && don; ++l) {
                if (L[l].node >= 0) continue;
                while (di < nl && !(L[di].node >= 0 && L[di].sp > 0)) ++di;
                if (di >= nl) break;
                L[l].node = L[di].st[--L[di].sp]; L[l].q = L[di].q; L[l].sp = 0; --don; ++di;
            }
        }
        int any = 0;
        for (int l = 0; l < nl; ++l) if (L[l].node >= 0) any = 1;
        if (!any) break;
        ++steps;
        for (int l = 0; l < nl; ++l) {
            if (L[l].node < 0) continue;
            ++*visits;
            const double *qb = boxes + 6 * (size_t)((n - 1) + L[l].q);
            int32_t hit[4]; int nh = 0;
            const int32_t c[2] = { left[L[l].node], right[L[l].node] };
            for (int k = 0; k < 2; ++k) {
                ++*tests;
                if (!ov(qb, boxes + 6 * (size_t)c[k])) continue;
                if (c[k] >= n - 1) continue;                                  /* a leaf: a candidate, no further step */
                if (!wide) { hit[nh++] = c[k]; continue; }
                const int32_t g[2] = { left[c[k]], right[c[k]] };            /* wide: the child's children are in the same record */
                for (int m = 0; m < 2; ++m) { ++*tests; if (ov(qb, boxes + 6 * (size_t)g[m]) && g[m] < n - 1) hit[nh++] = g[m]; }
            }
            L[l].node = -1;
            for (int k = nh - 1; k >= 1; --k) if (L[l].sp < STK) L[l].st[L[l].sp++] = hit[k];
            if (nh) L[l].node = hit[0];
            else if (L[l].sp) L[l].node = L[l].st[--L[l].sp];
        }
    }
    return steps;
}

void wide_sim(int n, const int32_t *left, const int32_t *right, const int32_t *rl, const double *boxes, uint64_t *out)
{
    int32_t *node_of = malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n - 1; ++i) { int l = left[i]; int sp = l >= n - 1 ? l - (n - 1) : rl[l]; node_of[sp] = i; }
    memset(out, 0, sizeof(uint64_t) * 16);
    Lane *A = malloc(sizeof(Lane) * 64), *B = malloc(sizeof(Lane) * 64);
    for (int g0 = 0; g0 < n; g0 += 64) {
        const int g_last = g0 + 63 < n - 1 ? g0 + 63 : n - 1, nl = g_last - g0 + 1;
        uint64_t s1a = 0;
        for (int l = 0; l < nl; ++l) { A[l].sp = 0; A[l].q = g0 + l; }
        /* phase 1a + 1b: which internal siblings each lane hits (the stacks phase 2 starts from) */
        for (int l = 0; l < nl; ++l) {
            const int j = g0 + l;
            const double *qb = boxes + 6 * (size_t)((n - 1) + j);
            uint64_t hops = 0;
            int s = j;
            while (s < n - 1) {
                const int i = node_of[s], c = right[i];
                if (s < g_last) ++hops;
                ++out[7]; ++out[10];
                if (c < n - 1 && ov(qb, boxes + 6 * (size_t)c) && A[l].sp < STK) A[l].st[A[l].sp++] = c;
                s = rl[i];
            }
            if (hops > s1a) s1a = hops;
        }
        uint64_t s1b = 0;
        for (int t = g_last; t < n - 1; t = rl[node_of[t]]) ++s1b;
        memcpy(B, A, sizeof(Lane) * 64);
        uint64_t v2 = 0, t2 = 0, v2w = 0, t2w = 0;
        const uint64_t p2 = phase2(A, nl, n, left, right, boxes, 0, &v2, &t2);
        const uint64_t p2w = phase2(B, nl, n, left, right, boxes, 1, &v2w, &t2w);
        out[0]++; out[1] += s1a; out[2] += s1b; out[3] += p2; out[4] += p2w; out[5] += (s1a + 1) / 2; out[6] += (s1b + 1) / 2;
        out[8] += v2; out[9] += v2w; out[10] += t2; out[11] += t2w;
        if (p2 > out[12]) out[12] = p2;
        if (p2w > out[13]) out[13] = p2w;
    }
    out[11] += out[7];                                                         /* (the chain's box tests are the same either way) */
    free(A); free(B); free(node_of);
}
