/* Experiment (CPU model, not product): the half traversal with phase 2 split off.  The chain kernel (phases 0 / 1a / 1b) writes one
 * (query, subtree) ITEM per internal sibling a query's chain hits; a second kernel works the item list off, 64 items per wave at a
 * time, every idle lane taking the next item of its wave's chunk (no barrier, no workgroup coupling).
 * Tree = the oracle's (orc_build_hierarchy + orc_refit), boxes FP64 (the device descends conservative fp32 boxes: a few more visits).
 *   out[0] items, [1] phase-2 visits, [2] max visits of one item, [3] sum over TODAY's waves (64 consecutive queries, private depth-first
 *   descent, no sharing) of the wave's step count = max over lanes of the lane's visits, [4] histogram base (out[8 + k] = items with k
 *   visits, k < 64, last bin open).  pool_chunks(): wave-steps of the item kernel for a chunk size C. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static int ov(const double *a, const double *b)
{
    return (a[0] - b[1]) * (b[0] - a[1]) > 0 && (a[2] - b[3]) * (b[2] - a[3]) > 0 && (a[4] - b[5]) * (b[4] - a[5]) > 0;
}
/* item_visits: caller-allocated, capacity cap; returns the item count (the first min(count, cap) are written), in production order
 * (wave by wave, lane by lane, hop by hop) */
uint64_t pool_items(int n, const int32_t *left, const int32_t *right, const int32_t *rl, const double *boxes,
                    uint32_t *item_visits, uint64_t cap, uint64_t *out)
{
    int32_t *node_of = malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n - 1; ++i) { int l = left[i]; int sp = l >= n - 1 ? l - (n - 1) : rl[l]; node_of[sp] = i; }
    memset(out, 0, sizeof(uint64_t) * 80);
    int32_t stack[256];
    uint64_t cnt = 0;
    for (int g0 = 0; g0 < n; g0 += 64) {
        int g_last = g0 + 63 < n - 1 ? g0 + 63 : n - 1;
        uint64_t wave_max = 0;
        for (int j = g0; j <= g_last; ++j) {
            const double *qb = boxes + 6 * (size_t)((n - 1) + j);
            uint64_t lane = 0;
            int s = j;
            while (s < n - 1) {
                int i = node_of[s], c = right[i];
                if (c < n - 1 && ov(qb, boxes + 6 * (size_t)c)) {
                    uint32_t v = 0;
                    int sp = 0; stack[sp++] = c;
                    while (sp) {
                        int nd = stack[--sp]; ++v;
                        int cl = left[nd], cr = right[nd];
                        if (cl < n - 1 && ov(qb, boxes + 6 * (size_t)cl)) stack[sp++] = cl;
                        if (cr < n - 1 && ov(qb, boxes + 6 * (size_t)cr)) stack[sp++] = cr;
                    }
                    if (cnt < cap) item_visits[cnt] = v;
                    ++cnt; out[1] += v; lane += v;
                    if (v > out[2]) out[2] = v;
                    out[8 + (v < 63 ? v : 63)]++;
                }
                s = rl[i];
            }
            if (lane > wave_max) wave_max = lane;
        }
        out[3] += wave_max;
    }
    out[0] = cnt;
    free(node_of);
    return cnt;
}
/* The item kernel: wave w owns items [w C, (w + 1) C); a lane that is idle takes the chunk's next item; a lane works an item off in
 * `visits` steps (private depth-first descent).  share != 0: busy lanes hand pending subtrees to idle lanes -- modelled as its ideal,
 * steps = max(ceil(sum / 64), longest item's remaining path is ignored) .. the truth lies between the two.
 * out[0] waves, [1] sum of wave-steps, [2] max wave-steps, [3] sum of visits */
void pool_chunks(const uint32_t *item_visits, uint64_t n_items, uint32_t C, int share, uint64_t *out)
{
    memset(out, 0, sizeof(uint64_t) * 4);
    for (uint64_t b = 0; b < n_items; b += C) {
        uint64_t e = b + C < n_items ? b + C : n_items, steps = 0, sum = 0;
        if (share) { for (uint64_t k = b; k < e; ++k) sum += item_visits[k]; steps = (sum + 63) / 64; }
        else {
            uint32_t lane[64]; memset(lane, 0, sizeof lane);
            uint64_t next = b; int busy = 0;
            for (;;) {
                for (int l = 0; l < 64; ++l) if (lane[l] == 0 && next < e) { lane[l] = item_visits[next++]; sum += lane[l]; }
                busy = 0;
                for (int l = 0; l < 64; ++l) if (lane[l]) { --lane[l]; busy = 1; }
                if (!busy) break;
                ++steps;
            }
        }
        out[0]++; out[1] += steps; if (steps > out[2]) out[2] = steps; out[3] += sum;
    }
}
