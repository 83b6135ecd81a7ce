/*
 * cd_oracle.c -- CPU ORACLE for the CollisionDetection hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference algorithm (Asichurter/GPU-Computing-Course,
 * CollisionDetection/).  It is the *checker*: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The shipped product (libmi355cd.so) never links,
 * loads or calls anything in this directory.
 *
 * Pinning status.  What of the reference compiles in this image WITHOUT stand-in headers is compiled, unmodified,
 * into oracle/_ref/ (oracle/Makefile) and its outputs are committed as fixtures the functions below must reproduce bit for bit
 * (tests/test_oracle_pins.py):
 *   - morton.h (plain host C++) -> libref_morton.so -> tests/golden/morton_ref.npz: orc_expand64, orc_morton3d, orc_centroid_morton
 *     and the keys orc_sort_by_key orders (2^17 expand inputs, 2^20 in-frame points, the 10^6 centroids of BASELINE config 3);
 *   - tri_contact.cuh, box.cuh, triangle.cuh, vec3f.cuh, mathop.cuh (g++ against the genuine <cuda_runtime.h> the image ships in its
 *     triton wheel) -> libref_contact.so -> tests/golden/contact_ref.npz: orc_tri_contact (1 179 648 pairs, seven families),
 *     orc_tri_contact_helper, orc_neighbor_count, box_set, box_merge, orc_box_overlap (2^20 pairs), project3 / project6 / cross / dot,
 *     AND the end result of orc_self_collide -- pair set + pairs tested -- on BASELINE config 2 (reference side: plain O(N^2) over its
 *     predicates), config 3, the 1 M soup and the full-double cloth;
 *   - delta / determineRange / findSplit / hierarchy, the internal boxes of the refit and the verifier counters have NO reference-held
 *     vector (bvh.cuh / collision.cuh / check.cuh need device intrinsics -> stand-ins -> not built).  They cannot change the pair set
 *     (a leaf is reached iff its own box overlaps the query's) and are pinned by (i) the reference's known-answer inputs
 *     (check.cuh:19-27), (ii) outputs SURVEY.md recorded when the survey ran the reference sources, (iii) an independent O(N^2).
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fPIC -shared (no FMA contraction: every decision
 * below is an FP64 compare whose operands must round exactly like the reference's host twin).
 *
 * Node numbering (index based, replaces the reference's pointer-linked Node, bvh.cuh:25-43):
 *   internal node i  -> id i            (0 .. n-2), root = 0        (bvh.cuh:162 "Node 0 is the root")
 *   leaf j           -> id (n-1) + j    (j = 0 .. n-1, Morton-sorted order)
 * Box layout: {x1,x2,y1,y2,z1,z2} exactly as box.cuh:9.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ mathop.cuh:17-44 */
static inline double fmax2_(double a, double b) { return (a > b) ? a : b; }   /* mathop.cuh:17-19 */
static inline double fmin2_(double a, double b) { return (a < b) ? a : b; }   /* mathop.cuh:21-23 */
static inline double fmax3_(double a, double b, double c)                      /* mathop.cuh:30-36 */
{ double t = a; if (b > t) t = b; if (c > t) t = c; return t; }
static inline double fmin3_(double a, double b, double c)                      /* mathop.cuh:38-44 */
{ double t = a; if (b < t) t = b; if (c < t) t = c; return t; }

/* ------------------------------------------------------------------ morton.h:7-29 */
uint64_t orc_expand64(uint64_t v)
{
    v &= 0x1fffffULL;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8)  & 0x100f00f00f00f00fULL;
    v = (v | v << 4)  & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2)  & 0x1249249249249249ULL;
    return v;
}

/* The reference's hard-coded normalisation frame, morton.h:43-58. */
const double ORC_REF_OFF[3]  = { 0.004501, -0.476622, -0.381965 };
const double ORC_REF_SPAN[3] = { 3.08, 0.76, 2.36 };

/* double -> u64 as the reference does implicitly at morton.h:80-82.  Negative / NaN inputs are
 * undefined behaviour in the reference (its assert is compiled out); this project defines them
 * as 0 and values >= 2^63 as 2^63-1 -- the GPU path uses the same rule. */
static inline uint64_t d2u64(double e)
{
    if (!(e > 0.0)) return 0;
    if (e >= 9223372036854775808.0) return 0x7fffffffffffffffULL;
    return (uint64_t)e;
}

/* morton.h:70-89 with the frame made a parameter.  (x - off)/span * 2^20, truncate, interleave. */
uint64_t orc_morton3d(double x, double y, double z, const double off[3], const double span[3])
{
    const unsigned int scale = 1048576;
    double ex = ((x - off[0]) / span[0]) * scale;
    double ey = ((y - off[1]) / span[1]) * scale;
    double ez = ((z - off[2]) / span[2]) * scale;
    uint64_t xx = orc_expand64(d2u64(ex));
    uint64_t yy = orc_expand64(d2u64(ey));
    uint64_t zz = orc_expand64(d2u64(ez));
    return (xx << 2) | (yy << 1) | zz;
}

/* morton3D / expand64Bits over arrays (the reference-compiled fixture tests/golden/morton_ref.npz is replayed through these) */
void orc_morton3d_batch(const double *xyz, uint64_t n, const double off[3], const double span[3], uint64_t *keys)
{
    for (uint64_t i = 0; i < n; ++i) keys[i] = orc_morton3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], off, span);
}
void orc_expand64_batch(const uint64_t *v, uint64_t n, uint64_t *out)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = orc_expand64(v[i]);
}

/* ------------------------------------------------------------------ the ADAPTIVE frame (CD_FRAME_AUTO since round 6)
 * NOT reference behaviour: the reference has one frame, the constants of morton.h:43-58, and interleaves 20 bits an axis x, y, z
 * (the functions above; CD_FRAME_REFERENCE / CD_FRAME_CUSTOM).  For a mesh those constants do not fit the library derives a frame
 * from the mesh, and -- SURVEY section 7, "key freedom": the pair set does not depend on the keys -- deals the 60 key bits to the
 * axes so that the cells of every level of the tree are as near to cubes as powers of two allow, IN UNITS OF THE TRIANGLES' OWN
 * EXTENT along each axis:
 *   a box query of size s meets a cell of length L with probability ~ (L + s); halving the cell along an axis costs
 *   (L + 2 s) / (L + s) -- least along the axis with the largest L / s, not the largest L.  A cloth is thin along one axis AND so
 *   are its triangles: its sheets lie on top of each other there, and what separates them is worth a split early.
 * Statistic: per axis the mean of log2(box extent) over the triangles whose box is not flat on that axis, in fixed point
 * (flog2_: 8 fraction bits, piecewise linear) and summed as INTEGERS -- any order of summation gives the same sums, so device
 * and oracle agree bit for bit.  An axis whose triangles are thinner than those of the axis where they are largest counts as
 * longer by that ratio, at most 2^ORC_LAYOUT_CAP (an axis on which every box is flat: the cap): E[a] = flog2(extent of the
 * centroids) + min(Lref - Lmean[a], cap).  Then
 *   axes ordered by E, A >= B >= C (ties: the lower axis first);
 *   nA  = round(E[A] - E[B]) leading bits split A alone;  nAB = round(E[B] - E[C]) pairs of bits (A, B) follow;
 *   nABC triples (A, B, C) take the rest of the 60 bits; one or two bits left over go to nA / nAB.
 * Per-axis normalisation with the fixed x, y, z interleave (round 5) made cells of 400 : 1 on a thin, long mesh: 44 node visits a
 * query where this layout walks 29 (tools/sim/frame_study.py).  An isotropic frame walks ~33 and leaves the thin axes' leading key
 * bits constant -- the sort's 16 global bits would hold 10 that vary.  Here all 60 vary.
 * This restatement is what the device code (cd_math.h: frame_layout, morton3d_layout) is checked against.
 * layout word: bit 63 set | A | B << 2 | C << 4 | nA << 8 | nAB << 16 | nABC << 24;  0 = the reference's interleave. */
#define ORC_LAYOUT_VALID (1ull << 63)
#define ORC_LAYOUT_CAP 4
/* 256 log2(x), piecewise linear between powers of two; x > 0 and normal */
static inline int64_t flog2_(double x)
{
    uint64_t u; memcpy(&u, &x, sizeof u);
    return (((int64_t)((u >> 52) & 0x7ff) - 1023) << 8) + (int64_t)((u >> 44) & 0xff);
}
#define ORC_FLOG_MIN 1e-300       /* below this an extent counts as 0 (no subnormals in the statistic) */
/* the statistic of one triangle box: adds flog2(extent) to sum[a] and 1 to cnt[a] for every axis on which the box is not flat */
static inline void layout_stat_(const double *p1, const double *p2, const double *p3, int64_t sum[3], int64_t cnt[3])
{
    for (int a = 0; a < 3; ++a) {
        const double e = fmax3_(p1[a], p2[a], p3[a]) - fmin3_(p1[a], p2[a], p3[a]);
        if (e > ORC_FLOG_MIN) { sum[a] += flog2_(e); cnt[a] += 1; }
    }
}
/* (cap_bits: ORC_LAYOUT_CAP in the product's rule; a parameter for tools/sim/frame_study.py, which prices the rule) */
uint64_t orc_frame_layout_cap(const double lo[3], const double hi[3], const int64_t sum[3], const int64_t cnt[3], int cap_bits)
{
    const int64_t NONE = -((int64_t)1 << 40);
    int64_t E[3], Lm[3], Lref = NONE; int ord[3] = { 0, 1, 2 };
    for (int a = 0; a < 3; ++a) {
        Lm[a] = NONE;
        if (cnt[a] > 0) { Lm[a] = (int64_t)floor((double)sum[a] / (double)cnt[a]); if (Lm[a] > Lref) Lref = Lm[a]; }   /* floor of the mean: one IEEE division of exactly represented integers */
    }
    for (int a = 0; a < 3; ++a) {
        const double e = hi[a] - lo[a];
        if (!(e > ORC_FLOG_MIN)) { E[a] = NONE; continue; }
        int64_t d = (int64_t)cap_bits << 8;
        if (Lm[a] != NONE) { d = Lref - Lm[a]; if (d > ((int64_t)cap_bits << 8)) d = (int64_t)cap_bits << 8; }
        if (Lref == NONE) d = 0;                                         /* every box flat on every axis: points */
        E[a] = flog2_(e) + d;
    }
    /* stable insertion sort, descending */
    for (int i = 1; i < 3; ++i) for (int j = i; j > 0 && E[ord[j]] > E[ord[j - 1]]; --j) { int t = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t; }
    const int64_t EA = E[ord[0]], EB = E[ord[1]], EC = E[ord[2]];
    int64_t nA = EA == NONE ? 0 : (EB == NONE ? 60 : (EA - EB + 128) >> 8);
    if (nA > 60) nA = 60;
    int64_t rem = 60 - nA;
    int64_t nAB = EB == NONE ? 0 : (EC == NONE ? 30 : (EB - EC + 128) >> 8);
    if (2 * nAB > rem) nAB = rem / 2;
    rem -= 2 * nAB;
    const int64_t nABC = rem / 3, left = rem % 3;
    if (left == 1) ++nA;
    if (left == 2) ++nAB;
    return ORC_LAYOUT_VALID | (uint64_t)ord[0] | ((uint64_t)ord[1] << 2) | ((uint64_t)ord[2] << 4) | ((uint64_t)nA << 8) | ((uint64_t)nAB << 16) | ((uint64_t)nABC << 24);
}
uint64_t orc_frame_layout(const double lo[3], const double hi[3], const int64_t sum[3], const int64_t cnt[3]) { return orc_frame_layout_cap(lo, hi, sum, cnt, ORC_LAYOUT_CAP); }
/* the statistic of a mesh (what orc_auto_frame feeds orc_frame_layout), and the centroids' bounds */
void orc_layout_stat(const double *verts, const uint32_t *vidx, uint32_t n, double lo[3], double hi[3], int64_t sum[3], int64_t cnt[3])
{
    for (int a = 0; a < 3; ++a) { lo[a] = 1e300; hi[a] = -1e300; sum[a] = 0; cnt[a] = 0; }
    for (uint32_t t = 0; t < n; ++t) {
        const double *p1 = verts + 3 * (size_t)vidx[3 * t + 0], *p2 = verts + 3 * (size_t)vidx[3 * t + 1], *p3 = verts + 3 * (size_t)vidx[3 * t + 2];
        for (int a = 0; a < 3; ++a) { const double c = (p1[a] + p2[a] + p3[a]) / 3; if (c < lo[a]) lo[a] = c; if (c > hi[a]) hi[a] = c; }
        layout_stat_(p1, p2, p3, sum, cnt);
    }
}
/* spread the low 32 bits to the even positions */
static inline uint64_t expand2_(uint64_t v)
{
    v &= 0xffffffffULL;
    v = (v | v << 16) & 0x0000ffff0000ffffULL;
    v = (v | v << 8)  & 0x00ff00ff00ff00ffULL;
    v = (v | v << 4)  & 0x0f0f0f0f0f0f0f0fULL;
    v = (v | v << 2)  & 0x3333333333333333ULL;
    v = (v | v << 1)  & 0x5555555555555555ULL;
    return v;
}
/* The cell of a triangle along an axis, in a frame WITH a layout:  floor(((p1 + p2 + p3) - 3 off) * (2^bits / (3 span)))  clamped to [0, 2^bits - 1] -- the vertex
 * SUM s against thrice the offset, times one factor per axis and frame: no division per key (the device's reason: cd_math.h).  Every operation is one IEEE operation
 * on both sides (3.0 * off, 3.0 * span, 2^bits / that, s - that, times that). */
static inline uint64_t cell_(double s, double off, double span, int bits)
{
    if (bits == 0) return 0;
    const double off3 = 3.0 * off, k = (double)(1ull << bits) / (3.0 * span);
    const uint64_t v = d2u64((s - off3) * k), top = (1ull << bits) - 1;
    return v > top ? top : v;                                          /* a centroid beyond the frame takes the last cell: the key stays below 2^60 */
}
/* sx, sy, sz: p1 + p2 + p3 per axis, in that order.  layout 0: x, y, z are taken as the centroid itself and the key is morton.h:70-89's (orc_morton3d). */
uint64_t orc_morton3d_layout(double x, double y, double z, const double off[3], const double span[3], uint64_t layout)
{
    if (!(layout & ORC_LAYOUT_VALID)) return orc_morton3d(x, y, z, off, span);
    const double c[3] = { x, y, z };
    const int A = (int)(layout & 3), B = (int)((layout >> 2) & 3), C = (int)((layout >> 4) & 3);
    const int nA = (int)((layout >> 8) & 255), p = (int)((layout >> 16) & 255), t = (int)((layout >> 24) & 255);
    const uint64_t ia = cell_(c[A], off[A], span[A], nA + p + t), ib = cell_(c[B], off[B], span[B], p + t), ic = cell_(c[C], off[C], span[C], t);
    const uint64_t mt = (1ull << t) - 1, mp = (1ull << p) - 1;
    const uint64_t triples = (orc_expand64(ia & mt) << 2) | (orc_expand64(ib & mt) << 1) | orc_expand64(ic & mt);
    const uint64_t pairs = (expand2_((ia >> t) & mp) << 1) | expand2_((ib >> t) & mp);
    const uint64_t top = (nA + p + t) >= 64 ? 0 : (ia >> (p + t));
    return (2 * p + 3 * t >= 64 ? 0 : (top << (2 * p + 3 * t))) | (pairs << (3 * t)) | triples;
}
/* The whole AUTO frame as the device forms it: bounds of the centroids, span widened by 2^-20 (span 1 where the extent is 0), the statistic, layout.
 * frame: off[3], span[3]; returns the layout word. */
uint64_t orc_auto_frame(const double *verts, const uint32_t *vidx, uint32_t n, double frame[6])
{
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    int64_t sum[3] = { 0, 0, 0 }, cnt[3] = { 0, 0, 0 };
    for (uint32_t t = 0; t < n; ++t) {
        const double *p1 = verts + 3 * (size_t)vidx[3 * t + 0], *p2 = verts + 3 * (size_t)vidx[3 * t + 1], *p3 = verts + 3 * (size_t)vidx[3 * t + 2];
        for (int a = 0; a < 3; ++a) { const double c = (p1[a] + p2[a] + p3[a]) / 3; if (c < lo[a]) lo[a] = c; if (c > hi[a]) hi[a] = c; }
        layout_stat_(p1, p2, p3, sum, cnt);
    }
    for (int a = 0; a < 3; ++a) {
        double span = (hi[a] - lo[a]) * (1.0 + 1.0 / 1048576.0);
        if (!(span > 0.0)) span = 1.0;
        frame[a] = lo[a]; frame[3 + a] = span;
    }
    return orc_frame_layout(lo, hi, sum, cnt);
}
void orc_centroid_morton_layout(const double *verts, const uint32_t *vidx, uint32_t n, const double off[3], const double span[3], uint64_t layout, uint64_t *keys)
{
    for (uint32_t t = 0; t < n; ++t) {
        const double *p1 = verts + 3 * (size_t)vidx[3 * t + 0], *p2 = verts + 3 * (size_t)vidx[3 * t + 1], *p3 = verts + 3 * (size_t)vidx[3 * t + 2];
        if (layout & ORC_LAYOUT_VALID) keys[t] = orc_morton3d_layout(p1[0] + p2[0] + p3[0], p1[1] + p2[1] + p3[1], p1[2] + p2[2] + p3[2], off, span, layout);
        else keys[t] = orc_morton3d((p1[0] + p2[0] + p3[0]) / 3, (p1[1] + p2[1] + p3[1]) / 3, (p1[2] + p2[2] + p3[2]) / 3, off, span);
    }
}
void orc_morton3d_layout_batch(const double *xyz, uint64_t n, const double off[3], const double span[3], uint64_t layout, uint64_t *keys)
{
    for (uint64_t i = 0; i < n; ++i) keys[i] = orc_morton3d_layout(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], off, span, layout);
}

/* load_obj.h:89-101: centroid = (p1+p2+p3)/3 per axis, then morton3D.  centroids may be NULL. */
void orc_centroid_morton(const double *verts, const uint32_t *vidx, uint32_t n,
                         const double off[3], const double span[3],
                         uint64_t *keys, double *centroids)
{
    for (uint32_t t = 0; t < n; ++t) {
        const double *p1 = verts + 3 * (size_t)vidx[3 * t + 0];
        const double *p2 = verts + 3 * (size_t)vidx[3 * t + 1];
        const double *p3 = verts + 3 * (size_t)vidx[3 * t + 2];
        double ax = (p1[0] + p2[0] + p3[0]) / 3;
        double ay = (p1[1] + p2[1] + p3[1]) / 3;
        double az = (p1[2] + p2[2] + p3[2]) / 3;
        keys[t] = orc_morton3d(ax, ay, az, off, span);
        if (centroids) { centroids[3 * t] = ax; centroids[3 * t + 1] = ay; centroids[3 * t + 2] = az; }
    }
}

/* load_obj.h:107 thrust::sort_by_key(mortons, triangles) on host iterators: a stable ascending
 * sort by key.  LSD radix, 8 x 8 bits; perm[i] = original index of the i-th smallest key. */
void orc_sort_by_key(uint64_t *keys, uint32_t *perm, uint32_t n)
{
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    uint32_t *p2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) perm[i] = i;
    uint64_t *ks = keys, *kd = k2; uint32_t *ps = perm, *pd = p2;
    for (int pass = 0; pass < 8; ++pass) {
        size_t cnt[257]; memset(cnt, 0, sizeof cnt);
        int sh = pass * 8;
        for (uint32_t i = 0; i < n; ++i) cnt[((ks[i] >> sh) & 255) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (uint32_t i = 0; i < n; ++i) { size_t o = cnt[(ks[i] >> sh) & 255]++; kd[o] = ks[i]; pd[o] = ps[i]; }
        uint64_t *tk = ks; ks = kd; kd = tk; uint32_t *tp = ps; ps = pd; pd = tp;
    }
    /* 8 passes: result is back in keys/perm */
    free(k2); free(p2);
}

/* ------------------------------------------------------------------ bvh.cuh:48 */
/* CUDA __clzll(0) == 64. */
int orc_clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }

/* delta macro, bvh.cuh:48.  tiebreak == 0: literal reference (equal keys -> 64, no index
 * tie-break, the reference's duplicate-key defect).  tiebreak != 0: equal keys compare by index
 * (64 + clz32(i^j)), the standard Karras fix; identical to the literal form on unique keys. */
static inline int delta_(const uint64_t *keys, int n, int i, int j, int tiebreak)
{
    if (!(j >= 0 && j < n)) return -1;
    uint64_t x = keys[i] ^ keys[j];
    if (x || !tiebreak) return orc_clz64(x);
    uint32_t y = (uint32_t)i ^ (uint32_t)j;
    return 64 + (y ? __builtin_clz(y) : 32);
}
int orc_delta(const uint64_t *keys, int n, int i, int j, int tiebreak) { return delta_(keys, n, i, j, tiebreak); }

/* findSplit, bvh.cuh:57-98 (cpu.cuh:22-63 is the same code). */
int orc_find_split(const uint64_t *keys, int n, int first, int last, int tiebreak)
{
    uint64_t firstCode = keys[first], lastCode = keys[last];
    if (!tiebreak) {
        if (firstCode == lastCode) return (first + last) >> 1;             /* bvh.cuh:66-67 */
    }
    int commonPrefix = delta_(keys, n, first, last, tiebreak);             /* bvh.cuh:72 */
    int split = first;
    int step = last - first;
    do {
        step = (step + 1) >> 1;
        int newSplit = split + step;
        if (newSplit < last) {
            int splitPrefix = delta_(keys, n, first, newSplit, tiebreak);  /* bvh.cuh:88-89 */
            if (splitPrefix > commonPrefix) split = newSplit;
        }
    } while (step > 1);
    return split;
}

/* determineRange, bvh.cuh:100-123 (cpu.cuh:65-87). */
void orc_determine_range(const uint64_t *keys, int n, int i, int tiebreak, int *first, int *last)
{
    int d = (delta_(keys, n, i, i + 1, tiebreak) - delta_(keys, n, i, i - 1, tiebreak)) >= 0 ? 1 : -1;
    int delta_min = delta_(keys, n, i, i - d, tiebreak);
    int mlen = 2;
    while (delta_(keys, n, i, i + mlen * d, tiebreak) > delta_min) mlen <<= 1;
    int l = 0;
    for (int t = mlen >> 1; t >= 1; t >>= 1)
        if (delta_(keys, n, i, i + (l + t) * d, tiebreak) > delta_min) l += t;
    int j = i + l * d;
    *first = i < j ? i : j;
    *last  = i < j ? j : i;
}

/* generateHierarchyParallel, bvh.cuh:146-199 (cpu.cuh:110-165), index based.
 * left/right: n-1 entries (unified node ids); parent: 2n-1 entries, -1 = NULL.
 * parent_wrong counts children whose parent was already set (bvh.cuh:192,194). */
void orc_build_hierarchy(const uint64_t *keys, int n, int tiebreak,
                         int32_t *left, int32_t *right, int32_t *parent,
                         int32_t *range_first, int32_t *range_last, uint32_t *parent_wrong)
{
    for (int i = 0; i < 2 * n - 1; ++i) parent[i] = -1;
    uint32_t wrong = 0;
    for (int idx = 0; idx < n - 1; ++idx) {
        int first, last;
        orc_determine_range(keys, n, idx, tiebreak, &first, &last);
        int split = orc_find_split(keys, n, first, last, tiebreak);
        int32_t a = (split == first)    ? (n - 1) + split       : split;        /* bvh.cuh:176-179 */
        int32_t b = (split + 1 == last) ? (n - 1) + (split + 1) : split + 1;    /* bvh.cuh:183-186 */
        left[idx] = a; right[idx] = b;
        if (parent[a] != -1) wrong++;
        parent[a] = idx;
        if (parent[b] != -1) wrong++;
        parent[b] = idx;
        if (range_first) range_first[idx] = first;
        if (range_last)  range_last[idx]  = last;
    }
    if (parent_wrong) *parent_wrong = wrong;
}

/* ------------------------------------------------------------------ box.cuh */
/* Box::set, box.cuh:13-22 */
static inline void box_set(double *b, const double *v1, const double *v2, const double *v3)
{
    b[0] = fmin3_(v1[0], v2[0], v3[0]); b[1] = fmax3_(v1[0], v2[0], v3[0]);
    b[2] = fmin3_(v1[1], v2[1], v3[1]); b[3] = fmax3_(v1[1], v2[1], v3[1]);
    b[4] = fmin3_(v1[2], v2[2], v3[2]); b[5] = fmax3_(v1[2], v2[2], v3[2]);
}
/* Box::merge, box.cuh:24-32 */
static inline void box_merge(double *o, const double *a, const double *b)
{
    o[0] = fmin2_(a[0], b[0]); o[1] = fmax2_(a[1], b[1]);
    o[2] = fmin2_(a[2], b[2]); o[3] = fmax2_(a[3], b[3]);
    o[4] = fmin2_(a[4], b[4]); o[5] = fmax2_(a[5], b[5]);
}
/* checkBoxOverlap, box.cuh:40-43: strict, product form. */
int orc_box_overlap(const double *a, const double *b)
{
    if ((a[0] - b[1]) * (b[0] - a[1]) > 0 && (a[2] - b[3]) * (b[2] - a[3]) > 0 &&
        (a[4] - b[5]) * (b[4] - a[5]) > 0) return 1;
    return 0;
}

/* calBoundingBox, bvh.cuh:258-285, sequential twin cpu.cuh:167-194.
 * perm[j] = triangle stored in leaf j.  boxes: (2n-1) x 6.  bounded: n-1 counters (must end 2).
 * child_count: 2n-1 (bvh.cuh:265,279), may be NULL. */
void orc_refit(const double *verts, const uint32_t *vidx, const uint32_t *perm, int n,
               const int32_t *left, const int32_t *right, const int32_t *parent,
               double *boxes, uint32_t *bounded, uint32_t *child_count)
{
    for (int i = 0; i < n - 1; ++i) bounded[i] = 0;
    for (int j = 0; j < n; ++j) {
        uint32_t t = perm[j];
        int node = (n - 1) + j;
        box_set(boxes + 6 * (size_t)node, verts + 3 * (size_t)vidx[3 * t], verts + 3 * (size_t)vidx[3 * t + 1],
                verts + 3 * (size_t)vidx[3 * t + 2]);
        if (child_count) child_count[node] = 1;
        int cur = parent[node];
        while (cur != -1) {
            if (bounded[cur]++ == 0) break;
            box_merge(boxes + 6 * (size_t)cur, boxes + 6 * (size_t)left[cur], boxes + 6 * (size_t)right[cur]);
            if (child_count) child_count[cur] = 1 + child_count[left[cur]] + child_count[right[cur]];
            cur = parent[cur];
        }
    }
}

/* ------------------------------------------------------------------ triangle.cuh:18-30 */
int orc_neighbor_count(const uint32_t *a, const uint32_t *b)
{
    int c = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c += (a[i] == b[j]);
    return c;
}

/* ------------------------------------------------------------------ vec3f.cuh / tri_contact.cuh */
typedef struct { double x, y, z; } v3;
static inline v3 v3sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }      /* vec3f.cuh:100-103 */
static inline v3 v3neg(v3 a) { v3 r = { -a.x, -a.y, -a.z }; return r; }                           /* vec3f.cuh:91-93 */
static inline v3 v3cross(v3 a, v3 b)                                                               /* vec3f.cuh:118-121 */
{ v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; return r; }
static inline double v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }               /* vec3f.cuh:123-125 */

/* project3, vec3f.cuh:257-270 */
static inline int project3(v3 ax, v3 p1, v3 p2, v3 p3)
{
    double P1 = v3dot(ax, p1), P2 = v3dot(ax, p2), P3 = v3dot(ax, p3);
    double mx1 = fmax3_(P1, P2, P3), mn1 = fmin3_(P1, P2, P3);
    if (mn1 > 0) return 0;
    if (0 > mx1) return 0;
    return 1;
}
/* project6, vec3f.cuh:272-291 */
static inline int project6(v3 ax, v3 p1, v3 p2, v3 p3, v3 q1, v3 q2, v3 q3)
{
    double P1 = v3dot(ax, p1), P2 = v3dot(ax, p2), P3 = v3dot(ax, p3);
    double Q1 = v3dot(ax, q1), Q2 = v3dot(ax, q2), Q3 = v3dot(ax, q3);
    double mx1 = fmax3_(P1, P2, P3), mn1 = fmin3_(P1, P2, P3);
    double mx2 = fmax3_(Q1, Q2, Q3), mn2 = fmin3_(Q1, Q2, Q3);
    if (mn1 > mx2) return 0;
    if (mn2 > mx1) return 0;
    return 1;
}

/* checkTriangleContact, tri_contact.cuh:19-78.  P*,Q*: 3 doubles each. */
int orc_tri_contact(const double *P1, const double *P2, const double *P3,
                    const double *Q1, const double *Q2, const double *Q3)
{
    v3 vP1 = { P1[0], P1[1], P1[2] }, vP2 = { P2[0], P2[1], P2[2] }, vP3 = { P3[0], P3[1], P3[2] };
    v3 vQ1 = { Q1[0], Q1[1], Q1[2] }, vQ2 = { Q2[0], Q2[1], Q2[2] }, vQ3 = { Q3[0], Q3[1], Q3[2] };
    v3 p1 = { 0, 0, 0 };                                   /* tri_contact.cuh:21 default ctor */
    v3 p2 = v3sub(vP2, vP1), p3 = v3sub(vP3, vP1);
    v3 q1 = v3sub(vQ1, vP1), q2 = v3sub(vQ2, vP1), q3 = v3sub(vQ3, vP1);
    v3 e1 = v3sub(p2, p1), e2 = v3sub(p3, p2), e3 = v3sub(p1, p3);
    v3 f1 = v3sub(q2, q1), f2 = v3sub(q3, q2), f3 = v3sub(q1, q3);
    v3 n1 = v3cross(e1, e2), m1 = v3cross(f1, f2);
    v3 g1 = v3cross(e1, n1), g2 = v3cross(e2, n1), g3 = v3cross(e3, n1);
    v3 h1 = v3cross(f1, m1), h2 = v3cross(f2, m1), h3 = v3cross(f3, m1);
    v3 ef11 = v3cross(e1, f1), ef12 = v3cross(e1, f2), ef13 = v3cross(e1, f3);
    v3 ef21 = v3cross(e2, f1), ef22 = v3cross(e2, f2), ef23 = v3cross(e2, f3);
    v3 ef31 = v3cross(e3, f1), ef32 = v3cross(e3, f2), ef33 = v3cross(e3, f3);

    if (!project3(n1, q1, q2, q3)) return 0;
    if (!project3(m1, v3neg(q1), v3sub(p2, q1), v3sub(p3, q1))) return 0;
    if (!project6(ef11, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(ef12, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(ef13, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(ef21, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(ef22, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(ef23, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(ef31, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(ef32, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(ef33, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(g1, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(g2, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(g3, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(h1, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(h2, p1, p2, p3, q1, q2, q3)) return 0;
    if (!project6(h3, p1, p2, p3, q1, q2, q3)) return 0;
    return 1;
}

/* checkTriangleContactHelper, tri_contact.cuh:80-87: ID rule, then gather 6 vertices. */
int orc_tri_contact_helper(uint32_t id_a, const uint32_t *va, uint32_t id_b, const uint32_t *vb,
                           const double *verts)
{
    if (id_a >= id_b) return 0;
    return orc_tri_contact(verts + 3 * (size_t)va[0], verts + 3 * (size_t)va[1], verts + 3 * (size_t)va[2],
                           verts + 3 * (size_t)vb[0], verts + 3 * (size_t)vb[1], verts + 3 * (size_t)vb[2]);
}

/* Batch form used to check the GPU pair-test kernel: out[k] = helper(pair k). */
void orc_tri_contact_batch(const double *verts, const uint32_t *vidx, const uint32_t *ids,
                           const uint32_t *pairs, uint64_t np, uint8_t *out)
{
    for (uint64_t k = 0; k < np; ++k) {
        uint32_t a = pairs[2 * k], b = pairs[2 * k + 1];
        uint32_t ia = ids ? ids[a] : a, ib = ids ? ids[b] : b;
        out[k] = (uint8_t)(orc_neighbor_count(vidx + 3 * (size_t)a, vidx + 3 * (size_t)b) < 1 &&
                           orc_tri_contact_helper(ia, vidx + 3 * (size_t)a, ib, vidx + 3 * (size_t)b, verts) > 0);
    }
}

/* Batch forms of the predicates above on explicit operands: what tests/test_oracle_pins.py replays against the
 * reference-compiled vectors of tests/golden/contact_ref.npz (layouts as oracle/ref_contact.cpp's entry points). */
void orc_tri_contact_points_batch(const double *tri, uint64_t n, int32_t *out)
{
    for (uint64_t k = 0; k < n; ++k) {
        const double *t = tri + 18 * k;
        out[k] = orc_tri_contact(t, t + 3, t + 6, t + 9, t + 12, t + 15);
    }
}
void orc_helper_batch(const double *verts, const uint32_t *va, const uint32_t *ida, const uint32_t *vb, const uint32_t *idb,
                      uint64_t n, int32_t *out)
{
    for (uint64_t k = 0; k < n; ++k) out[k] = orc_tri_contact_helper(ida[k], va + 3 * k, idb[k], vb + 3 * k, verts);
}
void orc_neighbor_count_batch(const uint32_t *va, const uint32_t *vb, uint64_t n, int32_t *out)
{
    for (uint64_t k = 0; k < n; ++k) out[k] = orc_neighbor_count(va + 3 * k, vb + 3 * k);
}
void orc_box_set_batch(const double *verts, const uint32_t *vidx, uint64_t n, double *out)
{
    for (uint64_t k = 0; k < n; ++k)
        box_set(out + 6 * k, verts + 3 * (size_t)vidx[3 * k], verts + 3 * (size_t)vidx[3 * k + 1], verts + 3 * (size_t)vidx[3 * k + 2]);
}
void orc_box_merge_batch(const double *a, const double *b, uint64_t n, double *out)
{
    for (uint64_t k = 0; k < n; ++k) box_merge(out + 6 * k, a + 6 * k, b + 6 * k);
}
void orc_box_overlap_batch(const double *a, const double *b, uint64_t n, int32_t *out)
{
    for (uint64_t k = 0; k < n; ++k) out[k] = orc_box_overlap(a + 6 * k, b + 6 * k);
}
static inline v3 v3at(const double *p) { v3 r = { p[0], p[1], p[2] }; return r; }
void orc_project3_batch(const double *v, uint64_t n, int32_t *out)
{
    for (uint64_t k = 0; k < n; ++k) { const double *t = v + 12 * k; out[k] = project3(v3at(t), v3at(t + 3), v3at(t + 6), v3at(t + 9)); }
}
void orc_project6_batch(const double *v, uint64_t n, int32_t *out)
{
    for (uint64_t k = 0; k < n; ++k) {
        const double *t = v + 21 * k;
        out[k] = project6(v3at(t), v3at(t + 3), v3at(t + 6), v3at(t + 9), v3at(t + 12), v3at(t + 15), v3at(t + 18));
    }
}
void orc_cross_dot_batch(const double *v, uint64_t n, double *cross_out, double *dot_out)
{
    for (uint64_t k = 0; k < n; ++k) {
        v3 a = v3at(v + 6 * k), b = v3at(v + 6 * k + 3), c = v3cross(a, b);
        cross_out[3 * k] = c.x; cross_out[3 * k + 1] = c.y; cross_out[3 * k + 2] = c.z;
        dot_out[k] = v3dot(a, b);
    }
}

/* ------------------------------------------------------------------ collision.cuh:19-88 */
typedef struct {
    uint64_t n_pairs;        /* contacts found (count, collision.cuh:40)                      */
    uint64_t pairs_tested;   /* (query, leaf) pairs whose AABBs strictly overlap  */
    uint64_t node_visits;    /* internal nodes popped                                         */
    uint32_t max_stack;      /* deepest stack pointer reached (reference stack is 32)         */
    uint32_t overflow;       /* pairs beyond cap (not written)                                */
} orc_stats;

#define ORC_STACK 256

/* findCollisionIterative for one external query (collision.cuh:19-71).
 * q_id/q_vidx/q_box describe the query triangle; tree triangles are perm/ids/vidx. */
static void traverse_one(uint32_t q_id, const uint32_t *q_vidx, const double *q_box,
                         const double *verts, const uint32_t *vidx, const uint32_t *ids,
                         const uint32_t *perm, int n,
                         const int32_t *left, const int32_t *right, const double *boxes,
                         uint32_t *pairs, uint64_t cap, orc_stats *st)
{
    int32_t stack[ORC_STACK];
    unsigned sptr = 0;
    stack[sptr++] = -1;
    int32_t node = 0;                                           /* root = internal[0], main.cu:142 */
    if (n == 1) {                                               /* degenerate: a single leaf, no internal node */
        node = -1;
    }
    while (node != -1) {
        st->node_visits++;
        int32_t ch[2] = { left[node], right[node] };
        int ov[2] = { orc_box_overlap(q_box, boxes + 6 * (size_t)ch[0]),
                      orc_box_overlap(q_box, boxes + 6 * (size_t)ch[1]) };
        for (int s = 0; s < 2; ++s) {                           /* L then R, collision.cuh:35,52 */
            if (ov[s] > 0) {
                if (ch[s] >= n - 1) {                           /* isLeaf */
                    st->pairs_tested++;
                    uint32_t t = perm[ch[s] - (n - 1)];
                    const uint32_t *tv = vidx + 3 * (size_t)t;
                    if (orc_neighbor_count(q_vidx, tv) < 1) {
                        uint32_t tid = ids ? ids[t] : t;
                        if (orc_tri_contact_helper(q_id, q_vidx, tid, tv, verts) > 0) {
                            uint64_t cur = st->n_pairs++;
                            if (cur < cap) { pairs[2 * cur] = q_id; pairs[2 * cur + 1] = tid; }
                            else st->overflow++;
                        }
                    }
                } else {
                    if (sptr < ORC_STACK) stack[sptr++] = ch[s];
                    if (sptr > st->max_stack) st->max_stack = sptr;
                }
            }
        }
        node = stack[--sptr];
    }
}

/* findCollisions, collision.cuh:73-88 / cpu.cuh:247-271: every leaf queries the whole tree. */
void orc_find_collisions(const double *verts, const uint32_t *vidx, const uint32_t *ids,
                         const uint32_t *perm, int n,
                         const int32_t *left, const int32_t *right, const double *boxes,
                         uint32_t *pairs, uint64_t cap, orc_stats *st)
{
    memset(st, 0, sizeof *st);
    for (int j = 0; j < n; ++j) {
        uint32_t t = perm[j];
        traverse_one(ids ? ids[t] : t, vidx + 3 * (size_t)t, boxes + 6 * (size_t)((n - 1) + j),
                     verts, vidx, ids, perm, n, left, right, boxes, pairs, cap, st);
    }
}

/* External queries against a tree (the cross-rank pass of SURVEY 8e uses the same traversal):
 * q_verts: nq x 9 doubles (three vertices), q_vidx: nq x 3 (global vertex ids), q_ids: nq. */
void orc_find_collisions_queries(const double *q_verts, const uint32_t *q_vidx, const uint32_t *q_ids, int nq,
                                 const double *verts, const uint32_t *vidx, const uint32_t *ids,
                                 const uint32_t *perm, int n,
                                 const int32_t *left, const int32_t *right, const double *boxes,
                                 uint32_t vbase, uint32_t *pairs, uint64_t cap, orc_stats *st)
{
    memset(st, 0, sizeof *st);
    /* vbase: global id of local vertex 0 -- the queries carry GLOBAL vertex ids (they come from another
     * rank), so the neighbour filter compares them with local index + vbase. */
    /* The exact test reads vertices through indices into one array (tri_contact.cuh:83-84); give the
     * query its three vertices through a private 3-entry array and local indices. */
    for (int q = 0; q < nq; ++q) {
        double qb[6];
        box_set(qb, q_verts + 9 * (size_t)q, q_verts + 9 * (size_t)q + 3, q_verts + 9 * (size_t)q + 6);
        int32_t stack[ORC_STACK]; unsigned sptr = 0; stack[sptr++] = -1;
        int32_t node = (n == 1) ? -1 : 0;
        while (node != -1) {
            st->node_visits++;
            int32_t ch[2] = { left[node], right[node] };
            int ov[2] = { orc_box_overlap(qb, boxes + 6 * (size_t)ch[0]), orc_box_overlap(qb, boxes + 6 * (size_t)ch[1]) };
            for (int s = 0; s < 2; ++s) if (ov[s] > 0) {
                if (ch[s] >= n - 1) {
                    st->pairs_tested++;
                    uint32_t t = perm[ch[s] - (n - 1)];
                    const uint32_t *tv = vidx + 3 * (size_t)t;
                    const uint32_t tvg[3] = { tv[0] + vbase, tv[1] + vbase, tv[2] + vbase };
                    if (orc_neighbor_count(q_vidx + 3 * (size_t)q, tvg) < 1) {
                        uint32_t tid = ids ? ids[t] : t;
                        if (q_ids[q] < tid &&
                            orc_tri_contact(q_verts + 9 * (size_t)q, q_verts + 9 * (size_t)q + 3, q_verts + 9 * (size_t)q + 6,
                                            verts + 3 * (size_t)tv[0], verts + 3 * (size_t)tv[1], verts + 3 * (size_t)tv[2]) > 0) {
                            uint64_t cur = st->n_pairs++;
                            if (cur < cap) { pairs[2 * cur] = q_ids[q]; pairs[2 * cur + 1] = tid; }
                            else st->overflow++;
                        }
                    }
                } else {
                    if (sptr < ORC_STACK) stack[sptr++] = ch[s];
                    if (sptr > st->max_stack) st->max_stack = sptr;
                }
            }
            node = stack[--sptr];
        }
    }
}

/* checkDirectComp, check.cuh:117-141: O(N^2) all-pairs, no tree.  box_filter != 0 additionally
 * requires the strict leaf-AABB overlap that the BVH path applies implicitly (collision.cuh:31-36),
 * which is the set the traversal must reproduce; box_filter == 0 is the literal check.cuh count. */
uint64_t orc_brute_force(const double *verts, const uint32_t *vidx, const uint32_t *ids, int n,
                         int box_filter, uint32_t *pairs, uint64_t cap, uint64_t *tested)
{
    double *bx = (double *)malloc(sizeof(double) * 6 * (size_t)n);
    for (int i = 0; i < n; ++i)
        box_set(bx + 6 * (size_t)i, verts + 3 * (size_t)vidx[3 * i], verts + 3 * (size_t)vidx[3 * i + 1],
                verts + 3 * (size_t)vidx[3 * i + 2]);
    uint64_t cnt = 0, tst = 0;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            if (box_filter && !orc_box_overlap(bx + 6 * (size_t)i, bx + 6 * (size_t)j)) continue;
            tst++;
            if (orc_neighbor_count(vidx + 3 * (size_t)i, vidx + 3 * (size_t)j) < 1) {
                uint32_t ia = ids ? ids[i] : (uint32_t)i, ib = ids ? ids[j] : (uint32_t)j;
                if (orc_tri_contact_helper(ia, vidx + 3 * (size_t)i, ib, vidx + 3 * (size_t)j, verts)) {
                    if (cnt < cap) { pairs[2 * cnt] = ia; pairs[2 * cnt + 1] = ib; }
                    cnt++;
                }
            }
        }
    }
    if (tested) *tested = tst;
    free(bx);
    return cnt;
}

/* ------------------------------------------------------------------ check.cuh:64-96, 29-50 */
/* checkInternalNodes: out = {nullParentNum, wrongBoundNum, nullChildNum, notInternalCount, uninitBoxCount}
 * in the order main.cu:115,119 prints them.  box_init: per-node "init" flag (box.cuh:21,31). */
void orc_check_internal(int n, const int32_t *left, const int32_t *right, const int32_t *parent,
                        const uint32_t *bounded, const uint8_t *box_init, uint32_t out[5])
{
    memset(out, 0, 5 * sizeof(uint32_t));
    for (int i = 0; i < n - 1; ++i) {
        if (bounded[i] != 2) out[1]++;
        if (parent[i] == -1) out[0]++;
        if (left[i] == -1) out[2]++;
        if (right[i] == -1) out[2]++;
        if (box_init && box_init[i] == 0) out[4]++;
    }
}
/* checkLeafNodes: out = {nullParentNum, nullTriangleNum, notLeafCount, illegalBoxCount} (main.cu:123,127).
 * nullTriangle also counts Triangle::selfCheck failures (vertex index >= maxv; triangle.cuh:11-16
 * hard-codes 632674, made a parameter here). */
void orc_check_leaves(int n, const int32_t *parent, const uint32_t *perm, const uint32_t *vidx, uint32_t maxv,
                      const uint8_t *box_init, uint32_t out[4])
{
    memset(out, 0, 4 * sizeof(uint32_t));
    for (int j = 0; j < n; ++j) {
        int node = (n - 1) + j;
        if (parent[node] == -1) out[0]++;
        const uint32_t *tv = vidx + 3 * (size_t)perm[j];
        if (tv[0] >= maxv || tv[1] >= maxv || tv[2] >= maxv) out[1]++;
        if (box_init && box_init[node] == 0) out[3]++;
    }
}
/* checkTriangleIdx, check.cuh:29-50: one count per out-of-range vertex index. */
uint32_t orc_check_triangle_idx(int n, const uint32_t *perm, const uint32_t *vidx, uint32_t maxv)
{
    uint32_t c = 0;
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < 3; ++k) if (vidx[3 * (size_t)perm[j] + k] >= maxv) c++;
    return c;
}

/* ------------------------------------------------------------------ whole path, for the CPU baseline */
#include <time.h>
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

typedef struct {
    double ms_morton, ms_sort, ms_hierarchy, ms_refit, ms_traverse;
} orc_times;

/* The build stages on `threads` OpenMP threads (liboracle_omp.so): each is the reference's own data-parallel
 * formulation run with one OpenMP iteration per CUDA thread -- Morton keys per triangle (load_obj.h:89-101),
 * generateHierarchyParallel per internal node (bvh.cuh:146-199; the parent links of a valid tree are written by
 * exactly one node each, the double-parent counter uses an atomic exchange), calBoundingBox per leaf with the
 * second arriver continuing (bvh.cuh:258-285; the arrival counter is an acquire-release RMW, which the reference
 * lacks).  The sort stays the sequential LSD radix sort (the reference sorts on one host thread too, load_obj.h:107);
 * its share is reported in ms_sort. */
#ifdef _OPENMP
static void build_parallel(const double *verts, const uint32_t *vidx, int n, const double off[3], const double span[3], int threads,
                           uint64_t *keys, uint32_t *perm, int32_t *left, int32_t *right, int32_t *parent, double *boxes, uint32_t *bounded,
                           double t[5])
{
    t[0] = now_ms();
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int tt = 0; tt < n; ++tt) {
        const double *p1 = verts + 3 * (size_t)vidx[3 * (size_t)tt + 0], *p2 = verts + 3 * (size_t)vidx[3 * (size_t)tt + 1], *p3 = verts + 3 * (size_t)vidx[3 * (size_t)tt + 2];
        keys[tt] = orc_morton3d((p1[0] + p2[0] + p3[0]) / 3, (p1[1] + p2[1] + p3[1]) / 3, (p1[2] + p2[2] + p3[2]) / 3, off, span);
    }
    t[1] = now_ms();
    orc_sort_by_key(keys, perm, (uint32_t)n);
    t[2] = now_ms();
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int i = 0; i < 2 * n - 1; ++i) parent[i] = -1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4096)
    for (int idx = 0; idx < n - 1; ++idx) {
        int first, last;
        orc_determine_range(keys, n, idx, 1, &first, &last);
        int split = orc_find_split(keys, n, first, last, 1);
        int32_t a = (split == first) ? (n - 1) + split : split, b = (split + 1 == last) ? (n - 1) + (split + 1) : split + 1;
        left[idx] = a; right[idx] = b;
        (void)__atomic_exchange_n(&parent[a], idx, __ATOMIC_RELAXED);
        (void)__atomic_exchange_n(&parent[b], idx, __ATOMIC_RELAXED);
    }
    t[3] = now_ms();
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int i = 0; i < n - 1; ++i) bounded[i] = 0;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4096)
    for (int j = 0; j < n; ++j) {
        uint32_t tt = perm[j];
        int node = (n - 1) + j;
        box_set(boxes + 6 * (size_t)node, verts + 3 * (size_t)vidx[3 * (size_t)tt], verts + 3 * (size_t)vidx[3 * (size_t)tt + 1], verts + 3 * (size_t)vidx[3 * (size_t)tt + 2]);
        int cur = parent[node];
        while (cur != -1) {
            if (__atomic_fetch_add(&bounded[cur], 1u, __ATOMIC_ACQ_REL) == 0) break;      /* bvh.cuh:270: the first arriver stops */
            box_merge(boxes + 6 * (size_t)cur, boxes + 6 * (size_t)left[cur], boxes + 6 * (size_t)right[cur]);
            cur = parent[cur];
        }
    }
    t[4] = now_ms();
}
#endif

/* main.cu:64-146 minus I/O: morton -> sort -> hierarchy -> refit -> traversal, single thread
 * (threads == 1), or every stage but the sort on `threads` OpenMP threads (threads > 1; pair order then
 * differs, sets do not). Returns the pair count; pairs may be NULL (count only). */
uint64_t orc_self_collide(const double *verts, const uint32_t *vidx, const uint32_t *ids, int n,
                          const double off[3], const double span[3], int threads,
                          uint32_t *pairs, uint64_t cap, orc_stats *st, orc_times *tm)
{
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    uint32_t *perm = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    int32_t *left = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *right = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *parent = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
    double *boxes = (double *)malloc(sizeof(double) * 12 * (size_t)n);
    uint32_t *bounded = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    double t0, t1, t2, t3, t4;
#ifdef _OPENMP
    if (threads > 1) {
        double tt[5];
        build_parallel(verts, vidx, n, off, span, threads, keys, perm, left, right, parent, boxes, bounded, tt);
        t0 = tt[0]; t1 = tt[1]; t2 = tt[2]; t3 = tt[3]; t4 = tt[4];
    } else
#endif
    {
        t0 = now_ms();
        orc_centroid_morton(verts, vidx, (uint32_t)n, off, span, keys, NULL);
        t1 = now_ms();
        orc_sort_by_key(keys, perm, (uint32_t)n);
        t2 = now_ms();
        uint32_t wrong;
        orc_build_hierarchy(keys, n, 1, left, right, parent, NULL, NULL, &wrong);
        t3 = now_ms();
        orc_refit(verts, vidx, perm, n, left, right, parent, boxes, bounded, NULL);
        t4 = now_ms();
    }
    memset(st, 0, sizeof *st);
    if (threads <= 1) {
        orc_find_collisions(verts, vidx, ids, perm, n, left, right, boxes, pairs, pairs ? cap : 0, st);
    } else {
#ifdef _OPENMP
        uint64_t np = 0, ptst = 0, nv = 0; uint32_t mx = 0;
#pragma omp parallel num_threads(threads) reduction(+ : np, ptst, nv) reduction(max : mx)
        {
            orc_stats loc; memset(&loc, 0, sizeof loc);
#pragma omp for schedule(dynamic, 4096)
            for (int j = 0; j < n; ++j) {
                uint32_t t = perm[j];
                traverse_one(ids ? ids[t] : t, vidx + 3 * (size_t)t, boxes + 6 * (size_t)((n - 1) + j),
                             verts, vidx, ids, perm, n, left, right, boxes, NULL, 0, &loc);
            }
            np += loc.n_pairs; ptst += loc.pairs_tested; nv += loc.node_visits; if (loc.max_stack > mx) mx = loc.max_stack;
        }
        st->n_pairs = np; st->pairs_tested = ptst; st->node_visits = nv; st->max_stack = mx;
#else
        orc_find_collisions(verts, vidx, ids, perm, n, left, right, boxes, pairs, pairs ? cap : 0, st);
#endif
    }
    double t5 = now_ms();
    if (tm) { tm->ms_morton = t1 - t0; tm->ms_sort = t2 - t1; tm->ms_hierarchy = t3 - t2; tm->ms_refit = t4 - t3; tm->ms_traverse = t5 - t4; }
    free(keys); free(perm); free(left); free(right); free(parent); free(boxes); free(bounded);
    return st->n_pairs;
}
