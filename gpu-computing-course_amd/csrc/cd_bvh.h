// cd_bvh.h -- LBVH build kernels: centroid + Morton keys, leaf fill, Karras-2012 hierarchy,
// bottom-up AABB refit, structural verifier counters.  Index-based SoA tree:
//   internal node i -> id i (root 0), leaf j -> id (n-1)+j, -1 = NULL.
#pragma once
#include "cd_math.h"
#include "cd_sort.h"

namespace cd {

// Topology of an internal node in one 16-byte record, written with one store by the node's own thread:
//   x = left child, y = right child (unified ids), z = the other end of the node's leaf range (Karras: node i
//   covers [min(i,z), max(i,z)]), w = 0.  Parent links live in parent[] (2n-1 entries, unified ids) because they
//   are written by the PARENT's thread.  Replaces the reference's pointer-linked 112-byte Node (bvh.cuh:25-43).
typedef int4 NodeMeta;

// ---------------------------------------------------------------- fp32 boxes that keep TIES: the cell table
// The traversal compares fp32 copies of FP64 box bounds with the reference's strict '<' (box.cuh:40-43 as intervals:
// a.lo < b.hi).  Plain outward rounding (lo down, hi up) is conservative, but it turns every pair of bounds that are
// EQUAL and not fp32 values into an overlap of one ulp -- and in a mesh neighbouring triangles share box faces exactly
// (the same vertex coordinate), so a mesh whose coordinates are full doubles had 5 x the node visits and 70 x the
// candidates of the same mesh rounded to float.  No lossy representation can tell "equal" from "a hair apart" -- but
// the table below knows where it matters: a CELL is the set of doubles with the same fp32 floor, cell(x) = rd32(x), and a
// cell is AMBIGUOUS iff two DISTINCT doubles among the mesh's vertex coordinates of that axis lie in it (every box bound
// is a vertex coordinate).  Encoding:   lo' = rd32(lo)       hi' = rd32(hi), one ulp up iff cell(hi) is ambiguous AND hi is
// not the cell's base itself (nothing of the cell lies below its base).
//   * different cells: rd32 is monotone and cells are an ulp apart, so lo' < hi' <=> lo < hi, exactly;
//   * the same cell, hi' not moved: hi is the base, or the only double of its cell -- then lo >= hi: lo' == hi', '<' says no, exact;
//   * the same cell, hi' moved: hi' = lo' + ulp, '<' says "maybe" -- conservative, as outward rounding always was.
// hi' is monotone in hi (a lower cell's base + ulp is at most the next cell's base), so it commutes with max: the fp32
// box of a leaf range is still the min / max of its leaves' fp32 boxes, bit for bit (cd_build.h).  A box none of whose three
// hi bounds was moved is CERTAIN: any '<' between it and another certain box decides what FP64 decides (a comparison is a lo
// against a hi, and it is the hi's treatment that makes it exact).
// A mesh whose coordinates are all fp32 values has no ambiguous cell (distinct floats are distinct cells) and no table:
// keys == nullptr, hi' = rd32(hi) = hi -- the encoding of earlier rounds, bit for bit.  -0.0 and +0.0 are one cell.
// Bounds of OTHER meshes (external queries, peer root boxes) are not in the table: those comparisons treat an fp32 tie
// as "maybe" unless both sides are fp32 values (cd_traverse.h).  The table is rebuilt when the vertices are uploaded
// (cd_create, cd_update_vertices: mi355cd.hip amb_refresh), not per step: it is a function of the vertices alone.
// (keys == nullptr, mask != 0: no table although the mesh has coordinates that are not fp32 values -- CD_OPT_CELL_TABLE 0 -- : every such
//  coordinate counts as lying in an ambiguous cell, which is plain outward rounding, the encoding of rounds 1 - 3)
struct AmbTable { const unsigned long long *keys; uint32_t shift, mask; };
constexpr unsigned long long AMB_BIT = 1ull << 63;
__device__ __forceinline__ uint32_t amb_cell(double x)
{
    const uint32_t c = __float_as_uint(__double2float_rd(x));
    return c == 0x80000000u ? 0u : c;                                      // -0.0 and +0.0 compare equal: one cell
}
__device__ __host__ __forceinline__ unsigned long long amb_key(int axis, uint32_t cell) { return (((unsigned long long)axis << 32) | cell) + 1ull; }   // (never 0 = empty slot)
__device__ __host__ __forceinline__ uint32_t amb_hash(unsigned long long key, uint32_t shift) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift); }
// Is the cell of x (a vertex coordinate of axis `axis`) ambiguous?  A value that is not in the table -- cannot happen for
// a bound of this mesh -- counts as ambiguous (conservative).
__device__ __forceinline__ bool amb_lookup(const AmbTable &t, int axis, double x)
{
    if (!t.keys) return t.mask != 0u && (double)(float)x != x;
    const unsigned long long key = amb_key(axis, amb_cell(x));
    uint32_t h = amb_hash(key, t.shift);
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        const unsigned long long k = t.keys[h];
        if ((k & ~AMB_BIT) == key) return (k & AMB_BIT) != 0ull;
        if (k == 0ull) return true;
        h = (h + 1u) & t.mask;
    }
    return true;
}
__device__ __forceinline__ float f32_next_up(float f)
{
    const uint32_t u = __float_as_uint(f);
    if ((u << 1) == 0u) return __uint_as_float(1u);                        // +-0 -> the smallest positive value
    return __uint_as_float((u >> 31) ? u - 1u : u + 1u);
}
// the fp32 copy of an FP64 box as described above, and whether the box is certain
struct Enc32 { float lx, ly, lz, hx, hy, hz; bool certain; };
__device__ __forceinline__ Enc32 enc_box32(const AmbTable &t, const Box &b)
{
    Enc32 e;
    e.lx = __double2float_rd(b.x1); e.ly = __double2float_rd(b.y1); e.lz = __double2float_rd(b.z1);
    e.hx = __double2float_rd(b.x2); e.hy = __double2float_rd(b.y2); e.hz = __double2float_rd(b.z2);
    // (a hi that IS its cell's base has nothing of the cell below it: never moved)
    const bool ux = (double)e.hx != b.x2 && amb_lookup(t, 0, b.x2), uy = (double)e.hy != b.y2 && amb_lookup(t, 1, b.y2), uz = (double)e.hz != b.z2 && amb_lookup(t, 2, b.z2);
    if (ux) e.hx = f32_next_up(e.hx);
    if (uy) e.hy = f32_next_up(e.hy);
    if (uz) e.hz = f32_next_up(e.hz);
    e.certain = !(ux | uy | uz);
    return e;
}
// The same for a LEAF box, whose bounds are coordinates of its own three vertices: the vertices' flags (vamb[v] bit a: the cell of
// vertex v's coordinate a is ambiguous; written once per upload by k_amb_vertex, mi355cd.hip) stand in for probes of the table.
// Equal coordinates lie in one cell, so any vertex that attains a bound has the bound's flag.  vamb == nullptr: no table.
__device__ __forceinline__ Enc32 enc_leaf32(const AmbTable &t, const Box &b, const d3 A, const d3 B, const d3 C, const uint8_t *__restrict__ vamb, uint32_t va, uint32_t vb, uint32_t vc)
{
    if (!vamb && t.mask != 0u) return enc_box32(t, b);                      // (uniform: no table by request -- every inexact bound counts as ambiguous)
    Enc32 e;
    e.lx = __double2float_rd(b.x1); e.ly = __double2float_rd(b.y1); e.lz = __double2float_rd(b.z1);
    e.hx = __double2float_rd(b.x2); e.hy = __double2float_rd(b.y2); e.hz = __double2float_rd(b.z2);
    e.certain = true;
    if (!vamb) return e;                                                   // (uniform: no table, nothing is moved)
    const uint32_t fa = vamb[va], fb = vamb[vb], fc = vamb[vc];
    auto pick = [&](double bound, double a, double bb, int bit) -> bool { const uint32_t f = (a == bound) ? fa : ((bb == bound) ? fb : fc); return ((f >> bit) & 1u) != 0u; };
    const bool ux = (double)e.hx != b.x2 && pick(b.x2, A.x, B.x, 0), uy = (double)e.hy != b.y2 && pick(b.y2, A.y, B.y, 1), uz = (double)e.hz != b.z2 && pick(b.z2, A.z, B.z, 2);
    if (ux) e.hx = f32_next_up(e.hx);
    if (uy) e.hy = f32_next_up(e.hy);
    if (uz) e.hz = f32_next_up(e.hz);
    e.certain = !(ux | uy | uz);
    return e;
}

// fp32 traversal record, one 64-byte line: both child boxes as fp32 copies (lo down; hi down, one ulp up where its cell is
// ambiguous: above) + both child links.  Internal-node boxes only cull; a conservative (superset) box can never lose a
// pair, and every leaf hit is decided by the exact FP64 product-form test of box.cuh:40-43 before it counts -- which for
// two CERTAIN leaf boxes is what the fp32 '<' already decided.
//
// Records are NAMED BY SPLIT: recs[s] is the internal node whose left child covers [first, s] and whose right
// child covers [s + 1, last] (every s in [0, n-2] is the split of exactly one Karras node, so this is a
// permutation of the Karras numbering that meta[] / parent[] / the exported tree keep).  A child link is the
// child's own split for an internal child and ~j for leaf j.  With that naming the right siblings hanging off
// the root path of leaf j -- the subtrees that partition the leaves (j, n-1] -- are reached bottom-up with no
// parent pointers: s = j; { right child of recs[s]; s = recs[s].last; } until last == n-1 (cd_traverse.h).
// The two 32-byte halves of a record live in two ARRAYS (right halves first, then the left halves, in one allocation
// of n x 64 bytes): a hop of that chain reads only the right half, so the chain walks a dense 32-byte-per-node array
// (rec_right); a descent step reads both (rec_left, rec_right).  NodeRec32 remains the logical record.
constexpr uint32_t REC_LAST_MASK = 0x3fffffffu;     // n <= 2^30 (the candidate encoding has the same limit)
constexpr uint32_t REC_L_CERTAIN = 0x40000000u;     // in `last`: the left / right child is a leaf whose box is CERTAIN (none of its hi bounds
constexpr uint32_t REC_R_CERTAIN = 0x80000000u;     //   moved: an fp32 '<' against another certain box of this mesh is exact)
constexpr uint32_t REC_L_EXACT = 0x40000000u;       // in `first`: the left / right child is a leaf whose box is EXACT (certain, and its six
constexpr uint32_t REC_R_EXACT = 0x80000000u;       //   bounds are fp32 values: the fp32 copy IS the box -- what a query from another mesh needs)
struct alignas(64) NodeRec32 {
    float l_lo[3], l_hi[3]; int32_t cl; uint32_t first;     // quad 0, quad 1: left child, range start | REC_*_EXACT
    float r_lo[3], r_hi[3]; int32_t cr; uint32_t last;      // quad 2, quad 3: right child, range end | REC_*_CERTAIN
};
static_assert(sizeof(NodeRec32) == 64, "NodeRec32 is two 32-byte halves");
// quads (16 bytes) of the two halves of record s; `recs` is the allocation's base, n the number of leaves (= record slots)
__device__ __forceinline__ const float4 *rec_right(const NodeRec32 *recs, int n, uint32_t s) { return reinterpret_cast<const float4 *>(recs) + 2 * (size_t)s; }
__device__ __forceinline__ const float4 *rec_left(const NodeRec32 *recs, int n, uint32_t s) { return reinterpret_cast<const float4 *>(recs) + 2 * (size_t)n + 2 * (size_t)s; }

// fp32 query box of leaf j, encoded like the records', written by the refit: 32 coalesced bytes per query
// instead of the 48-byte FP64 box (k_descend and the packers read it; k_descend_half takes the same box and the same
// certain bit out of the leaf's parent record, which it reads anyway -- cd_traverse.h).  flags bit 0: the box is EXACT (fp32 values, certain); bit 2: CERTAIN; bit 1: the FP64 box strictly
// overlaps itself (box.cuh:40-43 with a == b -- false for a box that is flat along an axis): the query's hit on its
// own leaf, which every traversal of the reference meets once (collision.cuh:31-32), is decided here, exactly.
struct alignas(32) LeafBox32 { float lo[3], hi[3]; uint32_t flags, pad; };
static_assert(sizeof(LeafBox32) == 32, "LeafBox32 layout");
constexpr uint32_t LB_EXACT = 1u, LB_SELF = 2u, LB_CERTAIN = 4u;

// FP64 box of leaf j.  An EXACT box IS its query box (fp32 values in unambiguous cells: nothing was moved; widening is exact, signed zeros included), and the
// fused build (cd_build.h) does not store the FP64 copy of such a leaf at all: every reader of leaf boxes on the
// traversal side comes through here.  (The stage-wise refit writes all of boxes[]: that is what cd_export_tree shows.)
__device__ __forceinline__ Box load_box(const double *boxes, int node);
__device__ __forceinline__ Box leaf_box64(const double *__restrict__ boxes, const LeafBox32 *__restrict__ qbox32, int n, int j)
{
    const float4 a = reinterpret_cast<const float4 *>(qbox32 + j)[0], b = reinterpret_cast<const float4 *>(qbox32 + j)[1];
    if (__float_as_uint(b.z) & LB_EXACT) return Box{(double)a.x, (double)a.w, (double)a.y, (double)b.x, (double)a.z, (double)b.y};
    return load_box(boxes, (n - 1) + j);
}

// ---------------------------------------------------------------- which workgroup of the half traversal takes which 64 leaves
// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD and its 4 MiB L2).  An XCD works on HALF_XSUB chunks of consecutive
// groups from different parts of the mesh rather than ONE contiguous eighth: its L2 still sees an eighth of the tree, and the work per query is not
// even over a mesh (where the surfaces meet, a query has candidates; elsewhere none) -- an XCD with a busy eighth was the kernel's tail.
// 1 M cloth: 1 / 2 / 4 / 8 / 16 chunks -> 57.8 / 57.9 / 54.1 / 54.4 / 54.6 us.  Speed only: any bijection of [0, nb) gives the same results.
constexpr int HALF_XSUB = 4;
__host__ __device__ __forceinline__ uint32_t half_vblock_hd(uint32_t b, uint32_t nb);
__device__ __forceinline__ uint32_t half_vblock(uint32_t b, uint32_t nb) { return half_vblock_hd(b, nb); }
inline uint32_t half_vblock_host(uint32_t b, uint32_t nb) { return half_vblock_hd(b, nb); }
__host__ __device__ __forceinline__ uint32_t half_vblock_hd(uint32_t b, uint32_t nb)
{
    const uint32_t per = nb >> 3;
    uint32_t v = (b < (per << 3)) ? (b & 7u) * per + (b >> 3) : b;
    const uint32_t c = per / HALF_XSUB;
    if (c > 0 && b < c * HALF_XSUB * 8u) {
        const uint32_t x = b & 7u, l = b >> 3, sub = l / c, off = l % c;
        v = (sub * 8u + x) * c + off;
    } else if (c > 0) v = b;                             // (the remainder keeps its own index: c * HALF_XSUB * 8 <= b < nb are not produced above)
    return v;
}
// The ORDER HINT.  The half traversal's kernel ends with its unluckiest wave slot: 1.9 rounds of waves that take 13 .. 40 us each, dispatched in index order
// (profiles/r03_experiments/descent_ablation.log: dispatch order 51.8 us, the same waves longest-first 40.5).  Nothing known inside a step predicts a long
// wave -- but the previous step does, if its times are carried over BY TRIANGLE: a mesh that moves a quarter of a quad sorts into other groups of 64 (positions
// shift along the whole Morton order: times remembered by position are a random order, which is worse than the index order), while the triangles near a contact
// curve are still near it.  Every wave leaves its duration (a class of 1.28 us, ORDER_CLASSES of them) with each of its 64 TRIANGLES (tri_cost[original index], a
// byte); the next step's k_build_block, whose wave w of block b holds exactly the leaves of one group, takes the MAX of its leaves' bytes as the group's score
// (cost[group]); 8 workgroups of k_cross_fused then sort, per XCD, the groups that XCD works on by score, longest first (stable: equal scores keep half_vblock's
// order), into order[]: the descent's workgroup b takes group order[b].  Which groups an XCD works on does not change, so its L2 sees what it saw.
// A hint only: tri_cost[] may hold anything (zeros in the first step) -- order[] is a permutation whatever it holds, and any permutation gives the same results.
// tools/hint_predictors.py (orders installed from outside): sheet B moving 0.25 / 1 / 4 quads a frame, descent 53.5 us in the index order, 47.5 with the frame's
// OWN times (a perfect predictor), 55 by position, 49.8 / 48.7 / 53.8 by triangle with the max (51 / 51 / 54 with the mean).
// One workgroup of T threads per XCD list (x = 0 .. 7) and chunk of ORDER_MAX_ITEMS list positions, in LDS the caller lends it (OrderLds<T>).  A stable counting
// sort in five barriers: the scores of the list into LDS (one gather), a count per thread and class over the thread's run of consecutive items, a scan down each
// class's column, the items placed run by run.
constexpr int ORDER_CLASSES = 32, ORDER_SHIFT = 7 /* 2^7 ticks of the 100 MHz wall clock */, ORDER_MAX_ITEMS = 2048;
template <int T> struct OrderLds { uint8_t cls[ORDER_MAX_ITEMS]; uint16_t cnt[T][ORDER_CLASSES]; uint32_t base[ORDER_CLASSES]; };
// (round 5) A list of more than ORDER_MAX_ITEMS groups -- a tree of more than 1 M leaves -- is sorted CHUNK by chunk of ORDER_MAX_ITEMS consecutive list
// positions, a workgroup each: the descent works a chunk's groups off longest first, and what decides when the kernel ends, the last chunk, is packed like a small tree's list.
template <int T>
__device__ __forceinline__ void build_half_order(uint32_t x, uint32_t chunk, uint32_t nb, const uint32_t *__restrict__ cost, uint32_t *__restrict__ order, OrderLds<T> &L)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t all = x < nb ? (nb - x + 7u) / 8u : 0u;                   // workgroups b = 8 l + x < nb
    const uint32_t lbase = chunk * (uint32_t)ORDER_MAX_ITEMS;                // this chunk: list positions lbase .. lbase + cnt - 1
    const uint32_t cnt = all > lbase ? (all - lbase < (uint32_t)ORDER_MAX_ITEMS ? all - lbase : (uint32_t)ORDER_MAX_ITEMS) : 0u;
    order += 8u * lbase; 
    auto vb = [&](uint32_t l) { return half_vblock(8u * (lbase + l) + x, nb); };
    for (uint32_t l = tid; l < cnt; l += T) { const uint32_t c = cost[vb(l)]; L.cls[l] = (uint8_t)(c < ORDER_CLASSES ? c : ORDER_CLASSES - 1); }
    for (int k = 0; k < ORDER_CLASSES; ++k) L.cnt[tid][k] = 0;
    __syncthreads();
    const uint32_t per = (cnt + T - 1) / T, l0 = tid * per < cnt ? tid * per : cnt, l1 = l0 + per < cnt ? l0 + per : cnt;   // this thread's run of the list
    for (uint32_t l = l0; l < l1; ++l) ++L.cnt[tid][L.cls[l]];
    __syncthreads();
    if (tid < ORDER_CLASSES) {                                               // class tid: the threads' counts -> how many of the class lie in EARLIER runs
        uint32_t run = 0;
        for (int t = 0; t < T; ++t) { const uint32_t v = L.cnt[t][tid]; L.cnt[t][tid] = (uint16_t)run; run += v; }
        L.base[tid] = run;
    }
    __syncthreads();
    uint32_t before = 0;
    if (tid < ORDER_CLASSES) for (uint32_t c = ORDER_CLASSES - 1; c > tid; --c) before += L.base[c];   // the longest class first
    __syncthreads();
    if (tid < ORDER_CLASSES) L.base[tid] = before;
    __syncthreads();
    for (uint32_t l = l0; l < l1; ++l) { const uint32_t c = L.cls[l]; const uint32_t pos = L.base[c] + L.cnt[tid][c]++; order[8u * pos + x] = vb(l); }
}

// Sorted-order leaf payload: {ID, vIdx[0..2]} (triangle.cuh:6,9) -- 16 B instead of the 56-byte Triangle.
struct alignas(16) LeafTri { uint32_t id, v0, v1, v2; };

// ---------------------------------------------------------------- centroid AABB (CD_FRAME_AUTO)
// Stage 1: per-workgroup min/max of centroids + the adaptive frame's statistic (cd_math.h); stage 2 (fold_frame: k_morton's workgroups
// themselves, or k_frame_from_bounds) folds the partials and forms frame = {off[3], span[3], layout word, 0}.  span is widened by 2^-20
// relative so that the max maps below the last cell's end.
__device__ __forceinline__ d3 centroid_of(const double *__restrict__ verts, const uint32_t *__restrict__ vidx, uint32_t t)
{
    const uint32_t a = vidx[3 * (size_t)t], b = vidx[3 * (size_t)t + 1], c = vidx[3 * (size_t)t + 2];
    const d3 p1 = load_vertex(verts, a), p2 = load_vertex(verts, b), p3 = load_vertex(verts, c);
    // load_obj.h:90: (p1 + p2 + p3) / 3 per axis
    return d3{(p1.x + p2.x + p3.x) / 3, (p1.y + p2.y + p3.y) / 3, (p1.z + p2.z + p3.z) / 3};
}

__device__ __forceinline__ double wave_min(double v) { for (int o = 32; o; o >>= 1) { double t = __shfl_xor(v, o); v = t < v ? t : v; } return v; }
__device__ __forceinline__ double wave_max(double v) { for (int o = 32; o; o >>= 1) { double t = __shfl_xor(v, o); v = t > v ? t : v; } return v; }

// BOX: also the min / max over the triangles' VERTICES -- the box of all leaves (what node 0 of the tree will hold),
// known before there is a tree: the multi-GPU step exchanges it first (cd_multi.h).
// partial: BOUNDS_STRIDE values x gridDim.x blocks, value-major: centroid lo[3] hi[3], vertex lo[3] hi[3] (BOX), and the adaptive
// frame's statistic (cd_math.h) as 64-bit INTEGERS in the same slots: sum[3] of flog2(box extent), cnt[3] of the boxes that are not flat.
constexpr int BOUNDS_STRIDE = 18, BOUNDS_STAT = 12;
__device__ __forceinline__ long long wave_sum_ll(long long v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }
template <bool BOX>
__global__ __launch_bounds__(256) void k_centroid_bounds(const double *__restrict__ verts, const uint32_t *__restrict__ vidx, uint32_t n,
                                                         double *__restrict__ partial /* BOUNDS_STRIDE x gridDim.x */)
{
    __shared__ double sm[4][BOUNDS_STRIDE];
    constexpr int SETS = BOX ? 2 : 1;
    double lo[SETS][3], hi[SETS][3];
    long long ssum[3] = {0, 0, 0}, scnt[3] = {0, 0, 0};
    for (int q = 0; q < SETS; ++q) for (int a = 0; a < 3; ++a) { lo[q][a] = 1e300; hi[q][a] = -1e300; }
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const uint32_t ia = vidx[3 * (size_t)t], ib = vidx[3 * (size_t)t + 1], ic = vidx[3 * (size_t)t + 2];
        const d3 p1 = load_vertex(verts, ia), p2 = load_vertex(verts, ib), p3 = load_vertex(verts, ic);
        const double c[3] = {(p1.x + p2.x + p3.x) / 3, (p1.y + p2.y + p3.y) / 3, (p1.z + p2.z + p3.z) / 3};      // load_obj.h:90
        for (int a = 0; a < 3; ++a) { lo[0][a] = c[a] < lo[0][a] ? c[a] : lo[0][a]; hi[0][a] = c[a] > hi[0][a] ? c[a] : hi[0][a]; }
        const Box b = box_set(p1, p2, p3);
        const double bl[3] = {b.x1, b.y1, b.z1}, bh[3] = {b.x2, b.y2, b.z2};
        for (int a = 0; a < 3; ++a) { const double e = bh[a] - bl[a]; if (e > FLOG_MIN) { ssum[a] += flog2_fixed(e); scnt[a] += 1; } }
        if (BOX)
            for (int a = 0; a < 3; ++a) { lo[SETS - 1][a] = bl[a] < lo[SETS - 1][a] ? bl[a] : lo[SETS - 1][a]; hi[SETS - 1][a] = bh[a] > hi[SETS - 1][a] ? bh[a] : hi[SETS - 1][a]; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int q = 0; q < SETS; ++q)
        for (int a = 0; a < 3; ++a) {
            const double l = wave_min(lo[q][a]), h = wave_max(hi[q][a]);
            if (lane == 0) { sm[w][6 * q + a] = l; sm[w][6 * q + 3 + a] = h; }
        }
    for (int a = 0; a < 3; ++a) {
        const long long s = wave_sum_ll(ssum[a]), c = wave_sum_ll(scnt[a]);
        if (lane == 0) { sm[w][BOUNDS_STAT + a] = __longlong_as_double(s); sm[w][BOUNDS_STAT + 3 + a] = __longlong_as_double(c); }
    }
    __syncthreads();
    if (threadIdx.x < 6 * SETS) {
        const bool is_lo = (threadIdx.x % 6) < 3;
        double v = sm[0][threadIdx.x];
        for (int ww = 1; ww < 4; ++ww) { const double t = sm[ww][threadIdx.x]; v = is_lo ? (t < v ? t : v) : (t > v ? t : v); }
        partial[threadIdx.x * gridDim.x + blockIdx.x] = v;                   // value-major: the folds below read consecutive blocks
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + 6) {
        const int k = BOUNDS_STAT + (int)threadIdx.x - 64;
        long long v = 0;
        for (int ww = 0; ww < 4; ++ww) v += __double_as_longlong(sm[ww][k]);
        partial[k * gridDim.x + blockIdx.x] = __longlong_as_double(v);
    }
}
// The fold of the per-block partials into the frame, by the calling workgroup (T threads, T / 64 <= 16 waves; every thread calls).
// min / max are exact and order-independent, the statistic is integer: any order gives the same frame.
// Returns (in every thread, after its barriers) through `sf`: off[3], span[3], the layout word's bits in sf[6] -- and, when box != nullptr, the vertex box in box[0..5].
template <int T>
__device__ __forceinline__ void fold_frame(const double *__restrict__ partial, uint32_t nblocks, bool with_box, double (*smw)[BOUNDS_STRIDE], double *sf /* [8] shared */, double *sbox /* [6] shared, or nullptr */)
{
    // A WAVE per value (k = 0 .. 17: which of the 18 rows of `partial`), the kind of fold uniform in the wave: a lane folds every 64th block of the row, then ONE wave
    // reduction.  (Round 6's first form -- every thread all 18 values, 18 wave reductions interleaved -- took 127 VGPRs with spills inside k_morton, whose own loop needs 47.)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll 1
    for (int k = w; k < BOUNDS_STRIDE; k += T / 64) {                         // (k is wave-uniform)
        if (!with_box && k >= 6 && k < BOUNDS_STAT) continue;
        const double *row = partial + (size_t)k * nblocks;
        if (k >= BOUNDS_STAT) {
            long long v = 0;
#pragma unroll 1
            for (uint32_t b = lane; b < nblocks; b += 64) v += __double_as_longlong(row[b]);
            v = wave_sum_ll(v);
            if (lane == 0) smw[0][k] = __longlong_as_double(v);
        } else if ((k % 6) < 3) {
            double v = 1e300;
#pragma unroll 1
            for (uint32_t b = lane; b < nblocks; b += 64) { const double t = row[b]; v = t < v ? t : v; }
            v = wave_min(v);
            if (lane == 0) smw[0][k] = v;
        } else {
            double v = -1e300;
#pragma unroll 1
            for (uint32_t b = lane; b < nblocks; b += 64) { const double t = row[b]; v = t > v ? t : v; }
            v = wave_max(v);
            if (lane == 0) smw[0][k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double lo[3], hi[3]; long long sum[3], cnt[3];
        for (int a = 0; a < 3; ++a) {
            const double l = smw[0][a], h = smw[0][3 + a];
            lo[a] = l; hi[a] = h; sum[a] = __double_as_longlong(smw[0][BOUNDS_STAT + a]); cnt[a] = __double_as_longlong(smw[0][BOUNDS_STAT + 3 + a]);
            double span = (h - l) * (1.0 + 1.0 / 1048576.0);                  // widened by 2^-20 relative: the max maps below the last cell's end
            if (!(span > 0.0)) span = 1.0;
            sf[a] = l; sf[3 + a] = span;
        }
        sf[6] = __longlong_as_double((long long)frame_layout(lo, hi, sum, cnt));
        sf[7] = 0.0;
    } else if (with_box && threadIdx.x >= 64 && threadIdx.x < 64 + 3) {
        const int a = (int)threadIdx.x - 64;
        sbox[2 * a] = smw[0][6 + a]; sbox[2 * a + 1] = smw[0][9 + a];
    }
    __syncthreads();
}
// One workgroup of 256 folds the per-block partials (it used to be three threads walking all of them one dependent
// load after the other: 180 us for 1024 blocks).
// frame (may be NULL): off[3], span[3], layout word, 0 -- FRAME_WORDS doubles.  box (may be NULL): {x1,x2,y1,y2,z1,z2} from the vertex bounds
// (partials written by k_centroid_bounds<true>).
constexpr int FRAME_WORDS = 8;
__global__ __launch_bounds__(256) void k_frame_from_bounds(const double *__restrict__ partial, uint32_t nblocks, double *__restrict__ frame,
                                                           double *__restrict__ box)
{
    __shared__ double sm[4][BOUNDS_STRIDE];
    __shared__ double sf[FRAME_WORDS], sbox[6];
    fold_frame<256>(partial, nblocks, box != nullptr, sm, sf, sbox);
    if (frame && threadIdx.x < FRAME_WORDS) frame[threadIdx.x] = sf[threadIdx.x];
    if (box && threadIdx.x < 6) box[threadIdx.x] = sbox[threadIdx.x];
}

// ---------------------------------------------------------------- Morton keys (load_obj.h:89-101, morton.h:70-89)
// Grid-stride; the workgroup also accumulates the radix sort's digit histograms (cd_sort.h) of the keys it writes.
constexpr int MORTON_THREADS = 1024;                  // one sort tile (SORT_TILE = 4096 keys) per workgroup pass: 4 keys per thread
// LAYOUT: the frame may carry a key layout (cd_math.h: the adaptive frame -- always so in CD_FRAME_AUTO, where the layout is formed on the device); false: the reference's interleave
// only (CD_FRAME_REFERENCE, CD_FRAME_CUSTOM without a layout: the host knows).  Two instances so that the reference's path keeps its registers: with both paths in one body the kernel
// took 91 VGPRs instead of 48 -- one 1024-thread workgroup a CU instead of two, 13.5 -> 16.2 us at 1 M keys.
template <bool LAYOUT>
__global__ __launch_bounds__(MORTON_THREADS) void k_morton(const double *__restrict__ verts, const uint32_t *__restrict__ vidx, uint32_t n,
                                                           const double *__restrict__ frame /* off[3], span[3], layout word, 0 (FRAME_WORDS) */,
                                                           uint64_t *__restrict__ keys, uint32_t *__restrict__ ghist /* [HIST_COPIES][8][256] */, int first_digit,
                                                           int down /* shifted hybrid sort: digits taken `down` bits lower */, uint32_t *__restrict__ overflow,
                                                           const double *__restrict__ partial /* auto frame: k_centroid_bounds' per-block bounds, else NULL */, uint32_t nparts,
                                                           double *__restrict__ frame_out, double *__restrict__ box_out /* NULL, or the box of all vertices (partials of k_centroid_bounds<true>) */,
                                                           uint32_t *__restrict__ tile_hist /* [tiles][256]: per SORT_TILE keys, the counts of digit `first_digit` -- the first global pass of the sort
                                                                                              takes its tile offsets from these (no look-back: k_os_pass) */)
{
    __shared__ uint32_t h[8][RADIX];
    __shared__ double smw[MORTON_THREADS / 64][BOUNDS_STRIDE];
    __shared__ double sframe[FRAME_WORDS], sbox[6];
    for (int i = threadIdx.x; i < 8 * RADIX; i += MORTON_THREADS) (&h[0][0])[i] = 0;
    // Auto frame: EVERY workgroup folds the per-block bounds itself (min / max are exact and order-independent, the statistic is integer:
    // the same frame in all of them) -- 72 KB of L2 reads and a microsecond, against a one-workgroup kernel of its own in front of
    // this one (k_frame_from_bounds: ~6 us + a launch gap).  Workgroup 0 stores the frame (and the vertex box) for the others.
    if (LAYOUT && partial) {                                                    // (uniform; the host launches the LAYOUT instance for an auto frame)
        fold_frame<MORTON_THREADS>(partial, nparts, box_out != nullptr, smw, sframe, sbox);
        if (blockIdx.x == 0) {
            if (frame_out && threadIdx.x < FRAME_WORDS) frame_out[threadIdx.x] = sframe[threadIdx.x];
            if (box_out && threadIdx.x < 6) box_out[threadIdx.x] = sbox[threadIdx.x];
        }
        frame = sframe;
    }
    __syncthreads();
    // the key layout of the frame (cd_math.h): 0 = the reference's interleave (morton.h:70-89, bit-identical), else the adaptive one
    unsigned long long layout = 0ull;
    KeyLayout kl = {};
    if constexpr (LAYOUT) {
        layout = (unsigned long long)__double_as_longlong(uniform_f64(frame[6]));
        kl = key_layout(layout, frame, frame + 3);
    }
    uint64_t above = 0;
    // a workgroup takes whole sort tiles (SORT_TILE consecutive keys): beside the digit histograms of ALL keys it leaves, per tile,
    // the counts of the first global digit -- with those the sort's first pass needs no rendezvous between its tiles
    const uint32_t ntile = (n + SORT_TILE - 1) / SORT_TILE;
    uint32_t seen = 0;                                   // thread d < 256: what h[first_digit][d] held when this tile began (the tile's counts are the difference)
    for (uint32_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
#pragma unroll
        for (int it = 0; it < SORT_TILE / MORTON_THREADS; ++it) {
            const uint32_t t = tile * SORT_TILE + it * MORTON_THREADS + threadIdx.x;
            if (t < n) {
                uint64_t k;
                if constexpr (LAYOUT) {                                   // (the host launches this instance only for a frame WITH a layout: an auto frame, or cd_set_morton_frame_layout's)
                    const uint32_t ia = vidx[3 * (size_t)t], ib = vidx[3 * (size_t)t + 1], ic = vidx[3 * (size_t)t + 2];
                    const d3 p1 = load_vertex(verts, ia), p2 = load_vertex(verts, ib), p3 = load_vertex(verts, ic);
                    k = morton3d_layout(p1.x + p2.x + p3.x, p1.y + p2.y + p3.y, p1.z + p2.z + p3.z, kl);      // the vertex sum: no division per key (cd_math.h)
                } else { const d3 c = centroid_of(verts, vidx, t); k = morton3d(c.x, c.y, c.z, frame, frame + 3); }
                keys[t] = k;
                hist_add(h, k, first_digit, down);
                above |= k;
            }
        }
        __syncthreads();
        if (threadIdx.x < RADIX) { const uint32_t now = h[first_digit][threadIdx.x]; tile_hist[(size_t)tile * RADIX + threadIdx.x] = now - seen; seen = now; }
        if (tile + gridDim.x < ntile) __syncthreads();   // (more than 2048 tiles: the next tile's counts start on top of these)
    }
    // a centroid outside the Morton frame sets key bits the shifted digits do not cover (bits 64 - down .. 63): the
    // shifted hybrid sort then does not apply -- same flag, same repair (the next form) as a run too long to window
    if (down && (above >> (64 - down))) atomicOr(overflow, SORTF_ABOVE);   // a key beyond the shifted digits -- the mesh has left the frame (cd_sort.h: the flag word's bits)
    __syncthreads();
    for (int i = threadIdx.x + first_digit * RADIX; i < 8 * RADIX; i += MORTON_THREADS) {
        const uint32_t v = (&h[0][0])[i];
        if (v) atomicAdd(&ghist[(size_t)(blockIdx.x & (HIST_COPIES - 1)) * HIST_STRIDE + i], v);
    }
}

// ---------------------------------------------------------------- fillLeafNodes (bvh.cuh:125-144)
// leaf j <- triangle perm[j]; also resets the parent links and the refit arrival counters that the
// reference gets from zero-initialised cudaMalloc memory (main.cu:84-85).
__device__ __forceinline__ void fill_leaf(uint32_t j, uint32_t t, const uint32_t *__restrict__ vidx, const uint32_t *__restrict__ ids, uint32_t n,
                                          LeafTri *__restrict__ leaf, int32_t *__restrict__ parent, uint32_t *__restrict__ bounded, double *__restrict__ boxes)
{
    LeafTri lt;
    lt.id = ids ? ids[t] : t;
    lt.v0 = vidx[3 * (size_t)t]; lt.v1 = vidx[3 * (size_t)t + 1]; lt.v2 = vidx[3 * (size_t)t + 2];
    leaf[j] = lt;
    if (!parent) return;                                   // the fused build follows: it reads leaf[] only (12 bytes per leaf less to write)
    parent[(n - 1) + j] = -1;
    // boxes are "uninitialised" until the refit writes them (Box::init, box.cuh:10,21,31): poison x1
    if (boxes) reinterpret_cast<uint64_t *>(boxes)[6 * (size_t)((n - 1) + j)] = 0xFFFFFFFFFFFFFFFFull;
    if (j < n - 1) { parent[j] = -1; bounded[j] = 0; if (boxes) reinterpret_cast<uint64_t *>(boxes)[6 * (size_t)j] = 0xFFFFFFFFFFFFFFFFull; }
}

// what k_local_sort's epilogue (cd_sort.h) does with a key's final position: fill that leaf (fill_leaf in two halves:
// the gather by triangle number does not wait for the position)
struct LeafFill {
    const uint32_t *vidx, *ids; uint32_t n; LeafTri *leaf; int32_t *parent; uint32_t *bounded;
    typedef LeafTri Payload;
    __device__ __forceinline__ LeafTri load(uint32_t t) const
    {
        LeafTri lt;
        lt.id = ids ? ids[t] : t;
        lt.v0 = vidx[3 * (size_t)t]; lt.v1 = vidx[3 * (size_t)t + 1]; lt.v2 = vidx[3 * (size_t)t + 2];
        return lt;
    }
    __device__ __forceinline__ void store(uint32_t j, uint32_t, const LeafTri &lt) const
    {
        leaf[j] = lt;
        if (!parent) return;
        parent[(n - 1) + j] = -1;
        if (j < n - 1) { parent[j] = -1; bounded[j] = 0; }
    }
};

__global__ __launch_bounds__(256) void k_fill_leaves(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ vidx,
                                                     const uint32_t *__restrict__ ids, uint32_t n,
                                                     LeafTri *__restrict__ leaf, int32_t *__restrict__ parent, uint32_t *__restrict__ bounded,
                                                     double *__restrict__ boxes /* nullptr: do not poison (fused path: the refit follows at once) */)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) fill_leaf(j, perm[j], vidx, ids, n, leaf, parent, bounded, boxes);
}

// Half-key sort: the fix-up hop (cd_sort.h) knows the final position of every triangle, so it fills the leaves too.
__global__ __launch_bounds__(256) void k_sort_fixup_fill(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                         uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n,
                                                         uint32_t *__restrict__ overflow,
                                                         const uint32_t *__restrict__ vidx, const uint32_t *__restrict__ ids,
                                                         LeafTri *__restrict__ leaf, int32_t *__restrict__ parent, uint32_t *__restrict__ bounded)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t k0 = keys_in[i];
    const uint32_t t = vals_in[i];
    uint32_t pos;
    // run too long: flag it (the host redoes the sort with 8 passes) but still emit a VALID permutation and valid
    // leaves -- the rest of the fused pipeline runs on this output before the host sees the flag
    if (!fixup_position(keys_in, n, i, k0, pos)) { atomicOr(overflow, SORTF_FIXUP); pos = i; }
    keys_out[pos] = k0;
    vals_out[pos] = t;
    fill_leaf(pos, t, vidx, ids, n, leaf, parent, bounded, nullptr);
}

// ---------------------------------------------------------------- delta / determineRange / findSplit
// bvh.cuh:48: delta(i,j) = clzll(k[i]^k[j]) for j in range, else -1.  Equal keys fall through to the
// index tie-break (64 + clz32(i^j)) -- identical to the reference whenever keys are unique (the only
// case in which the reference builds a valid tree, load_obj.h:109-115).
__device__ __forceinline__ int delta_k(const uint64_t *__restrict__ keys, int n, int i, uint64_t ki, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint64_t x = ki ^ keys[j];
    if (x) return __clzll((long long)x);
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// generateHierarchyParallel, bvh.cuh:146-199.  One thread per internal node.
__global__ __launch_bounds__(256) void k_hierarchy(const uint64_t *__restrict__ keys, int n,
                                                   NodeMeta *__restrict__ meta, int32_t *__restrict__ parent, uint32_t *__restrict__ parent_wrong)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const uint64_t ki = keys[i];
    // determineRange, bvh.cuh:100-123
    const int d = (delta_k(keys, n, i, ki, i + 1) - delta_k(keys, n, i, ki, i - 1)) >= 0 ? 1 : -1;
    const int delta_min = delta_k(keys, n, i, ki, i - d);
    int mlen = 2;
    while (delta_k(keys, n, i, ki, i + mlen * d) > delta_min) mlen <<= 1;
    int l = 0;
    for (int t = mlen >> 1; t >= 1; t >>= 1)
        if (delta_k(keys, n, i, ki, i + (l + t) * d) > delta_min) l += t;
    const int j = i + l * d;
    const int first = min(i, j), last = max(i, j);
    // findSplit, bvh.cuh:57-98
    const uint64_t kf = keys[first];
    const int common = delta_k(keys, n, first, kf, last);
    int split = first, step = last - first;
    do {
        step = (step + 1) >> 1;
        const int ns = split + step;
        if (ns < last) {
            if (delta_k(keys, n, first, kf, ns) > common) split = ns;
        }
    } while (step > 1);
    // bvh.cuh:174-195
    const int a = (split == first) ? (n - 1) + split : split;
    const int b = (split + 1 == last) ? (n - 1) + split + 1 : split + 1;
    meta[i] = make_int4(a, b, j, 0);
    if (atomicExch(&parent[a], i) != -1) atomicAdd(parent_wrong, 1u);     // bvh.cuh:192-195
    if (atomicExch(&parent[b], i) != -1) atomicAdd(parent_wrong, 1u);
}

// ---------------------------------------------------------------- calBoundingBox (bvh.cuh:258-285)
// One thread per leaf; the second thread to reach an internal node merges and continues.  The
// reference has no fence between a child's box store and the sibling path's read (a race); here the
// arrival counter is an agent-scope acq_rel RMW, which orders the first arriver's box store before
// the second arriver's load on any CU / XCD.
__device__ __forceinline__ void store_box(double *__restrict__ boxes, int node, const Box &b)
{
    double2 *p = reinterpret_cast<double2 *>(boxes + 6 * (size_t)node);
    p[0] = make_double2(b.x1, b.x2); p[1] = make_double2(b.y1, b.y2); p[2] = make_double2(b.z1, b.z2);
}
__device__ __forceinline__ Box load_box(const double *boxes, int node)
{
    const double2 *p = reinterpret_cast<const double2 *>(boxes + 6 * (size_t)node);
    const double2 a = p[0], b = p[1], c = p[2];
    return Box{a.x, a.y, b.x, b.y, c.x, c.y};
}

// fp32 traversal record of one internal node (64 bytes): both child boxes rounded outward, both child ids
// already encoded (internal i >= 0, leaf j -> ~j).
// A box whose six coordinates are fp32 values (the reference parses OBJ coordinates as float, load_obj.h:38, so that
// is the normal case): its outward-rounded fp32 copy IS the box, and an fp32 overlap test against another such box
// decides exactly what box.cuh:40-43 decides in FP64.
__device__ __forceinline__ bool box_is_fp32(const Box &b)
{
    return (double)(float)b.x1 == b.x1 && (double)(float)b.x2 == b.x2 && (double)(float)b.y1 == b.y1 &&
           (double)(float)b.y2 == b.y2 && (double)(float)b.z1 == b.z1 && (double)(float)b.z2 == b.z2;
}

__device__ __forceinline__ void store_rec32(NodeRec32 *__restrict__ recs, int n, uint32_t split, const Box &bl, const Box &br, int2 ch, uint32_t first, uint32_t last,
                                            const AmbTable &amb)
{
    float4 *pl = const_cast<float4 *>(rec_left(recs, n, split)), *pr = const_cast<float4 *>(rec_right(recs, n, split));
    const Enc32 el = enc_box32(amb, bl), er = enc_box32(amb, br);        // (the flags only matter where the child is a leaf)
    pl[0] = make_float4(el.lx, el.ly, el.lz, el.hx);
    pl[1] = make_float4(el.hy, el.hz, __int_as_float(ch.x),
                        __uint_as_float(first | ((el.certain && box_is_fp32(bl)) ? REC_L_EXACT : 0u) | ((er.certain && box_is_fp32(br)) ? REC_R_EXACT : 0u)));
    pr[0] = make_float4(er.lx, er.ly, er.lz, er.hx);
    pr[1] = make_float4(er.hy, er.hz, __int_as_float(ch.y), __uint_as_float(last | (el.certain ? REC_L_CERTAIN : 0u) | (er.certain ? REC_R_CERTAIN : 0u)));
}

// Link stored in a record for child `c` (unified Karras id): ~j for leaf j, the child's own split for an internal
// child -- the decoded left-child id of meta[c] (Karras: the left child's index IS the split position).
__device__ __forceinline__ int32_t child_link(const NodeMeta *__restrict__ meta, const int32_t *__restrict__ split_of, int c, int nleaf_base)
{
    if (c >= nleaf_base) return ~(c - nleaf_base);
    if (split_of) return split_of[c];                                        // fused build: the splits are an array of their own
    const int mx = meta[c].x;
    return mx >= nleaf_base ? mx - nleaf_base : mx;
}

// ---------------------------------------------------------------- calBoundingBox as RANGE QUERIES
// The reference climbs from every leaf with an (unfenced) atomic arrival counter per node, bvh.cuh:258-285.
// On MI355X that shape is a chain of ~20 dependent, atomically synchronised steps (a literal port with
// agent-scope ordering: 5 ms at 1 M; a three-phase LDS / workgroup-scope version: 250 us).  A Karras node i is
// nothing but the leaf range [first,last] it covers, and its box is the min/max over those leaves, so the
// refit is restated as order-preserving range queries over an implicit segment tree of the leaf boxes:
//   * comb(L, R) = box_merge(L, R) (box.cuh:24-32) with the LEFT range first.  fmin2/fmax2 return the right
//     operand on ties, so any left-to-right tree of merges yields the value of the rightmost extremal leaf --
//     bit-identical to the reference's merge(childA, childB) recursion whatever the tree shape (+-0 included).
//   * k_refit_seg_local: one workgroup per 512 consecutive leaves builds its 9-level segment tree in LDS
//     (also written to global memory), then thread t answers node b0+t: childA box = query(first, split),
//     childB box = query(split+1, last) -> the 64-byte traversal record, the node's box, bounded = 2.
//     No atomics, no dependent global loads, every node of the block in parallel.
//   * k_refit_seg_top: one workgroup builds the levels above the 512-leaf blocks.
//   * k_refit_seg_cross: the few nodes whose range leaves their block (O(N/512)) query the global tree.
constexpr int REFIT_BLK = 512;     // leaves per workgroup
constexpr int REFIT_LOG = 9;
constexpr int SEG_MIN_LEVEL = 3;   // lowest level of the segment tree that is stored in memory (see seg_piece)
#ifndef SEG32_MIN
#define SEG32_MIN 3
#endif
constexpr int SEG32_MIN_LEVEL = SEG32_MIN;   // the same for the fp32 trees of the fused build (seg32): its lower levels are 24 bytes a node, and every level
                                     // that is not stored is a dependent gather of 2^level leaf boxes on the cross nodes' critical path

__device__ __forceinline__ Box box_identity()
{
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    return Box{inf, -inf, inf, -inf, inf, -inf};
}

// Given the exact boxes of both children, write the node's 64-byte fp32 traversal record -- at its SPLIT -- and return
// the node's exact box (bvh.cuh:277 merge(childA, childB)).
__device__ __forceinline__ Box emit_node(const Box &bl, const Box &br, const NodeMeta *__restrict__ meta,
                                         int cl, int cr, int split, int first, int last, NodeRec32 *__restrict__ recs32, int nleaf_base, const AmbTable &amb)
{
    store_rec32(recs32, nleaf_base + 1, (uint32_t)split, bl, br, make_int2(child_link(meta, nullptr, cl, nleaf_base), child_link(meta, nullptr, cr, nleaf_base)),
                (uint32_t)first, (uint32_t)last, amb);
    return box_merge(bl, br);
}

__device__ __forceinline__ Box lds_box(const double (*t)[6], int k) { return Box{t[k][0], t[k][1], t[k][2], t[k][3], t[k][4], t[k][5]}; }

// Ordered range query [l, r] (inclusive, local leaf indices) over the block's LDS segment tree `t`
// (1-based heap layout, leaves at REFIT_BLK + j).
__device__ __forceinline__ Box seg_query_lds(const double (*t)[6], int l, int r)
{
    Box accL = box_identity(), accR = box_identity();
    l += REFIT_BLK; r += REFIT_BLK + 1;
    // (the level bound only matters for arguments outside [0, REFIT_BLK): a negative l would never reach r)
    for (int lev = 0; lev <= REFIT_LOG && l < r; ++lev) {
        if (l & 1) { accL = box_merge(accL, lds_box(t, l)); ++l; }
        if (r & 1) { --r; accR = box_merge(lds_box(t, r), accR); }
        l >>= 1; r >>= 1;
    }
    return box_merge(accL, accR);
}

// The fused build (cd_build.h) also BUILDS the hierarchy of the nodes whose range stays inside their 512 leaves (about 98 %
// of them) instead of reading it from k_hierarchy's meta[]: delta(i, j) of any leaf range is the minimum of the adjacent
// deltas dl[p] = delta(p, p+1) inside it (sorted keys, index tie-break), and adjacent deltas that bound a node are pairwise
// distinct, so determineRange (bvh.cuh:100-123) is a nearest-smaller-value query on dl and findSplit (bvh.cuh:57-98) the
// position of the range minimum -- both answered from a min-sparse-table over the block's deltas in LDS, where
// k_hierarchy's galloping / binary searches over the 64-bit keys in memory make a wave wait for its widest node (5.7 x
// the useful probes).  meta[] / parent[] (the reference's tree, what cd_export_tree and the verifier read) are then not
// written at all: the host materialises them with k_hierarchy when somebody asks (mi355cd.hip).
constexpr int DL_N = REFIT_BLK + 2;             // dl positions b0-1 .. b0+512
constexpr int DL_LEVELS = 10;                   // 2^9 = 512 < DL_N <= 2^10
constexpr int DL_STRIDE = 520;

__global__ __launch_bounds__(REFIT_BLK) void k_refit_seg_local(const double *__restrict__ verts, const LeafTri *__restrict__ leaf, int n,
                                                               const NodeMeta *__restrict__ meta,
                                                               double *__restrict__ boxes, uint32_t *__restrict__ bounded,
                                                               NodeRec32 *__restrict__ recs32, LeafBox32 *__restrict__ qbox32,
                                                               int32_t *__restrict__ root_name, int write_internal /* 0: FP64 boxes of the root and the leaves only */,
                                                               double *__restrict__ seg /* P x 6, heap order, node 0 unused */, int nbp2,
                                                               int32_t *__restrict__ cross_list /* cross_cap entries */, uint32_t *__restrict__ cross_count /* its length */,
                                                               uint32_t cross_cap, AmbTable amb, const uint8_t *__restrict__ vamb)
{
    __shared__ double t[2 * REFIT_BLK][6];          // 48 KB
    __shared__ int32_t lcross[REFIT_BLK];
    __shared__ uint32_t lcount, lbase;
    if (threadIdx.x == 0) lcount = 0;
    const int b = blockIdx.x, b0 = b * REFIT_BLK, tid = threadIdx.x;
    const int j = b0 + tid;
    Box mine = box_identity();
    if (j < n) {
        const LeafTri lt = leaf[j];
        const d3 A = load_vertex(verts, lt.v0), B = load_vertex(verts, lt.v1), C = load_vertex(verts, lt.v2);
        mine = box_set(A, B, C);                                           // box.cuh:13-22
        store_box(boxes, (n - 1) + j, mine);
        float4 *qp = reinterpret_cast<float4 *>(qbox32 + j);
        const Enc32 e = enc_leaf32(amb, mine, A, B, C, vamb, lt.v0, lt.v1, lt.v2);
        qp[0] = make_float4(e.lx, e.ly, e.lz, e.hx);
        qp[1] = make_float4(e.hy, e.hz, __uint_as_float(((e.certain && box_is_fp32(mine)) ? LB_EXACT : 0u) | (e.certain ? LB_CERTAIN : 0u) |
                                                        (box_overlap(mine, mine) ? LB_SELF : 0u)), 0.f);
    }
    {
        double *d = t[REFIT_BLK + tid];
        d[0] = mine.x1; d[1] = mine.x2; d[2] = mine.y1; d[3] = mine.y2; d[4] = mine.z1; d[5] = mine.z2;
    }
    // build the 9 levels above the leaves; global index of local node k at depth dd: ((nbp2 + b) << dd) + (k - 2^dd)
    for (int dd = REFIT_LOG - 1; dd >= 0; --dd) {
        __syncthreads();
        const int cnt = 1 << dd;
        if (tid < cnt) {
            const int k = cnt + tid;
            const Box m = box_merge(lds_box(t, 2 * k), lds_box(t, 2 * k + 1));
            double *d = t[k];
            d[0] = m.x1; d[1] = m.x2; d[2] = m.y1; d[3] = m.y2; d[4] = m.z1; d[5] = m.z2;
            if (REFIT_LOG - dd >= SEG_MIN_LEVEL) store_box(seg, (int)((((size_t)nbp2 + b) << dd) + tid), m);   // cross queries rebuild the lowest levels from leaves
        }
    }
    __syncthreads();
    const int i = j;                                                   // internal node with the same index
    int first = 0, last = 0, split = 0; bool have = false;
    if (i < n - 1) {
        const NodeMeta m = meta[i];
        first = min(i, m.z); last = max(i, m.z);
        if (first < b0 || last >= b0 + REFIT_BLK) {
            lcross[atomicAdd(&lcount, 1u)] = i;                        // leaves the block: k_refit_seg_cross
        } else { have = true; split = (m.x >= n - 1) ? m.x - (n - 1) : m.x; }    // childA covers [first, split], childB [split+1, last]
    }
    if (have) {
        // children as unified Karras ids (bvh.cuh:174-195): the left child is leaf `split` or internal node `split`
        const int ca = (split == first) ? (n - 1) + split : split, cb = (split + 1 == last) ? (n - 1) + split + 1 : split + 1;
        const Box bl = seg_query_lds(t, first - b0, split - b0);
        const Box br = seg_query_lds(t, split + 1 - b0, last - b0);
        const Box whole = emit_node(bl, br, meta, ca, cb, split, first, last, recs32, n - 1, amb);
        // The FP64 boxes of internal nodes are the OUTPUT of calBoundingBox (bvh.cuh:277): written on request
        if (write_internal || i == 0) store_box(boxes, i, whole);
        if (i == 0) *root_name = split;
        bounded[i] = 2;                                                // Node::bounded (bvh.cuh:270): both children merged
    }
    __syncthreads();
    // hand the block's cross nodes over: ONE global atomic per workgroup on the list's length
    const uint32_t cnt = lcount;
    if (cnt == 0) return;
    if (tid == 0) lbase = atomicAdd(cross_count, cnt);
    __syncthreads();
    if ((uint32_t)tid < cnt && lbase + tid < cross_cap) cross_list[lbase + tid] = lcross[tid];
}

// Levels above the 512-leaf blocks: heap nodes [1, nbp2).  One workgroup; children at or beyond the last real
// block are the identity.  (nbp2 = blocks rounded up to a power of two.)  A level is a dependent step, so the
// levels with at most TOP_LDS nodes live in LDS (a workgroup barrier + LDS latency per level instead of an L2 round
// trip): at 1 M triangles that is all 11 of them; wider levels of larger inputs go through memory first.
constexpr int TOP_LDS = 1024;
__global__ __launch_bounds__(1024) void k_refit_seg_top(double *seg, int nbp2, int nblocks)
{
    __shared__ double t[2 * TOP_LDS][6];                                // heap slots [1, 2 TOP_LDS): 96 KB of the CU's 160 KB
    int cnt = nbp2 >> 1;
    for (; cnt > TOP_LDS; cnt >>= 1) {                                  // level with `cnt` nodes: k in [cnt, 2 cnt)
        for (int u = threadIdx.x; u < cnt; u += blockDim.x) {
            const int k = cnt + u;
            // a child 2k / 2k+1 on the block level (>= nbp2) exists only if its block index < nblocks
            const int c0 = 2 * k, c1 = 2 * k + 1;
            const bool blocklevel = c0 >= nbp2;
            const Box L = (blocklevel && c0 - nbp2 >= nblocks) ? box_identity() : load_box(seg, c0);
            const Box R = (blocklevel && c1 - nbp2 >= nblocks) ? box_identity() : load_box(seg, c1);
            store_box(seg, k, box_merge(L, R));
        }
        __syncthreads();                                                // workgroup-scope: the next level reads these
    }
    if (cnt < 1) return;
    {   // first LDS level: children still come from memory
        const int u = threadIdx.x;
        if (u < cnt) {
            const int k = cnt + u, c0 = 2 * k, c1 = 2 * k + 1;
            const bool blocklevel = c0 >= nbp2;
            const Box L = (blocklevel && c0 - nbp2 >= nblocks) ? box_identity() : load_box(seg, c0);
            const Box R = (blocklevel && c1 - nbp2 >= nblocks) ? box_identity() : load_box(seg, c1);
            const Box m = box_merge(L, R);
            double *d = t[k];
            d[0] = m.x1; d[1] = m.x2; d[2] = m.y1; d[3] = m.y2; d[4] = m.z1; d[5] = m.z2;
            store_box(seg, k, m);
        }
    }
    for (cnt >>= 1; cnt >= 1; cnt >>= 1) {
        __syncthreads();
        const int u = threadIdx.x;
        if (u < cnt) {
            const int k = cnt + u;
            const Box m = box_merge(lds_box(t, 2 * k), lds_box(t, 2 * k + 1));
            double *d = t[k];
            d[0] = m.x1; d[1] = m.x2; d[2] = m.y1; d[3] = m.y2; d[4] = m.z1; d[5] = m.z2;
            store_box(seg, k, m);
        }
    }
}

// Box of heap node k at level p (2^p leaves).  Levels below SEG_MIN_LEVEL are not stored (three quarters of the tree's
// bytes for a handful of reads): such a piece is merged from its 1, 2 or 4 leaf boxes, left to right.
__device__ __forceinline__ Box seg_piece(const double *__restrict__ seg, const double *__restrict__ boxes, int n, long long P, long long k, int p)
{
    if (p >= SEG_MIN_LEVEL) return load_box(seg, (int)k);
    const long long j0 = (k << p) - P;
    Box x = box_identity();
    for (int u = 0; u < (1 << p); ++u) { const long long j = j0 + u; if (j < n) x = box_merge(x, load_box(boxes, (n - 1) + (int)j)); }
    return x;
}

// Ordered range query [l0, r0] (inclusive leaf positions) over the global tree by ONE WAVE: the iterative
// bottom-up query takes at most one left piece and one right piece per level; lane p < 32 owns the left piece
// of level p, lane 32+q the right piece of level 31-q, so lane order == left-to-right order of the pieces.
// Every lane loads its piece (all loads in flight together), then an order-preserving shuffle reduction
// (lower lanes are the LEFT operand) combines them.  Internal heap nodes come from `seg` (P = nbp2*512
// leaves), level-0 pieces from boxes[(n-1)+j].  Returns the result in every lane.
__device__ __forceinline__ Box seg_query_wave(const double *__restrict__ seg, const double *__restrict__ boxes, int n, long long P, int l0, int r0, int lane)
{
    const bool is_left = lane < 32;
    const int p = is_left ? lane : 63 - lane;                              // level of this lane's piece
    const long long lp = ((long long)l0 + P + ((1ll << p) - 1)) >> p;      // l at level p  (ceil)
    const long long rp = ((long long)r0 + P + 1) >> p;                     // r at level p  (floor), half-open
    Box x = box_identity();
    if (lp < rp) {
        const long long k = is_left ? lp : rp - 1;
        const bool take = is_left ? (lp & 1) : (rp & 1);
        if (take) {
            x = seg_piece(seg, boxes, n, P, k, p);
        }
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {                                      // after the step, lanes that are multiples of 2s hold [lane, lane+2s)
        Box y;
        y.x1 = __shfl_down(x.x1, s); y.x2 = __shfl_down(x.x2, s); y.y1 = __shfl_down(x.y1, s);
        y.y2 = __shfl_down(x.y2, s); y.z1 = __shfl_down(x.z1, s); y.z2 = __shfl_down(x.z2, s);
        x = box_merge(x, y);                                               // mine is LEFT of the one s lanes up
    }
    Box r;
    r.x1 = __shfl(x.x1, 0); r.x2 = __shfl(x.x2, 0); r.y1 = __shfl(x.y1, 0);
    r.y2 = __shfl(x.y2, 0); r.z1 = __shfl(x.z1, 0); r.z2 = __shfl(x.z2, 0);
    return r;
}

// One step of the ordered reduction: x = merge(x, x of the lane `shift` to the right inside the 16-lane row).
template <int CTRL>
__device__ __forceinline__ double dpp_row_shl(double v, double fill)
{
    const long long b = __double_as_longlong(v), f = __double_as_longlong(fill);
    const int lo = __builtin_amdgcn_update_dpp((int)f, (int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL>
__device__ __forceinline__ Box dpp_step(const Box &x)
{
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    Box y;
    y.x1 = dpp_row_shl<CTRL>(x.x1, inf); y.x2 = dpp_row_shl<CTRL>(x.x2, -inf); y.y1 = dpp_row_shl<CTRL>(x.y1, inf);
    y.y2 = dpp_row_shl<CTRL>(x.y2, -inf); y.z1 = dpp_row_shl<CTRL>(x.z1, inf); y.z2 = dpp_row_shl<CTRL>(x.z2, -inf);
    return box_merge(x, y);                                                // mine is LEFT of the one to the right
}

// Both child queries of a node in ONE pass: lanes 0-31 answer [la, ra], lanes 32-63 answer [lb, rb].  Inside a half,
// lane h < 16 owns the left piece of level h and lane 31 - q the right piece of level q, so lane order is again
// left-to-right order; 16 levels cover every range shorter than 65536 leaves (the caller checks).  One 5-step
// segmented reduction instead of two 6-step ones: the shuffles (ds_bpermute) are what this kernel issues most.
// Returns [la, ra] in lane 0 and [lb, rb] in lane 32.
__device__ __forceinline__ Box seg_query_halves(const double *__restrict__ seg, const double *__restrict__ boxes, int n, long long P,
                                                int la, int ra, int lb, int rb, int lane)
{
    const int hl = lane & 31;
    const bool second = lane >= 32, is_left = hl < 16;
    const int l0 = second ? lb : la, r0 = second ? rb : ra;
    const int p = is_left ? hl : 31 - hl;                                  // level of this lane's piece
    const long long lp = ((long long)l0 + P + ((1ll << p) - 1)) >> p;      // l at level p  (ceil)
    const long long rp = ((long long)r0 + P + 1) >> p;                     // r at level p  (floor), half-open
    Box x = box_identity();
    if (lp < rp) {
        const long long k = is_left ? lp : rp - 1;
        const bool take = is_left ? (lp & 1) : (rp & 1);
        if (take) {
            x = seg_piece(seg, boxes, n, P, k, p);
        }
    }
    // Steps 1, 2, 4, 8 stay inside a row of 16 lanes: DPP row_shl (a VALU move, no LDS traffic); a lane whose source
    // would be outside its row keeps the identity -- only lanes whose result is never consumed are affected.
    x = dpp_step<0x101>(x); x = dpp_step<0x102>(x); x = dpp_step<0x104>(x); x = dpp_step<0x108>(x);
    {                                                                       // step 16: lanes 0 and 32 pull rows 1 and 3 of their half
        Box y;
        y.x1 = __shfl_down(x.x1, 16); y.x2 = __shfl_down(x.x2, 16); y.y1 = __shfl_down(x.y1, 16);
        y.y2 = __shfl_down(x.y2, 16); y.z1 = __shfl_down(x.z1, 16); y.z2 = __shfl_down(x.z2, 16);
        x = box_merge(x, y);                                               // consumed only in lanes 0 and 32, whose source is their own half
    }
    return x;
}

// Fused build: range and split of the cross nodes (the ~2 % whose range leaves their 512-leaf block), by the searches of
// generateHierarchyParallel (bvh.cuh:146-199) -- each of which looks for the last position at which a monotone predicate
// on delta still holds.  A few ten thousand nodes on a whole chip is pure latency: what counts is how many of these
// dependent chains are in flight and how many round trips each takes.  So a GROUP of 16 lanes takes a node -- four nodes
// per wave, every node of a 1 M-triangle tree in flight at once -- and every round probes 16 positions (all doublings of
// the galloping phase in one round, then 16 cut points per round): the same answers in 5-7 dependent round trips
// instead of 40-60.  (One wave per node, 64 probes per round: 13.3 us; one thread per node: 23 us.)
// Writes meta[i] (what the cross-record kernels read) and split_of[i] (child links).
constexpr int XG = 16;                                                      // lanes per cross node
// this lane's group's 16 bits of a wave-wide ballot
__device__ __forceinline__ uint32_t group_ballot(bool c, int g) { return (uint32_t)(__builtin_amdgcn_ballot_w64(c) >> (g * XG)) & ((1u << XG) - 1u); }
// largest x in [lo, hi) with pred(x), given pred(lo) and !pred(hi); lo, hi are uniform inside a group (pred is monotone:
// true ... true false ... false); groups of a wave run their rounds together until the last one is done
template <class Pred>
__device__ __forceinline__ int group_last_true(bool live, int lo, int hi, int g, int gl, Pred pred)
{
    for (;;) {
        const bool more = live && hi - lo > 1;
        if (!__builtin_amdgcn_ballot_w64(more)) break;                      // (wave-uniform)
        const int w = hi - lo;
        const int step = (w + XG - 1) / XG;
        const long long x = (long long)lo + (long long)gl * step;
        const bool ok = more && x < hi && (gl == 0 || pred((int)x));
        const uint32_t m = group_ballot(ok, g);                             // a prefix of the group's lanes (bit 0 set while `more`)
        if (more) {
            const int top = 31 - __clz((int)m);
            const int nlo = lo + top * step;
            hi = (nlo + step < hi) ? nlo + step : hi;
            lo = nlo;
        }
    }
    return lo;
}

// The levels above the 512-leaf blocks (heap nodes [1, nbp2)) for trees of up to TOP_IN_BLOCK blocks, by ONE workgroup
// of 256 threads without an LDS tree: a thread folds its nbp2 / 256 consecutive block boxes in registers, the waves fold
// across lanes with shuffles (lower lane = LEFT operand: the order of box.cuh:24-32's ties is kept), four values cross
// the waves through LDS.  Every node is stored: the cross nodes' queries read them.  Called by block 0 of k_cross_meta,
// which has nothing to do with it -- a kernel of its own costs ~6 us for this microsecond of work (k_refit_seg_top).
constexpr int TOP_IN_BLOCK = 2048;
__device__ __forceinline__ Box box_shfl_down(const Box &x, int s)
{
    return Box{__shfl_down(x.x1, s), __shfl_down(x.x2, s), __shfl_down(x.y1, s), __shfl_down(x.y2, s), __shfl_down(x.z1, s), __shfl_down(x.z2, s)};
}
// b0, span: the workgroup folds the span (<= TOP_IN_BLOCK, a power of two) blocks that start at block b0, up to their common
// ancestor.  Heap node over the blocks [b, b + 2^l) = (nbp2 + b) >> l: the whole tree for span == nbp2; for a larger tree one
// workgroup per span of 2048 blocks, then the same function again on the heap's upper part (k_top_levels, below).
__device__ __forceinline__ void top_tree_one_block(double *__restrict__ seg, int nbp2, int nblocks, int b0, int span)
{
    __shared__ double wbox[4][6];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (span < 2) return;
    const int T = span < 256 ? span : 256;                                  // threads that own block boxes
    const int per = span / T;                                               // 1, 2, 4 or 8 consecutive blocks per thread
    const int kt = (nbp2 + b0) / per + tid;                                 // heap node of the thread's blocks
    Box x = box_identity();
    if (tid < T) {
        Box v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int b = b0 + tid * per + u; v[u] = (u < per && b < nblocks) ? load_box(seg, nbp2 + b) : box_identity(); }
        // in-thread levels: width 8 -> 4 -> 2 -> 1 of the thread's `per` boxes
        int width = per;
#pragma unroll
        for (int l = 1; l <= 3; ++l) {
            if ((1 << l) > per) break;
            width >>= 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (u < width) { v[u] = box_merge(v[2 * u], v[2 * u + 1]); store_box(seg, ((nbp2 + b0 + tid * per) >> l) + u, v[u]); }
        }
        x = v[0];
    }
    // across lanes: after the step with stride s, lanes that are multiples of 2s hold node kt / (2s)
    for (int s = 1; s < 64 && s < T; s <<= 1) {
        const Box y = box_shfl_down(x, s);
        x = box_merge(x, y);
        if (tid < T && (lane & (2 * s - 1)) == 0) store_box(seg, kt / (2 * s), x);
    }
    if (T <= 64) return;
    if (lane == 0) { double *d = wbox[w]; d[0] = x.x1; d[1] = x.x2; d[2] = x.y1; d[3] = x.y2; d[4] = x.z1; d[5] = x.z2; }
    __syncthreads();
    if (tid == 0) {                                                         // the T / 64 = 2 or 4 wave results are the heap nodes kt / 64 .. + T / 64 - 1
        auto wb = [&](int k) { return Box{wbox[k][0], wbox[k][1], wbox[k][2], wbox[k][3], wbox[k][4], wbox[k][5]}; };
        const int kw = kt / 64;                                             // (tid == 0: the first of them)
        if (T == 128) store_box(seg, kw / 2, box_merge(wb(0), wb(1)));
        else {
            const Box l = box_merge(wb(0), wb(1)), r = box_merge(wb(2), wb(3));
            store_box(seg, kw / 2, l); store_box(seg, kw / 2 + 1, r); store_box(seg, kw / 4, box_merge(l, r));
        }
    }
}

// Trees of more than TOP_IN_BLOCK blocks: one workgroup per span of TOP_IN_BLOCK blocks; launched again with
// (nbp2 / TOP_IN_BLOCK, ceil(nblocks / TOP_IN_BLOCK)) it folds the span roots -- they ARE the block level of the heap's
// upper part (same array, same indices).  (k_refit_seg_top's single workgroup takes 65 us at 8 M triangles.)
__global__ __launch_bounds__(256) void k_top_levels(double *__restrict__ seg, int nbp2, int nblocks)
{
    const int span = nbp2 < TOP_IN_BLOCK ? nbp2 : TOP_IN_BLOCK;
    top_tree_one_block(seg, nbp2, nblocks, (int)blockIdx.x * span, span);
}

__global__ __launch_bounds__(256) void k_cross_meta(const uint64_t *__restrict__ keys, int n, NodeMeta *__restrict__ meta, int32_t *__restrict__ split_of,
                                                    const int32_t *__restrict__ dense, const uint32_t *__restrict__ dense_total, uint32_t dense_cap,
                                                    double *__restrict__ seg /* non-NULL: block 0 also builds the levels above the blocks */, int nbp2, int nblocks)
{
    if (seg && blockIdx.x == 0) top_tree_one_block(seg, nbp2, nblocks, 0, nbp2);   // (workgroup-uniform)
    const int tid = threadIdx.x, lane = tid & 63, g = lane / XG, gl = lane % XG;
    constexpr int PER_WAVE = 64 / XG, PER_BLOCK = 256 / XG;
    const uint32_t total = min(*dense_total, dense_cap);
    for (uint32_t wbase = blockIdx.x * PER_BLOCK + (tid >> 6) * PER_WAVE; wbase < total; wbase += gridDim.x * PER_BLOCK) {   // (wbase is wave-uniform)
        const uint32_t k = wbase + g;
        const bool live = k < total;
        const int i = live ? dense[k] : 0;
        const uint64_t ki = keys[i];
        const int d = (delta_k(keys, n, i, ki, i + 1) - delta_k(keys, n, i, ki, i - 1)) >= 0 ? 1 : -1;
        const int delta_min = delta_k(keys, n, i, ki, i - d);
        // galloping phase: lane L of the group asks about 2^(L+1), in a second round about 2^(L+17); the first "no" is mlen
        // (a position outside the keys says no)
        int mlen = 0;
        for (int r = 0; r < 2; ++r) {
            const bool todo = live && mlen == 0;
            if (!__builtin_amdgcn_ballot_w64(todo)) break;                  // (wave-uniform)
            const int e = r * XG + gl;                                      // exponent - 1
            const long long o = (long long)i + (long long)d * (2ll << (e < 31 ? e : 31));
            const bool yes = todo && e < 31 && o >= 0 && o < n && delta_k(keys, n, i, ki, (int)o) > delta_min;
            const uint32_t no = ~group_ballot(yes, g) & ((1u << XG) - 1u);
            if (todo && no) mlen = 2 << (r * XG + __ffs((int)no) - 1);
        }
        const int l = group_last_true(live, mlen >> 1, mlen, g, gl, [&](int x) { return delta_k(keys, n, i, ki, i + x * d) > delta_min; });
        const int j = i + l * d;
        const int first = min(i, j), last = max(i, j);
        const uint64_t kf = keys[first];
        const int common = delta_k(keys, n, first, kf, last);
        // findSplit: the last s in [first, last) with delta(first, s) > common (s == first counts as yes)
        const int split = group_last_true(live, first, last, g, gl, [&](int x) { return delta_k(keys, n, first, kf, x) > common; });
        if (live && gl == 0) {
            const int a = (split == first) ? (n - 1) + split : split;
            const int b = (split + 1 == last) ? (n - 1) + split + 1 : split + 1;
            meta[i] = make_int4(a, b, j, 0);
            split_of[i] = split;
        }
    }
}

// Nodes whose range leaves their 512-leaf block: one WAVE per node, its two range queries side by side in the two
// halves of the wave (seg_query_halves; ranges of 65536 leaves or more take two full-wave queries).
// Queries read only leaf boxes and segment-tree nodes, never another cross node's output: no ordering needed.
__global__ __launch_bounds__(256) void k_refit_seg_cross(int n, const NodeMeta *__restrict__ meta, const double *__restrict__ seg, int nbp2,
                                                         double *boxes, uint32_t *__restrict__ bounded, NodeRec32 *__restrict__ recs32,
                                                         int32_t *__restrict__ root_name, int write_internal,
                                                         const int32_t *__restrict__ dense, const uint32_t *__restrict__ dense_total, uint32_t dense_cap, AmbTable amb)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t total = min(*dense_total, dense_cap);
    const long long P = (long long)nbp2 * REFIT_BLK;
    for (uint32_t k = blockIdx.x * 4 + (tid >> 6); k < total; k += gridDim.x * 4) {   // one wave per node (k is wave-uniform)
        const int i = __builtin_amdgcn_readfirstlane(dense[k]);
        const NodeMeta m = meta[i];
        const int first = min(i, m.z), last = max(i, m.z);
        const int split = (m.x >= n - 1) ? m.x - (n - 1) : m.x;
        Box bl, br;
        if (last - first < 65535) {                                         // (wave-uniform) nearly all of them
            const Box h = seg_query_halves(seg, boxes, n, P, first, split, split + 1, last, lane);
            bl = h;                                                         // valid in lane 0, the only lane that stores
            br.x1 = __shfl(h.x1, 32); br.x2 = __shfl(h.x2, 32); br.y1 = __shfl(h.y1, 32);
            br.y2 = __shfl(h.y2, 32); br.z1 = __shfl(h.z1, 32); br.z2 = __shfl(h.z2, 32);
        } else {
            bl = seg_query_wave(seg, boxes, n, P, first, split, lane);
            br = seg_query_wave(seg, boxes, n, P, split + 1, last, lane);
        }
        if (lane == 0) {
            const Box whole = emit_node(bl, br, meta, m.x, m.y, split, first, last, recs32, n - 1, amb);
            if (write_internal || i == 0) store_box(boxes, i, whole);
            if (i == 0) *root_name = split;
            bounded[i] = 2;
        }
    }
}

// ---------------------------------------------------------------- verifier counters (check.cuh)
constexpr uint64_t BOX_UNINIT_BITS = 0xFFFFFFFFFFFFFFFFull;   // x1 of every box is poisoned by k_fill_leaves until the refit writes it

// checkInternalNodes, check.cuh:64-79: out[0]=nullParent out[1]=wrongBound out[2]=nullChild out[3]=notInternal out[4]=uninitBox
__global__ __launch_bounds__(256) void k_check_internal(int n, const NodeMeta *__restrict__ meta, const int32_t *__restrict__ parent,
                                                        const uint32_t *__restrict__ bounded, const double *__restrict__ boxes,
                                                        uint32_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    if (bounded[i] != 2) atomicAdd(&out[1], 1u);
    const NodeMeta m = meta[i];
    if (parent[i] == -1) atomicAdd(&out[0], 1u);
    const int2 ch = make_int2(m.x, m.y);
    if (ch.x == -1) atomicAdd(&out[2], 1u);
    if (ch.y == -1) atomicAdd(&out[2], 1u);
    if (ch.x >= 2 * n - 1 || ch.y >= 2 * n - 1 || ch.x < -1 || ch.y < -1) atomicAdd(&out[3], 1u);
    if (reinterpret_cast<const uint64_t *>(boxes)[6 * (size_t)i] == BOX_UNINIT_BITS) atomicAdd(&out[4], 1u);
}
// checkLeafNodes, check.cuh:81-96: out[0]=nullParent out[1]=nullTriangle(+selfCheck) out[2]=notLeaf out[3]=illegalBox
__global__ __launch_bounds__(256) void k_check_leaves(int n, const int32_t *__restrict__ parent, const LeafTri *__restrict__ leaf,
                                                      uint32_t maxv, const double *__restrict__ boxes, uint32_t *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int node = (n - 1) + j;
    if (parent[node] == -1) atomicAdd(&out[0], 1u);
    const LeafTri lt = leaf[j];
    if (lt.v0 >= maxv || lt.v1 >= maxv || lt.v2 >= maxv) atomicAdd(&out[1], 1u);   // triangle.cuh:11-16
    if (reinterpret_cast<const uint64_t *>(boxes)[6 * (size_t)node] == BOX_UNINIT_BITS) atomicAdd(&out[3], 1u);
}
// checkTriangleIdx, check.cuh:29-50
__global__ __launch_bounds__(256) void k_check_triangle_idx(int n, const LeafTri *__restrict__ leaf, uint32_t maxv, uint32_t *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const LeafTri lt = leaf[j];
    const uint32_t c = (lt.v0 >= maxv) + (lt.v1 >= maxv) + (lt.v2 >= maxv);
    if (c) atomicAdd(out, c);
}

}  // namespace cd
