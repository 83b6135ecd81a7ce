// cd_build.h -- the FUSED build of the fused entry points (cd_self_collide / cd_build_tree / cd_multi_step): Karras
// hierarchy (bvh.cuh:100-199) + AABB refit (bvh.cuh:258-285) + the fp32 traversal records, straight from the sorted keys.
//
// What the traversal reads of an internal node is its 64-byte record: both child boxes as fp32 copies (cd_bvh.h: lo rounded
// down; hi rounded down, one ulp up where its cell is ambiguous).  Both maps are monotone, so they commute with min / max:
// the fp32 copy of a leaf range's box is the min / max of the leaves' fp32 copies -- bit for bit what encoding the FP64
// merge (bvh.cuh:277) gives.  The fused build therefore
// keeps ONE fp32 segment tree of the leaf boxes per 512-leaf block (hardware v_min_f32 / v_max_f32, 6 instructions a
// merge against 18 for the FP64 compare-selects of box.cuh:24-32) and never forms an internal FP64 box, except
//   * the FP64 boxes of the leaves that are NOT exact in fp32 (k_exact and cd_pack_queries decide their candidates in
//     FP64; an exact leaf's fp32 box is its FP64 box: leaf_box64, cd_bvh.h), and
//   * the FP64 box of ALL leaves (node 0: cd_root_box, the multi-GPU root exchange), reduced per block from the few leaves
//     whose fp32 value equals the block's fp32 extreme (only they can hold the FP64 extreme), then by k_refit_seg_top.
// The CERTAIN / EXACT flags of a record only matter for LEAF children -- a candidate is a pair of leaves --
// and a leaf child's box is the leaf's box, whose flags the leaf's thread knows.
// The reference's tree (meta[] / parent[] / FP64 boxes of internal nodes) is materialised by k_hierarchy + the stage-wise
// refit when somebody asks for it (cd_export_tree, the verifier, traversal variant 0): mi355cd.hip.
#pragma once
#include "cd_bvh.h"

namespace cd {

struct B32 { float lx, ly, lz, hx, hy, hz; };

// (inline asm: fminf / fmaxf on loaded values cost an extra canonicalising v_max_f32 x, x each under IEEE mode)
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ B32 b32_identity()
{
    const float inf = __uint_as_float(0x7f800000u);
    return B32{inf, inf, inf, -inf, -inf, -inf};
}
__device__ __forceinline__ B32 b32_merge(const B32 &a, const B32 &b)
{
    return B32{hw_min(a.lx, b.lx), hw_min(a.ly, b.ly), hw_min(a.lz, b.lz), hw_max(a.hx, b.hx), hw_max(a.hy, b.hy), hw_max(a.hz, b.hz)};
}
__device__ __forceinline__ B32 b32_of(const AmbTable &amb, const Box &b)  // the fp32 copy of an FP64 box (cd_bvh.h enc_box32: what store_rec32 does)
{
    const Enc32 e = enc_box32(amb, b);
    return B32{e.lx, e.ly, e.lz, e.hx, e.hy, e.hz};
}
__device__ __forceinline__ B32 b32_load(const float *p)                    // 24-byte node, 8-byte aligned
{
    const float2 a = reinterpret_cast<const float2 *>(p)[0], b = reinterpret_cast<const float2 *>(p)[1], c = reinterpret_cast<const float2 *>(p)[2];
    return B32{a.x, a.y, b.x, b.y, c.x, c.y};
}
__device__ __forceinline__ void b32_store(float *p, const B32 &v)
{
    reinterpret_cast<float2 *>(p)[0] = make_float2(v.lx, v.ly); reinterpret_cast<float2 *>(p)[1] = make_float2(v.lz, v.hx);
    reinterpret_cast<float2 *>(p)[2] = make_float2(v.hy, v.hz);
}
__device__ __forceinline__ B32 b32_of_leaf(const LeafBox32 *__restrict__ qbox32, int j)
{
    const float4 a = reinterpret_cast<const float4 *>(qbox32 + j)[0];
    const float2 b = reinterpret_cast<const float2 *>(qbox32 + j)[2];
    return B32{a.x, a.y, a.z, a.w, b.x, b.y};
}

// Range query [l, r] (inclusive, local leaf indices; l > r: nothing) over the block's fp32 segment tree in LDS (1-based
// heap, leaves at REFIT_BLK + j; slot 0 holds the identity).  min / max commute, so one accumulator takes the left and
// the right pieces as they come.  Written without branches: a level that contributes no piece reads slot 0, and the
// loop ends when the widest range of the WAVE is done.
__device__ __forceinline__ B32 seg_query32(const float (*t)[6], int l, int r)
{
    B32 acc = b32_identity();
    l += REFIT_BLK; r += REFIT_BLK + 1;
    for (int lev = 0; lev <= REFIT_LOG; ++lev) {
        const bool on = l < r;
        if (!__builtin_amdgcn_ballot_w64(on)) break;
        const int il = (on & ((l & 1) != 0)) ? l : 0, ir = (on & ((r & 1) != 0)) ? r - 1 : 0;
        acc = b32_merge(acc, b32_load(t[il]));
        acc = b32_merge(acc, b32_load(t[ir]));
        l = (l + 1) >> 1; r >>= 1;
    }
    return acc;
}

// The adjacent deltas of a block and their min-sparse-table as 16-bit keys: T[k][x] = (m + 1) << 9 | o, where m is the
// minimum of delta over the positions [x, x + 2^k - 1] and x + o the leftmost position that has it (o < 2^k <= 512;
// delta + 1 <= 96: bvh.cuh:48 with the index tie-break of equal keys) -- the minimum of a range AND where it is, in
// one value whose order is the order of the minima.
constexpr int DK_SHIFT = 9;
typedef uint16_t DKey;
constexpr DKey DK_PAD = 0xffffu;
static_assert(DL_LEVELS - 1 <= DK_SHIFT && ((64 + 32 + 1) << DK_SHIFT) < 0xffff, "key layout");
// Nearest position with a delta below thr, walking right from s (first p >= s, or DL_N) or left from s (last p <= s,
// or -1): a 10-step descent over the table, the same instructions for both directions.
__device__ __forceinline__ int nsv_dir(const DKey (*T)[DL_STRIDE], bool right, int s, int thr)
{
    const int tk = thr << DK_SHIFT;
    int p = s;
#pragma unroll
    for (int k = DL_LEVELS - 1; k >= 0; --k) {
        const int q = right ? p + (1 << k) : p - (1 << k);
        const int idx = right ? p : q + 1;                                 // the 2^k positions between p and q
        const bool inside = right ? (q <= DL_N) : (q >= -1);
        const DKey v = T[k][(idx >= 0 && idx < DL_N) ? idx : 0];
        p = (inside & (v >= (DKey)tk)) ? q : p;
    }
    return p;
}

// Doubles as unsigned integers of the same order (-0 below +0), for the LDS min / max of the block's FP64 extremes.
__device__ __forceinline__ unsigned long long f64_ordered(double d)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_from_ordered(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

// One workgroup per 512 consecutive leaves.  Thread t: leaf b0+t (FP64 box from the vertices -> boxes[] when it is not
// exact in fp32, fp32 query box -> qbox32[], fp32 leaf of the LDS tree) and internal node b0+t:
//   1. adjacent deltas of the block as 16-bit (min, offset) keys + their min-sparse-table, and the fp32 segment tree, on
//      the same 9 barriers;
//   2. determineRange as a nearest-smaller-value search on the table (one 10-step descent, either direction), findSplit
//      as the position of the range minimum (two table reads); a node whose search runs off the block goes on the cross
//      list (k_cross_meta, k_cross_records);
//   3. ONE range query per node: its own box -> nb[] in LDS (over the table, which is dead by then); after a barrier its
//      record takes the two child boxes from nb[] (children of an in-block node are in-block) or from the tree's leaves.
// Levels >= SEG32_MIN_LEVEL of the block's tree go to seg32 (what the cross nodes query), the FP64 box of the block's
// leaves to seg[nbp2 + b] (the levels above the blocks fold those into the box of all leaves: cd_bvh.h, top_tree_one_block).
// A fused step zeroes its scratch inside its own kernels instead of with a memset in front of the pipeline (a launch and
// ~3 us of idle GPU per step): the sort's histograms, tickets and look-back granules are dead once the sort is done, so
// k_build_block clears them for the NEXT step (the sort flags are not touched: a non-zero flag makes the host redo the
// step, with a memset), and the traversal counters for THIS step.  All pointers null: nothing to do.
struct ZeroPlan { uint32_t *w0; uint32_t nw0; uint4 *q; uint32_t nq; uint32_t *w1; uint32_t nw1;
                  uint32_t *flag; };    // one more word, cleared in every launch: k_cross_fused's "upper levels published" flag (a graph replay carries the same sequence number every time)

__global__ __launch_bounds__(REFIT_BLK) void k_build_block(const double *__restrict__ verts, const LeafTri *__restrict__ leaf, int n,
                                                           const uint64_t *__restrict__ keys, int32_t *__restrict__ split_of,
                                                           double *__restrict__ boxes, NodeRec32 *__restrict__ recs32, LeafBox32 *__restrict__ qbox32,
                                                           int32_t *__restrict__ root_name, double *__restrict__ seg, float *__restrict__ seg32, int nbp2,
                                                           int32_t *__restrict__ cross_list, uint32_t *__restrict__ cross_count, uint32_t cross_cap, ZeroPlan zp,
                                                           int seg_min /* lowest level of the block's tree that goes to seg32: SEG32_MIN_LEVEL for the trees k_cross_fused takes,
                                                                          SEG_MIN_LEVEL beyond (there the kernel is bound by its writes, and levels 1 and 2 are 18 bytes a leaf) */,
                                                           AmbTable amb, const uint8_t *__restrict__ vamb,
                                                           const uint32_t *__restrict__ hint_perm, const uint8_t *__restrict__ hint_tri, uint32_t *__restrict__ hint_cost /* all NULL, or the
                                                                          half traversal's order hint (cd_bvh.h): sorted position -> triangle, the time class the last traversal left with
                                                                          every triangle; out: per group of 64 leaves -- one wave of this kernel -- the max over its leaves */,
                                                           unsigned long long *__restrict__ leaf_side /* one bit a leaf: it is the LEFT child of recs[j] (else the right child of recs[j - 1]) */,
                                                           int store_qbox /* 0: qbox32[] is not written (round 6: 32 of this kernel's 106 bytes a leaf, and the kernel is bound by its stores from 4 M leaves up).
                                                                             Nothing on the default path reads it: the half traversal takes its query boxes out of the records, k_cross_fused its
                                                                             leaf-level pieces too (below).  Who does read it -- the from-the-root descent, the packers, k_exact on a mesh with a cell
                                                                             table, k_cross_records -- asks the host for it (mi355cd.hip: qbox_wanted / ensure_qbox) */)
{
    {
        const uint32_t G = gridDim.x * REFIT_BLK;
        for (uint32_t x = blockIdx.x * REFIT_BLK + threadIdx.x; x < zp.nq; x += G) zp.q[x] = make_uint4(0u, 0u, 0u, 0u);
        if (blockIdx.x == gridDim.x - 1) {
            for (uint32_t x = threadIdx.x; x < zp.nw0; x += REFIT_BLK) zp.w0[x] = 0u;
            for (uint32_t x = threadIdx.x; x < zp.nw1; x += REFIT_BLK) zp.w1[x] = 0u;
            if (threadIdx.x == 0 && zp.flag) *zp.flag = 0u;
        }
    }
    __shared__ float t[2 * REFIT_BLK][6];           // 24 KB
    // the sparse table of the deltas (10 KB) is dead once every node knows its range and split; the nodes' own boxes
    // (12 KB) then take its place
    constexpr size_t TABLE_BYTES = sizeof(DKey) * DL_LEVELS * DL_STRIDE, NB_BYTES = sizeof(float) * 6 * REFIT_BLK;
    __shared__ __align__(16) unsigned char scratch[TABLE_BYTES > NB_BYTES ? TABLE_BYTES : NB_BYTES];
    DKey (*dt)[DL_STRIDE] = reinterpret_cast<DKey (*)[DL_STRIDE]>(scratch);
    float (*nb)[6] = reinterpret_cast<float (*)[6]>(scratch);
    __shared__ int16_t lsplit[REFIT_BLK];
    __shared__ uint8_t lcov[REFIT_BLK];             // leaf b0 + t is a child of an IN-BLOCK node: that node's thread writes the leaf's box, in its record
    __shared__ unsigned long long lexact[REFIT_BLK / 64], lcertain[REFIT_BLK / 64];
    __shared__ unsigned long long acc[6];
    __shared__ int32_t lcross[64];
    __shared__ uint32_t lcount, lbase;
    const int b = blockIdx.x, b0 = b * REFIT_BLK, tid = threadIdx.x;
    const int j = b0 + tid;
    if (tid == 0) { lcount = 0; b32_store(t[0], b32_identity()); }
    lcov[tid] = 0;
    if (tid < 6) acc[tid] = (tid & 1) ? 0ull : ~0ull;                      // x1 x2 y1 y2 z1 z2: min, max, min, max, min, max
    // adjacent deltas of the positions b0-1 .. b0+512 (thread t: position b0-1+t; threads 0 and 1 also take the last two)
    for (int x = tid; x < DL_STRIDE; x += REFIT_BLK) {
        int v = 0;
        const int p = b0 - 1 + x;
        if (x < DL_N && p >= 0 && p < n - 1) v = delta_k(keys, n, p, keys[p], p + 1) + 1;      // 0 = the out-of-range -1 of bvh.cuh:48
        dt[0][x] = x < DL_N ? (DKey)(v << DK_SHIFT) : DK_PAD;
    }
    Box mine = box_identity();
    B32 m32 = b32_identity();
    bool exact = false, certain = false;
    if (j < n) {
        const LeafTri lt = leaf[j];
        const d3 A = load_vertex(verts, lt.v0), B = load_vertex(verts, lt.v1), C = load_vertex(verts, lt.v2);
        mine = box_set(A, B, C);                                           // box.cuh:13-22
        const Enc32 e = enc_leaf32(amb, mine, A, B, C, vamb, lt.v0, lt.v1, lt.v2);
        m32 = B32{e.lx, e.ly, e.lz, e.hx, e.hy, e.hz};
        certain = e.certain;
        exact = certain && box_is_fp32(mine);
        if (!exact || n == 1) store_box(boxes, (n - 1) + j, mine);        // an exact box is its fp32 copy (leaf_box64, cd_bvh.h); n == 1: the leaf is node 0
        if (store_qbox) {                                                  // (uniform)
            float4 *qp = reinterpret_cast<float4 *>(qbox32 + j);
            qp[0] = make_float4(m32.lx, m32.ly, m32.lz, m32.hx);
            qp[1] = make_float4(m32.hy, m32.hz, __uint_as_float((exact ? LB_EXACT : 0u) | (certain ? LB_CERTAIN : 0u) | (box_overlap(mine, mine) ? LB_SELF : 0u)), 0.f);
        }
    }
    b32_store(t[REFIT_BLK + tid], m32);
    {
        const unsigned long long em = __builtin_amdgcn_ballot_w64(exact), cm = __builtin_amdgcn_ballot_w64(certain);
        if ((tid & 63) == 0) { lexact[tid >> 6] = em; lcertain[tid >> 6] = cm; }
    }
    // the 9 levels above the leaves and the 9 upper levels of the sparse table, one of each per barrier;
    // global index of local node k at depth dd: ((nbp2 + b) << dd) + (k - 2^dd)
    for (int dd = REFIT_LOG - 1; dd >= 0; --dd) {
        __syncthreads();
        {
            const int k = REFIT_LOG - dd;
            for (int x = tid; x < DL_STRIDE; x += REFIT_BLK) {
                const int y = x + (1 << (k - 1));
                const DKey u = dt[k - 1][x], w = y < DL_STRIDE ? dt[k - 1][y] : DK_PAD;
                dt[k][x] = (w >> DK_SHIFT) < (u >> DK_SHIFT) ? (DKey)(w + (1 << (k - 1))) : u;     // a tie keeps the left one
            }
        }
        const int cnt = 1 << dd;
        if (tid < cnt) {
            const int k = cnt + tid;
            const B32 m = b32_merge(b32_load(t[2 * k]), b32_load(t[2 * k + 1]));
            b32_store(t[k], m);
            if (REFIT_LOG - dd >= seg_min) b32_store(seg32 + 6 * ((((size_t)nbp2 + b) << dd) + tid), m);
        }
    }
    __syncthreads();
    // FP64 box of the block's leaves: rounding is monotone, so the leaf with the smallest FP64 x1 is among the leaves whose
    // rounded x1 equals the block's rounded minimum -- only those (normally one) touch the accumulator
    if (j < n) {
        const B32 tot = b32_load(t[1]);
        if (m32.lx == tot.lx) atomicMin(&acc[0], f64_ordered(mine.x1));
        if (m32.hx == tot.hx) atomicMax(&acc[1], f64_ordered(mine.x2));
        if (m32.ly == tot.ly) atomicMin(&acc[2], f64_ordered(mine.y1));
        if (m32.hy == tot.hy) atomicMax(&acc[3], f64_ordered(mine.y2));
        if (m32.lz == tot.lz) atomicMin(&acc[4], f64_ordered(mine.z1));
        if (m32.hz == tot.hz) atomicMax(&acc[5], f64_ordered(mine.z2));
    }
    if (hint_cost) {                                                       // (uniform) the group's score for the order hint: asked for here, the answer is not needed by anything below
        uint32_t c = j < n ? (uint32_t)hint_tri[hint_perm[j]] : 0u;
#pragma unroll
        for (int o = 32; o; o >>= 1) { const uint32_t u = (uint32_t)__shfl_xor((int)c, o); c = u > c ? u : c; }
        if ((tid & 63) == 0 && j < n) hint_cost[j >> 6] = c;
    }
    // Which record holds leaf j's box: the parent rule of k_cross_fused for a range of one leaf -- leaf j is the left child of the record named j when
    // delta(j, j + 1) > delta(j - 1, j) (-1 past the ends), else the right child of the record named j - 1.  One bit a leaf, a ballot a wave: what the cross
    // nodes' range queries need to find a leaf's fp32 box in the RECORDS (this kernel writes it there for every leaf: below) now that qbox32[] is not stored.
    const bool is_left = j < n && (dt[0][tid + 1] >> DK_SHIFT) > (dt[0][tid] >> DK_SHIFT);
    {
        const unsigned long long sm = __builtin_amdgcn_ballot_w64(is_left);
        if ((tid & 63) == 0 && j < n) leaf_side[j >> 6] = sm;
    }
    const int i = j;                                                       // internal node with the same index
    int first = 0, last = 0, split = 0; bool have = false, cross = false;
    if (i < n - 1) {
        // determineRange, bvh.cuh:100-123, on the adjacent deltas: dl index of position p is p - (b0 - 1)
        const int x = tid;                                                 // dl index of position i - 1; position i is x + 1
        const int dL = (int)(dt[0][x] >> DK_SHIFT), dR = (int)(dt[0][x + 1] >> DK_SHIFT);
        const bool right = dR >= dL;                                       // d = sign(delta(i, i+1) - delta(i, i-1)); equal only when both are out of range
        const int thr = right ? dL : dR;                                   // delta_min + 1
        // right: the first position p > i with delta(p, p+1) < delta_min (leaf p is the last of the range);
        // left: the last position p < i - 1 with delta(p, p+1) < delta_min (leaf p + 1 is the first of the range)
        int p = nsv_dir(dt, right, right ? x + 2 : x - 1, thr);
        // (node 0 has delta_min = -1, nothing is below it: its range is everything)
        if (right && thr == 0) p = (n - 1 <= b0 + REFIT_BLK - 1) ? (n - 1) - (b0 - 1) : DL_N;
        if (right) { first = i; last = b0 - 1 + p; have = p <= REFIT_BLK; }   // dl index 512 is position b0+511, the block's last leaf
        else { first = b0 + p; last = i; have = p >= 0; }                   // dl index 0 is position b0-1: found there, the range starts at b0
        cross = !have;                                                      // ran off the block
        if (have) {
            // findSplit, bvh.cuh:57-98: the position of the (unique) minimum adjacent delta inside [first, last - 1]
            const int a = first - (b0 - 1), e = last - 1 - (b0 - 1);       // dl indices
            const int k = 31 - __clz(e - a + 1);
            const int a1 = e - (1 << k) + 1;
            const DKey m0 = dt[k][a], m1 = dt[k][a1];                      // (equal minima of the two windows are the same position: the minimum is unique)
            const int omask = (1 << DK_SHIFT) - 1;
            split = b0 - 1 + (((m1 >> DK_SHIFT) < (m0 >> DK_SHIFT)) ? a1 + ((int)m1 & omask) : a + ((int)m0 & omask));
            split_of[i] = split;
            if (split == first) lcov[split - b0] = 1;                       // its leaf children's boxes go out with its record
            if (split + 1 == last) lcov[split + 1 - b0] = 1;
        }
        lsplit[tid] = have ? (int16_t)(split - b0) : (int16_t)-1;
        if (cross) {
            const uint32_t k = atomicAdd(&lcount, 1u);
            if (k < 64u) lcross[k] = i;
            else { const uint32_t g = atomicAdd(cross_count, 1u); if (g < cross_cap) cross_list[g] = i; }
        }
    }
    __syncthreads();                                                        // every node has its range and split: the table is dead
    b32_store(nb[tid], seg_query32(t, have ? first - b0 : 1, have ? last - b0 : 0));   // the node's own box (an empty query for the others)
    __syncthreads();                                                        // nb[], lsplit[], acc[], lcount of the whole block
    // A leaf whose parent is a CROSS node (its range leaves the block: k_cross_fused writes that record) puts its box into its half of that record itself --
    // box, link ~j, and in the last word its EXACT (bit 0) / CERTAIN (bit 1) flags for the record's owner, who overwrites the half with the same box and the
    // final words.  After this kernel EVERY leaf's fp32 box is in the record half leaf_side[] names.
    if (j < n && n > 1 && !lcov[tid]) {
        float4 *h = const_cast<float4 *>(is_left ? rec_left(recs32, n, (uint32_t)j) : rec_right(recs32, n, (uint32_t)(j - 1)));
        h[0] = make_float4(m32.lx, m32.ly, m32.lz, m32.hx);
        h[1] = make_float4(m32.hy, m32.hz, __int_as_float(~j), __uint_as_float((exact ? 1u : 0u) | (certain ? 2u : 0u)));
    }
    if (have) {
        // children (bvh.cuh:174-195): the left child is leaf `split` or internal node `split`, the right one leaf / node split + 1
        const bool leafL = split == first, leafR = split + 1 == last;
        const int sl = split - b0, sr = split + 1 - b0;
        const B32 bl = b32_load(leafL ? t[REFIT_BLK + sl] : nb[sl]);
        const B32 br = b32_load(leafR ? t[REFIT_BLK + sr] : nb[sr]);
        const int32_t la = leafL ? ~split : b0 + (int)lsplit[sl], lb = leafR ? ~(split + 1) : b0 + (int)lsplit[sr];
        const uint32_t fl = ((leafL && ((lcertain[sl >> 6] >> (sl & 63)) & 1ull)) ? REC_L_CERTAIN : 0u) |
                            ((leafR && ((lcertain[sr >> 6] >> (sr & 63)) & 1ull)) ? REC_R_CERTAIN : 0u);
        const uint32_t fx = ((leafL && ((lexact[sl >> 6] >> (sl & 63)) & 1ull)) ? REC_L_EXACT : 0u) |
                            ((leafR && ((lexact[sr >> 6] >> (sr & 63)) & 1ull)) ? REC_R_EXACT : 0u);
        float4 *pl = const_cast<float4 *>(rec_left(recs32, n, (uint32_t)split)), *pr = const_cast<float4 *>(rec_right(recs32, n, (uint32_t)split));
        pl[0] = make_float4(bl.lx, bl.ly, bl.lz, bl.hx); pl[1] = make_float4(bl.hy, bl.hz, __int_as_float(la), __uint_as_float((uint32_t)first | fx));
        pr[0] = make_float4(br.lx, br.ly, br.lz, br.hx); pr[1] = make_float4(br.hy, br.hz, __int_as_float(lb), __uint_as_float((uint32_t)last | fl));
        if (i == 0) *root_name = split;                                    // (a tree of one block: the root is an in-block node)
    }
    if (tid == 0) {
        const Box tot{f64_from_ordered(acc[0]), f64_from_ordered(acc[1]), f64_from_ordered(acc[2]), f64_from_ordered(acc[3]),
                      f64_from_ordered(acc[4]), f64_from_ordered(acc[5])};
        store_box(seg, nbp2 + b, tot);
        if (nbp2 == 1) store_box(boxes, 0, tot);                           // one block: this IS the box of node 0 (k_cross_records never sees it)
    }
    // hand the block's cross nodes over: ONE global atomic per workgroup on the list's length (about 2 000 workgroups
    // finishing over the kernel's duration: ~30 returning atomics per microsecond on that word, a third of what it takes)
    const uint32_t cnt = min(lcount, 64u);
    if (cnt == 0) return;
    if (tid == 0) lbase = atomicAdd(cross_count, cnt);
    __syncthreads();
    if ((uint32_t)tid < cnt && lbase + tid < cross_cap) cross_list[lbase + tid] = lcross[tid];
}

// qbox32[] on request (round 6): the fused build no longer stores it (k_build_block, store_qbox); a reader that the build did not know of -- cd_pack_queries, external
// queries, the deep pass behind a half traversal, cd_debug_records -- gets it from this pass over the leaves: the same box, encoding and flags k_build_block forms.
__global__ __launch_bounds__(256) void k_fill_qbox(const double *__restrict__ verts, const LeafTri *__restrict__ leaf, int n, LeafBox32 *__restrict__ qbox32,
                                                   AmbTable amb, const uint8_t *__restrict__ vamb)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const LeafTri lt = leaf[j];
    const d3 A = load_vertex(verts, lt.v0), B = load_vertex(verts, lt.v1), C = load_vertex(verts, lt.v2);
    const Box mine = box_set(A, B, C);
    const Enc32 e = enc_leaf32(amb, mine, A, B, C, vamb, lt.v0, lt.v1, lt.v2);
    const bool exact = e.certain && box_is_fp32(mine);
    float4 *qp = reinterpret_cast<float4 *>(qbox32 + j);
    qp[0] = make_float4(e.lx, e.ly, e.lz, e.hx);
    qp[1] = make_float4(e.hy, e.hz, __uint_as_float((exact ? LB_EXACT : 0u) | (e.certain ? LB_CERTAIN : 0u) | (box_overlap(mine, mine) ? LB_SELF : 0u)), 0.f);
}

// Box of heap node k at level p (2^p leaves) for the cross nodes' queries: the leaves' fp32 boxes (levels below
// SEG32_MIN_LEVEL are rebuilt from them), the blocks' fp32 trees (seg32), above the blocks the FP64 boxes of
// k_refit_seg_top rounded outward.
__device__ __forceinline__ B32 seg_piece32(const AmbTable &amb, const double *__restrict__ seg, const float *__restrict__ seg32, const LeafBox32 *__restrict__ qbox32,
                                           int n, long long P, long long k, int p, int seg_min = SEG_MIN_LEVEL)
{
    if (p > REFIT_LOG) return b32_of(amb, load_box(seg, (int)k));
    if (p >= seg_min) return b32_load(seg32 + 6 * (size_t)k);
    const long long j0 = (k << p) - P;
    B32 x = b32_identity();
    for (int u = 0; u < (1 << p); ++u) { const long long jj = j0 + u; if (jj < n) x = b32_merge(x, b32_of_leaf(qbox32, (int)jj)); }
    return x;
}

template <int CTRL>
__device__ __forceinline__ float dpp_row_shl32(float v, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ B32 dpp_step32(const B32 &x)
{
    const float inf = __uint_as_float(0x7f800000u);
    const B32 y{dpp_row_shl32<CTRL>(x.lx, inf), dpp_row_shl32<CTRL>(x.ly, inf), dpp_row_shl32<CTRL>(x.lz, inf),
                dpp_row_shl32<CTRL>(x.hx, -inf), dpp_row_shl32<CTRL>(x.hy, -inf), dpp_row_shl32<CTRL>(x.hz, -inf)};
    return b32_merge(x, y);
}
__device__ __forceinline__ B32 b32_shfl_down(const B32 &x, int s)
{
    return B32{__shfl_down(x.lx, s), __shfl_down(x.ly, s), __shfl_down(x.lz, s), __shfl_down(x.hx, s), __shfl_down(x.hy, s), __shfl_down(x.hz, s)};
}

// The records of the cross nodes (range and split from k_cross_meta): one WAVE per node, its two child ranges side by
// side in the two halves of the wave -- lanes 0-31 answer [first, split], lanes 32-63 [split + 1, last]; inside a half,
// lane h < 16 owns the left piece of level h of the iterative bottom-up query and lane 31 - q the right piece of level q
// (16 levels cover every range shorter than 65536 leaves; longer ranges take the 32 + 32 levels of a whole wave, one
// child after the other).  Every lane fetches its piece (all loads in flight together), a segmented reduction folds them.
// Lane 0 stores the left half of the record, lane 32 the right half.  Queries read leaf boxes and segment-tree nodes,
// never another cross node's output: no ordering between the waves.
__global__ __launch_bounds__(256) void k_cross_records(int n, const NodeMeta *__restrict__ meta, const double *__restrict__ seg, const float *__restrict__ seg32,
                                                       int nbp2, const LeafBox32 *__restrict__ qbox32, double *__restrict__ boxes, NodeRec32 *__restrict__ recs32,
                                                       const int32_t *__restrict__ split_of, int32_t *__restrict__ root_name,
                                                       const int32_t *__restrict__ dense, const uint32_t *__restrict__ dense_total, uint32_t dense_cap, AmbTable amb)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t total = min(*dense_total, dense_cap);
    const long long P = (long long)nbp2 * REFIT_BLK;
    for (uint32_t kk = blockIdx.x * 4 + (tid >> 6); kk < total; kk += gridDim.x * 4) {   // one wave per node (kk is wave-uniform)
        const int i = __builtin_amdgcn_readfirstlane(dense[kk]);
        const NodeMeta m = meta[i];
        const int first = min(i, m.z), last = max(i, m.z);
        const int split = (m.x >= n - 1) ? m.x - (n - 1) : m.x;
        const bool second = lane >= 32;
        B32 x = b32_identity();
        if (last - first < 65535) {                                         // (wave-uniform) nearly all of them
            const int hl = lane & 31;
            const bool is_left = hl < 16;
            const int l0 = second ? split + 1 : first, r0 = second ? last : split;
            const int p = is_left ? hl : 31 - hl;                           // level of this lane's piece
            const long long lp = ((long long)l0 + P + ((1ll << p) - 1)) >> p;   // l at level p (ceil)
            const long long rp = ((long long)r0 + P + 1) >> p;              // r at level p (floor), half-open
            if (lp < rp) {
                const long long k = is_left ? lp : rp - 1;
                const bool take = is_left ? (lp & 1) : (rp & 1);
                if (take) x = seg_piece32(amb, seg, seg32, qbox32, n, P, k, p);
            }
            // steps 1, 2, 4, 8 stay inside a row of 16 lanes (DPP row_shl, a VALU move); a lane whose source would be
            // outside its row keeps the identity -- only lanes whose result is never consumed are affected
            x = dpp_step32<0x101>(x); x = dpp_step32<0x102>(x); x = dpp_step32<0x104>(x); x = dpp_step32<0x108>(x);
            x = b32_merge(x, b32_shfl_down(x, 16));                         // consumed in lanes 0 and 32, whose source is their own half
        } else {
            B32 res[2];
            for (int h = 0; h < 2; ++h) {
                const int l0 = h ? split + 1 : first, r0 = h ? last : split;
                const bool is_left = lane < 32;
                const int p = is_left ? lane : 63 - lane;
                const long long lp = ((long long)l0 + P + ((1ll << p) - 1)) >> p;
                const long long rp = ((long long)r0 + P + 1) >> p;
                B32 y = b32_identity();
                if (lp < rp) {
                    const long long k = is_left ? lp : rp - 1;
                    const bool take = is_left ? (lp & 1) : (rp & 1);
                    if (take) y = seg_piece32(amb, seg, seg32, qbox32, n, P, k, p);
                }
                for (int s = 1; s < 64; s <<= 1) y = b32_merge(y, b32_shfl_down(y, s));
                res[h] = B32{__shfl(y.lx, 0), __shfl(y.ly, 0), __shfl(y.lz, 0), __shfl(y.hx, 0), __shfl(y.hy, 0), __shfl(y.hz, 0)};
            }
            x = second ? res[1] : res[0];
        }
        const bool leafL = split == first, leafR = split + 1 == last;       // (wave-uniform)
        uint32_t fl = 0, fx = 0;
        if (leafL) { const uint32_t f = qbox32[first].flags; if (f & LB_CERTAIN) fl |= REC_L_CERTAIN; if (f & LB_EXACT) fx |= REC_L_EXACT; }
        if (leafR) { const uint32_t f = qbox32[last].flags; if (f & LB_CERTAIN) fl |= REC_R_CERTAIN; if (f & LB_EXACT) fx |= REC_R_EXACT; }
        if (lane == 0) {
            const int32_t la = child_link(meta, split_of, m.x, n - 1);
            float4 *pl = const_cast<float4 *>(rec_left(recs32, n, (uint32_t)split));
            pl[0] = make_float4(x.lx, x.ly, x.lz, x.hx); pl[1] = make_float4(x.hy, x.hz, __int_as_float(la), __uint_as_float((uint32_t)first | fx));
            if (i == 0) { *root_name = split; store_box(boxes, 0, load_box(seg, 1)); }   // the box of all leaves, from k_refit_seg_top
        }
        if (lane == 32) {
            const int32_t lb = child_link(meta, split_of, m.y, n - 1);
            float4 *pr = const_cast<float4 *>(rec_right(recs32, n, (uint32_t)split));
            pr[0] = make_float4(x.lx, x.ly, x.lz, x.hx); pr[1] = make_float4(x.hy, x.hz, __int_as_float(lb), __uint_as_float((uint32_t)last | fl));
        }
    }
}

// ====================================================================================================
// k_cross_meta + k_cross_records in ONE launch (round 5: for trees of any size; the upper levels of a tree of more than TOP_IN_BLOCK blocks are
// published by k_top_publish / k_top_publish_upper in front of it).  Two things tied those kernels to
// separate launches, and both can be had without a kernel boundary:
//   * the levels above the blocks, which the range queries read and which block 0 of k_cross_meta builds: every
//     workgroup folds the (at most 2048) fp32 block boxes into its own copy of those levels in LDS -- 24 bytes per
//     block, a microsecond, no waiting for anyone (rounding outward is monotone: the fold of the rounded block boxes
//     is the rounded fold);
//   * the link to a cross child, which is the CHILD's split (records are named by split, cd_bvh.h) and so was read
//     from split_of[] after k_cross_meta had finished: here the child WRITES its name into its parent's record.  The
//     parent needs no search -- a node covering [first, last] is the left child of the record named `last` when
//     delta(last, last + 1) > delta(first - 1, first) and the right child of the record named first - 1 otherwise (the
//     range grows towards the neighbour it shares more key bits with) -- and a cross node's parent is a cross node.
//     The record's owner writes every other word of the two halves; in-block children and leaves are linked by the
//     owner as before (split_of[] of k_build_block, ~leaf).
// A group of 16 lanes finds range and split of a node as in k_cross_meta (four nodes per wave); the wave then answers
// the two range queries of each of its four nodes as k_cross_records does, all their loads in flight together.
// Workgroup 0 also folds the FP64 block boxes (in order) into the box of all leaves, boxes[0].
// ====================================================================================================
// The levels from THREE above the blocks upwards as a private copy in LDS (top[6 * k]: heap node k in [1, nbp2 / 4)), folded
// from the blocks' fp32 boxes: a thread folds the eight blocks under one node of the lowest stored level, the lanes of a
// wave fold with shuffles, the four wave results meet in LDS -- two barriers instead of one per level.  (The two levels
// directly above the blocks are not stored: a piece of them is two / four block boxes away, and LDS is what limits how many
// workgroups, i.e. how many searches, a CU holds -- with 24 KB a workgroup (levels from two above the blocks) a CU held six, the
// registers allow seven, and the kernel's ~1 600 workgroups are ONE round only at seven.)
__device__ __forceinline__ B32 block_box32(const float *__restrict__ seg32, int nbp2, int nblocks, int b)
{
    const B32 v = b32_load(seg32 + 6 * ((size_t)nbp2 + (b < nblocks ? b : 0)));     // (an unconditional load: several of these go out together)
    return b < nblocks ? v : b32_identity();
}
// (blk: the fp32 boxes of the nbp2 blocks -- of the whole tree, seg32 + 6 nbp2, or of one span of TOP_IN_BLOCK blocks of a larger tree, k_top_publish --
//  of which the first nblocks exist.  low(k, box): round 5 -- the two levels directly above the blocks, heap nodes [nbp2 / 4, nbp2), are handed to the caller
//  as the thread that holds their eight blocks forms them: they are published too, so that a piece of them is ONE fetch for a cross node instead of two / four
//  block boxes -- which makes room among a group's 32 memory items for the levels 1 and 2, rebuilt from leaves since k_build_block no longer stores them.)
struct U3 { unsigned long long a, b, c; };
__device__ __forceinline__ U3 b32_words(const B32 &v)
{
    auto w = [](float lo, float hi) { return (unsigned long long)__float_as_uint(lo) | ((unsigned long long)__float_as_uint(hi) << 32); };
    return U3{w(v.lx, v.ly), w(v.lz, v.hx), w(v.hy, v.hz)};
}
template <class Low>
__device__ __forceinline__ void top32_to_lds(float *top, const float *__restrict__ blk, int nbp2, int nblocks, Low low)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int M = nbp2 >> 3;                                                // nodes of the lowest stored level, M .. 2 M - 1 (<= 256: nbp2 <= 2048)
    if (M >= 1) {                                                           // (workgroup-uniform)
        const int T = M;                                                    // threads that own a node
        B32 x = b32_identity();
        if (tid < T) {
            const int b = 8 * tid;
            // eight consecutive block boxes are 192 contiguous bytes, 16-byte aligned (nbp2 is a multiple of 8 here): twelve quads
            const float4 *q = reinterpret_cast<const float4 *>(blk + 6 * (size_t)b);
            float4 v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) v[u] = q[u];
            const B32 id = b32_identity();
            B32 pr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                                   // blocks b + 2u, b + 2u + 1 out of the quads 3u .. 3u + 2
                const float4 q0 = v[3 * u], q1 = v[3 * u + 1], q2 = v[3 * u + 2];
                const B32 b0 = b + 2 * u < nblocks ? B32{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y} : id, b1 = b + 2 * u + 1 < nblocks ? B32{q1.z, q1.w, q2.x, q2.y, q2.z, q2.w} : id;
                pr[u] = b32_merge(b0, b1);
                low((nbp2 >> 1) + 4 * tid + u, pr[u]);                      // a node of two blocks
            }
            const B32 q0 = b32_merge(pr[0], pr[1]), q1 = b32_merge(pr[2], pr[3]);
            low((nbp2 >> 2) + 2 * tid, q0); low((nbp2 >> 2) + 2 * tid + 1, q1);   // of four
            x = b32_merge(q0, q1);
            b32_store(top + 6 * (M + tid), x);
        }
        const int kt = M + tid;                                             // the node x is the box of
        // across lanes: after the step with stride s, lanes that are multiples of 2s hold node kt / (2s)
        for (int s = 1; s < 64 && s < T; s <<= 1) {
            x = b32_merge(x, b32_shfl_down(x, s));
            if (tid < T && (lane & (2 * s - 1)) == 0) b32_store(top + 6 * (kt / (2 * s)), x);
        }
        __syncthreads();
        if (T > 64 && tid == 0) {                                           // the T / 64 = 2 or 4 wave results are the heap nodes T / 64 .. 2 T / 64 - 1
            const int kw = T / 64;
            if (T == 128) b32_store(top + 6, b32_merge(b32_load(top + 6 * 2), b32_load(top + 6 * 3)));
            else {
                const B32 l = b32_merge(b32_load(top + 6 * kw), b32_load(top + 6 * (kw + 1))), r = b32_merge(b32_load(top + 6 * (kw + 2)), b32_load(top + 6 * (kw + 3)));
                b32_store(top + 6 * 2, l); b32_store(top + 6 * 3, r); b32_store(top + 6, b32_merge(l, r));
            }
        }
    }
    __syncthreads();
}
// A tree of 2 or 4 blocks has nothing but those two levels: one thread forms its one or three nodes.
template <class Low>
__device__ __forceinline__ void top32_tiny(const float *__restrict__ blk, int nbp2, int nblocks, Low low)
{
    if (threadIdx.x != 0) return;
    auto blkbox = [&](int b) { const B32 v = b32_load(blk + 6 * (size_t)(b < nblocks ? b : 0)); return b < nblocks ? v : b32_identity(); };
    const B32 l = b32_merge(blkbox(0), blkbox(1));
    if (nbp2 == 2) { low(1, l); return; }
    const B32 r = b32_merge(blkbox(2), blkbox(3));                         // nbp2 == 4
    low(2, l); low(3, r); low(1, b32_merge(l, r));
}

// The FP64 box of all leaves from the blocks' FP64 boxes, by one workgroup, ties in order (lower block = LEFT operand of
// box.cuh:24-32's compare-select: a fold of any shape that keeps the operand order gives the same bits).
__device__ __forceinline__ void root_box_fold(const double *__restrict__ seg, int nbp2, int nblocks, double *__restrict__ out)
{
    __shared__ double wbox[4][6];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int T = nbp2 < 256 ? nbp2 : 256, per = nbp2 / T;
    Box x = box_identity();
    if (tid < T)
        for (int u = 0; u < per; ++u) { const int b = tid * per + u; if (b < nblocks) x = box_merge(x, load_box(seg, nbp2 + b)); }
    for (int s = 1; s < 64 && s < T; s <<= 1) x = box_merge(x, box_shfl_down(x, s));
    if (lane == 0) { double *d = wbox[w]; d[0] = x.x1; d[1] = x.x2; d[2] = x.y1; d[3] = x.y2; d[4] = x.z1; d[5] = x.z2; }
    __syncthreads();
    if (tid == 0) {
        auto wb = [&](int k) { return Box{wbox[k][0], wbox[k][1], wbox[k][2], wbox[k][3], wbox[k][4], wbox[k][5]}; };
        Box r = wb(0);
        for (int k = 1; k < (T + 63) / 64; ++k) r = box_merge(r, wb(k));
        store_box(out, 0, r);
    }
}

// Trees of more than TOP_IN_BLOCK blocks (more than 1 M leaves): the levels k_cross_fused's first workgroup publishes for a small tree come from two
// launches of their own in front of it -- a kernel boundary instead of the flag protocol; ~10 us against the 0.7 ms such a step takes.
//   k_top_publish:       a workgroup per span of TOP_IN_BLOCK blocks folds the levels above the blocks up to the span's root (top32_to_lds)
//                        and stores them at their heap positions in top_pub: local node k at depth d is heap node ((nspans + span) << d) + (k - 2^d);
//   k_top_publish_upper: ONE workgroup folds the span roots (heap nodes [nspans, 2 nspans), nspans <= 1024: n <= 2^30) into the nodes [1, nspans)
//                        and sets the flag word to this launch's sequence number (what k_cross_fused's lanes look for).
__global__ __launch_bounds__(256) void k_top_publish(const float *__restrict__ seg32, int nbp2, int nblocks, unsigned long long *__restrict__ top_pub)
{
    __shared__ float top[6 * (TOP_IN_BLOCK / 4)];
    const int span = blockIdx.x, nspans = nbp2 / TOP_IN_BLOCK, tid = threadIdx.x;
    const int live = nblocks - span * TOP_IN_BLOCK;
    auto heap = [&](int k) { const int d = 31 - __clz(k); return ((size_t)(nspans + span) << d) + (size_t)(k - (1 << d)); };   // local node k of the span -> heap node of the tree
    top32_to_lds(top, seg32 + 6 * ((size_t)nbp2 + (size_t)span * TOP_IN_BLOCK), TOP_IN_BLOCK, live < 0 ? 0 : (live > TOP_IN_BLOCK ? TOP_IN_BLOCK : live),
                 [&](int k, const B32 &v) { const U3 w = b32_words(v); unsigned long long *d = top_pub + 3 * heap(k); d[0] = w.a; d[1] = w.b; d[2] = w.c; });
    for (int k = 1 + tid; k < TOP_IN_BLOCK / 4; k += 256) {
        const size_t g = heap(k);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(top + 6 * k);
#pragma unroll
        for (int u = 0; u < 3; ++u) top_pub[3 * g + u] = src[u];
    }
}
constexpr int TOP_MAX_SPANS = 1024;                                         // 2^30 leaves / 512 / TOP_IN_BLOCK
__global__ __launch_bounds__(256) void k_top_publish_upper(unsigned long long *__restrict__ top_pub, int nspans, uint32_t *__restrict__ top_flag, uint32_t top_seq)
{
    __shared__ float h[2 * TOP_MAX_SPANS][6];
    const int tid = threadIdx.x;
    for (int k = nspans + tid; k < 2 * nspans; k += 256) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(h[k]);
#pragma unroll
        for (int u = 0; u < 3; ++u) dst[u] = top_pub[3 * (size_t)k + u];
    }
    for (int w = nspans >> 1; w >= 1; w >>= 1) {                            // level of the nodes [w, 2 w)
        __syncthreads();
        for (int k = w + tid; k < 2 * w; k += 256) {
            const B32 m = b32_merge(b32_load(h[2 * k]), b32_load(h[2 * k + 1]));
            b32_store(h[k], m);
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(h[k]);
#pragma unroll
            for (int u = 0; u < 3; ++u) top_pub[3 * (size_t)k + u] = src[u];
        }
    }
    if (tid == 0) *top_flag = top_seq;
}

// (76 VGPRs: six waves per SIMD.  Held at 72 / 64 registers for seven / eight -- amdgpu_waves_per_eu, 20 / 48 bytes of scratch -- the kernel took 19.9 / 23.8 us
//  instead of 17.7 at 1 M triangles and 134 / 154 instead of 127 at 8 M: profiles/r05_experiments/size_scaling.log)
__global__ __launch_bounds__(256, 2) void k_cross_fused(const uint64_t *__restrict__ keys, int n, const double *__restrict__ seg, const float *__restrict__ seg32,
                                                     int nbp2, int nblocks, const unsigned long long *__restrict__ leaf_side /* k_build_block: which record half holds a leaf's box */, double *__restrict__ boxes,
                                                     NodeRec32 *__restrict__ recs32, const int32_t *__restrict__ split_of, int32_t *__restrict__ root_name,
                                                     const int32_t *__restrict__ dense, const uint32_t *__restrict__ dense_total, uint32_t dense_cap,
                                                     unsigned long long *top_pub /* [3 * nbp2 / 4] the upper levels, published by workgroup 0 */, uint32_t *top_flag, uint32_t top_seq,
                                                     uint32_t nord /* 0, or 8 x chunks: the FIRST workgroups of the grid sort the groups' scores (k_build_block) into the half traversal's order hint,
                                                                      one chunk of ORDER_MAX_ITEMS groups of one XCD list each (cd_bvh.h, build_half_order) -- beside this kernel's latency chain, in the dynamic LDS */,
                                                     uint32_t order_groups, const uint32_t *__restrict__ hint_cost, uint32_t *__restrict__ hint_order)
{
    extern __shared__ float top[];                                          // workgroup 0 only: [max(nbp2 / 4, 1)][6], heap nodes [1, nbp2 / 4)
    if (blockIdx.x < nord) { build_half_order<256>(blockIdx.x & 7u, blockIdx.x >> 3, order_groups, hint_cost, hint_order, *reinterpret_cast<OrderLds<256> *>(top)); return; }
    const uint32_t bid = blockIdx.x - nord, nbid = gridDim.x - nord;       // (the roles below count without them)
    const int tid = threadIdx.x, lane = tid & 63, g = lane / XG, gl = lane % XG;
    // (a workgroup of its own, the last one of the grid: the fold is a chain of dependent loads, short against what the
    //  others do, long when it comes on top of it)
    if (bid == nbid - 1) { root_box_fold(seg, nbp2, nblocks, boxes); return; }   // (workgroup-uniform)
    // The levels from three above the blocks upwards (a node of 8, 16, ... blocks), which only the few nodes with ranges of 4096 leaves or
    // more ask for: ONE workgroup -- the first of the grid, so it is resident before anybody can wait for it, and it waits for nobody --
    // folds them (top32_to_lds) and PUBLISHES them: agent-scope stores (the 8 L2s are not coherent inside a kernel), a wait for those
    // stores, then the launch's sequence number into the flag word.  A lane that needs such a piece (below) polls the flag -- by then the
    // searches have taken longer than the fold -- and reads the node with agent-scope loads.  Round 2 and the first half of round 3 had
    // EVERY workgroup fold its own copy into LDS in front of its searches: 4.2 of the kernel's 24 us (and 48 KB of L2 reads a workgroup).
    if (bid == 0) {
        if (nbp2 > TOP_IN_BLOCK) return;                                    // a large tree: k_top_publish + k_top_publish_upper have run in front of this launch
        auto low = [&](int k, const B32 &v) {                               // (the two levels directly above the blocks go out as their thread forms them)
            const U3 w = b32_words(v); unsigned long long *d = top_pub + 3 * (size_t)k;
            __hip_atomic_store(d, w.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(d + 1, w.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d + 2, w.c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if (nbp2 < 8) top32_tiny(seg32 + 6 * (size_t)nbp2, nbp2, nblocks, low);
        else top32_to_lds(top, seg32 + 6 * (size_t)nbp2, nbp2, nblocks, low);
        const int nn = nbp2 < 8 ? 1 : nbp2 >> 2;                            // heap nodes [1, nn) are in LDS
        for (int k = 1 + tid; k < nn; k += 256) {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(top + 6 * k);
#pragma unroll
            for (int u = 0; u < 3; ++u) __hip_atomic_store(top_pub + 3 * (size_t)k + u, src[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0);                                      // this wave's stores have been acknowledged ...
        __syncthreads();                                                    // ... every wave's
        if (tid == 0) __hip_atomic_store(top_flag, top_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    constexpr int PER_WAVE = 64 / XG, PER_BLOCK = 256 / XG;
    const uint32_t total = min(*dense_total, dense_cap);
    const long long P = (long long)nbp2 * REFIT_BLK;
    for (uint32_t wbase = (bid - 1) * PER_BLOCK + (tid >> 6) * PER_WAVE; wbase < total; wbase += (nbid - 2) * PER_BLOCK) {   // (wbase is wave-uniform; workgroups 1 .. nbid - 2 search)
        // ---- range and split, as k_cross_meta
        const uint32_t kq = wbase + g;
        const bool live = kq < total;
        int first = 0, last = 0, split = 0;
        {
            // (a dependent load costs a round trip even when it hits: values that are already in registers are not fetched again)
            auto delta_v = [](uint64_t ka, int ia, uint64_t kb, int ib) { const uint64_t x = ka ^ kb; return x ? __clzll((long long)x) : 64 + __clz((int)((uint32_t)ia ^ (uint32_t)ib)); };
            const int i = live ? dense[kq] : 0;
            // (the node's key and its two neighbours in ONE round trip: unconditional loads at clamped positions -- a load inside a
            //  branch is waited for inside the branch, and three branches are three round trips)
            const uint64_t ki = keys[i], kprev = keys[i > 0 ? i - 1 : 0], knext = keys[i + 1 < n ? i + 1 : n - 1];
            const int dnext = i + 1 < n ? delta_v(ki, i, knext, i + 1) : -1, dprev = i > 0 ? delta_v(ki, i, kprev, i - 1) : -1;
            const int d = (dnext - dprev) >= 0 ? 1 : -1;
            const int delta_min = d > 0 ? dprev : dnext;                    // delta(i, i - d)
            int mlen = 0;
            for (int r = 0; r < 2; ++r) {
                const bool todo = live && mlen == 0;
                if (!__builtin_amdgcn_ballot_w64(todo)) break;              // (wave-uniform)
                const int e = r * XG + gl;                                  // exponent - 1
                const long long o = (long long)i + (long long)d * (2ll << (e < 31 ? e : 31));
                const bool yes = todo && e < 31 && o >= 0 && o < n && delta_k(keys, n, i, ki, (int)o) > delta_min;
                const uint32_t no = ~group_ballot(yes, g) & ((1u << XG) - 1u);
                if (todo && no) mlen = 2 << (r * XG + __ffs((int)no) - 1);
            }
            const int l = group_last_true(live, mlen >> 1, mlen, g, gl, [&](int x) { return delta_k(keys, n, i, ki, i + x * d) > delta_min; });
            const int j = i + l * d;
            first = min(i, j); last = max(i, j);
            const uint64_t kj = keys[j];
            // (the keys next to the range's ends, for the parent rule below: requested together with the far end's key)
            const uint64_t kbefore = first > 0 ? keys[first - 1] : 0ull, kafter = last < n - 1 ? keys[last + 1] : 0ull;
            const uint64_t kf = first == i ? ki : kj, kl = last == i ? ki : kj;
            const int common = delta_v(kf, first, kl, last);
            split = group_last_true(live, first, last, g, gl, [&](int x) { return delta_k(keys, n, first, kf, x) > common; });
            // ---- the node's name goes into its parent's record (the root has none: its name is the tree's entry point)
            if (live && gl == 0) {
                if (first == 0 && last == n - 1) *root_name = split;
                else {
                    const int dl = first > 0 ? delta_v(kf, first, kbefore, first - 1) : -1, dr = last < n - 1 ? delta_v(kl, last, kafter, last + 1) : -1;   // -1 past the ends
                    uint32_t *half = reinterpret_cast<uint32_t *>(const_cast<float4 *>(dr > dl ? rec_left(recs32, n, (uint32_t)last) : rec_right(recs32, n, (uint32_t)(first - 1))));
                    half[6] = (uint32_t)split;                              // the link word of that half
                }
            }
        }
        // (what the record's owner has to fetch besides the boxes: requested before the pieces, used after them)
        const bool leafL = split == first, leafR = split + 1 == last;
        // a child that is itself a cross node (its range leaves its 512-leaf block: exactly k_build_block's test) links itself
        const bool crossL = !leafL && first / REFIT_BLK != split / REFIT_BLK, crossR = !leafR && (split + 1) / REFIT_BLK != last / REFIT_BLK;
        // (unconditional loads at valid positions, consumed after the pieces: a load whose value decides a branch is waited for on
        //  the spot -- written as `if (flags & EXACT) ...` these were up to two round trips in front of the pieces')
        const bool own = live && gl == 0;
        // (a leaf child's EXACT / CERTAIN flags: k_build_block left them in the last word of the half this node is about to write -- the leaf's parent is THIS cross node)
        const uint32_t pwL = reinterpret_cast<const uint32_t *>(rec_left(recs32, n, (uint32_t)(own && leafL ? split : 0)))[7];
        const uint32_t pwR = reinterpret_cast<const uint32_t *>(rec_right(recs32, n, (uint32_t)(own && leafR ? split : 0)))[7];
        const uint32_t flagL = ((pwL & 1u) ? LB_EXACT : 0u) | ((pwL & 2u) ? LB_CERTAIN : 0u), flagR = ((pwR & 1u) ? LB_EXACT : 0u) | ((pwR & 2u) ? LB_CERTAIN : 0u);
        const int32_t soL = split_of[own && !leafL && !crossL ? split : 0], soR = split_of[own && !leafR && !crossR ? split + 1 : 0];
        // ---- the records: each group answers the two range queries of its own node.  A range [l, r] is the union of at most one
        // left and one right piece per level of the iterative bottom-up query; what a piece is made of depends on its level:
        //   levels 0 .. 2: one / two / four leaf boxes (round 5: k_build_block is bound by its writes, and the levels 1 and 2 of its trees were 18 bytes a
        //   leaf; round 6: qbox32[], another 32, is not written either -- a leaf's box is in the RECORDS, the left half of recs[j] or the right half of
        //   recs[j - 1] by leaf_side[]'s bit: both halves and the bit word are fetched together, unconditionally, and selected afterwards);
        //   levels 3 .. 9: one stored node of a block's tree (seg32); levels >= 10: one published node (top_pub).
        // The memory items of a child range are 14 + 14 = 28, of the node 56: lane gl fetches items gl and gl + 16 of either child
        // -- four 24-byte loads per lane, all unconditional (an item that is not taken reads a valid dummy and is dropped), ONE round
        // trip for the whole group.  (Before: a lane per level and a branch per kind of level -- divergent branches run one after
        // the other, each with its own round trip: three of them on this kernel's critical path.)  The LDS levels keep the lane-per-
        // level form (16 levels a sweep, a second sweep for ranges of 65536 leaves or more); then the 16 lanes fold with DPP row shifts.
        const int sweeps = __builtin_amdgcn_ballot_w64(live && last - first >= 65535) ? 2 : 1;   // (wave-uniform)
        B32 accL = b32_identity(), accR = b32_identity();
        auto piece = [&](int h, int p, int side, long long &k) {              // is the side's piece of child h taken at level p, and which heap node is it
            const int l0 = h ? split + 1 : first, r0 = h ? last : split;
            const long long lp = ((long long)l0 + P + ((1ll << p) - 1)) >> p;   // l at level p (ceil)
            const long long rp = ((long long)r0 + P + 1) >> p;              // r at level p (floor), half-open
            k = side ? rp - 1 : lp;
            return live && lp < rp && ((side ? rp : lp) & 1);
        };
        for (int sw = 0; sw < sweeps; ++sw) {
            const int p = gl + 16 * sw;
            long long kk[4]; bool tk[4]; bool any_top = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) { tk[u] = p > REFIT_LOG && piece(u >> 1, p, u & 1, kk[u]); any_top |= tk[u]; }
            if (!__builtin_amdgcn_ballot_w64(any_top)) continue;            // (wave-uniform) no range among the wave's four nodes takes a piece of two blocks or more
            bool ready = !any_top;
            // (a piece of the two levels directly above the blocks is two / four block boxes away: a lane that finds the flag not yet set folds those itself
            //  instead of holding its wave for the publisher -- only a piece of eight blocks or more is worth the wait)
            const bool must = any_top && p > REFIT_LOG + 2;
#ifdef CROSS_FORCE_FALLBACK                                                // (test builds: nobody sees the flag, every upper-level piece is folded by the lane that needs it)
            for (uint32_t spin = 0; spin < 0u; ++spin) {
#else
            for (uint32_t spin = 0; spin < (1u << 20); ++spin) {            // (bounded: a lane that never sees the flag folds its node itself, below)
#endif
                if (!ready) ready = __hip_atomic_load(top_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == top_seq;
                if (!__builtin_amdgcn_ballot_w64(must && !ready)) break;    // every lane that has to wait has seen this launch's number
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (tk[u]) {
                B32 v;
                if (ready) {
                    unsigned long long w[3];
#pragma unroll
                    for (int e = 0; e < 3; ++e) w[e] = __hip_atomic_load(top_pub + 3 * (size_t)kk[u] + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = B32{__uint_as_float((uint32_t)w[0]), __uint_as_float((uint32_t)(w[0] >> 32)), __uint_as_float((uint32_t)w[1]),
                            __uint_as_float((uint32_t)(w[1] >> 32)), __uint_as_float((uint32_t)w[2]), __uint_as_float((uint32_t)(w[2] >> 32))};
                } else {                                                    // the publisher never showed up (cannot happen: it is workgroup 0 and waits for nobody): fold the node's blocks here
                    v = b32_identity();
                    const long long b_first = (kk[u] << (p - REFIT_LOG)) - nbp2;
                    for (long long bb = 0; bb < (1ll << (p - REFIT_LOG)); ++bb) v = b32_merge(v, block_box32(seg32, nbp2, nblocks, (int)(b_first + bb)));
                }
                if (u < 2) accL = b32_merge(accL, v); else accR = b32_merge(accR, v);
            }
        }
        {
            // items of a child range: the levels below SEG32_MIN_LEVEL leaf by leaf (level p: 2 sides x 2^p leaves, items 2^(p+1) - 2 ...), then one item
            // per side of the levels SEG32_MIN_LEVEL .. REFIT_LOG of a block's tree (seg32); levels above the blocks: the published nodes (above)
            constexpr int SMIN = SEG32_MIN_LEVEL, LEAF_ITEMS = 2 * ((1 << SMIN) - 1), N_ITEMS = LEAF_ITEMS + 2 * (REFIT_LOG + 1 - SMIN);
            static_assert(XG == 16 && N_ITEMS <= 2 * XG, "a lane fetches items gl and gl + 16 of either child");
            static_assert(REFIT_BLK == (1 << REFIT_LOG) && SMIN >= 1 && SMIN <= REFIT_LOG, "item t >= LEAF_ITEMS is side (t - LEAF_ITEMS) & 1 of level SMIN + (t - LEAF_ITEMS) / 2 <= REFIT_LOG of a 2^REFIT_LOG-leaf block");
            B32 it[4]; bool tk[4];
            static_assert(LEAF_ITEMS <= XG, "the leaf items are a lane's FIRST item of either child (r = 0, 2)");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = r >> 1, t = gl + 16 * (r & 1);                // item t of child h
                int p, side, e;
                if (t < LEAF_ITEMS) { const int lg = 31 - __clz(t + 2); p = lg - 1; const int idx = t + 2 - (1 << lg); side = idx >> p; e = idx & ((1 << p) - 1); }
                else { const int u = t - LEAF_ITEMS; p = SMIN + (u >> 1); side = u & 1; e = 0; if (p > REFIT_LOG) p = REFIT_LOG; }
                long long k;
                bool take = t < N_ITEMS && piece(h, p, side, k);
                // (every load unconditional, addresses by select: a load inside a divergent branch is waited for inside it -- a round trip per arm)
                const bool leafitem = (r & 1) == 0 && p < SMIN;             // ((r & 1): compile time -- a lane's second items are never leaf items)
                uint32_t jl = 0u;
                if (leafitem) {
                    const long long jj = take ? ((k << p) - P + e) : 0;     // leaf e of the piece
                    take = take && jj < n;                                  // (past the last leaf: nothing there)
                    jl = take ? (uint32_t)jj : 0u;
                }
                // a leaf's box: the first 24 bytes of the left half of recs[jl] or of the right half of recs[jl - 1] (clamped: the half that is not the leaf's is fetched and dropped)
                const float *p1 = leafitem ? reinterpret_cast<const float *>(rec_left(recs32, n, jl + 1u < (uint32_t)n ? jl : 0u)) : seg32 + 6 * (size_t)(take ? k : (P >> p));
                const B32 v1 = b32_load(p1);
                if ((r & 1) == 0) {
                    const B32 vr = b32_load(leafitem ? reinterpret_cast<const float *>(rec_right(recs32, n, jl > 0u ? jl - 1u : 0u)) : p1);
                    const unsigned long long sw = leaf_side[jl >> 6];
                    const bool rt = leafitem && ((sw >> (jl & 63u)) & 1ull) == 0ull;
                    it[r] = B32{rt ? vr.lx : v1.lx, rt ? vr.ly : v1.ly, rt ? vr.lz : v1.lz, rt ? vr.hx : v1.hx, rt ? vr.hy : v1.hy, rt ? vr.hz : v1.hz};
                } else it[r] = v1;
                tk[r] = take;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) if (tk[r]) { if (r < 2) accL = b32_merge(accL, it[r]); else accR = b32_merge(accR, it[r]); }
        }
        accL = dpp_step32<0x101>(accL); accL = dpp_step32<0x102>(accL); accL = dpp_step32<0x104>(accL); accL = dpp_step32<0x108>(accL);
        accR = dpp_step32<0x101>(accR); accR = dpp_step32<0x102>(accR); accR = dpp_step32<0x104>(accR); accR = dpp_step32<0x108>(accR);
        if (own) {
            const uint32_t fl = ((leafL && (flagL & LB_CERTAIN)) ? REC_L_CERTAIN : 0u) | ((leafR && (flagR & LB_CERTAIN)) ? REC_R_CERTAIN : 0u);
            const uint32_t fx = ((leafL && (flagL & LB_EXACT)) ? REC_L_EXACT : 0u) | ((leafR && (flagR & LB_EXACT)) ? REC_R_EXACT : 0u);
            const int32_t linkL = leafL ? ~split : soL, linkR = leafR ? ~(split + 1) : soR;      // (a cross child writes its own name: the link word is not touched below)
            float4 *pl = const_cast<float4 *>(rec_left(recs32, n, (uint32_t)split)), *pr = const_cast<float4 *>(rec_right(recs32, n, (uint32_t)split));
            pl[0] = make_float4(accL.lx, accL.ly, accL.lz, accL.hx);
            pr[0] = make_float4(accR.lx, accR.ly, accR.lz, accR.hx);
            if (crossL) { reinterpret_cast<float2 *>(pl + 1)[0] = make_float2(accL.hy, accL.hz); reinterpret_cast<uint32_t *>(pl + 1)[3] = (uint32_t)first | fx; }
            else pl[1] = make_float4(accL.hy, accL.hz, __int_as_float(linkL), __uint_as_float((uint32_t)first | fx));
            if (crossR) { reinterpret_cast<float2 *>(pr + 1)[0] = make_float2(accR.hy, accR.hz); reinterpret_cast<uint32_t *>(pr + 1)[3] = (uint32_t)last | fl; }
            else pr[1] = make_float4(accR.hy, accR.hz, __int_as_float(linkR), __uint_as_float((uint32_t)last | fl));
        }
    }
}

}  // namespace cd
