// cd_bvh.h -- LBVH build kernels: centroid + Morton keys, leaf fill, Karras-2012 hierarchy,
// bottom-up AABB refit, structural verifier counters.  Index-based SoA tree:
//   internal node i -> id i (root 0), leaf j -> id (n-1)+j, -1 = NULL.
#pragma once
#include "cd_math.h"

namespace cd {

// Topology of an internal node in one 16-byte record, written with one store by the node's own thread:
//   x = left child, y = right child (unified ids), z = the other end of the node's leaf range (Karras: node i
//   covers [min(i,z), max(i,z)]), w = 0.  Parent links live in parent[] (2n-1 entries, unified ids) because they
//   are written by the PARENT's thread.  Replaces the reference's pointer-linked 112-byte Node (bvh.cuh:25-43).
typedef int4 NodeMeta;

// fp32 traversal record, one 64-byte line: both child boxes rounded OUTWARD to float (lo down, hi up) +
// both child ids.  Internal-node boxes only cull; a conservative (superset) box can never lose a pair, and
// every leaf hit is re-decided with the exact FP64 product-form test of box.cuh:40-43 before it counts.
struct alignas(64) NodeRec32 {
    float l_lo[3], l_hi[3];
    float r_lo[3], r_hi[3];
    int32_t cl, cr;
    int32_t pad[2];
};
static_assert(sizeof(NodeRec32) == 64, "NodeRec32 must be one 64-byte line");

// Sorted-order leaf payload: {ID, vIdx[0..2]} (triangle.cuh:6,9) -- 16 B instead of the 56-byte Triangle.
struct alignas(16) LeafTri { uint32_t id, v0, v1, v2; };

// ---------------------------------------------------------------- centroid AABB (CD_FRAME_AUTO)
// Stage 1: per-workgroup min/max of centroids; stage 2 (one workgroup) folds the partials and
// writes frame = {off[3], span[3]}.  span is widened by 2^-20 relative so that the max maps below 2^20.
__device__ __forceinline__ d3 centroid_of(const double *__restrict__ verts, const uint32_t *__restrict__ vidx, uint32_t t)
{
    const uint32_t a = vidx[3 * (size_t)t], b = vidx[3 * (size_t)t + 1], c = vidx[3 * (size_t)t + 2];
    const d3 p1 = load_vertex(verts, a), p2 = load_vertex(verts, b), p3 = load_vertex(verts, c);
    // load_obj.h:90: (p1 + p2 + p3) / 3 per axis
    return d3{(p1.x + p2.x + p3.x) / 3, (p1.y + p2.y + p3.y) / 3, (p1.z + p2.z + p3.z) / 3};
}

__device__ __forceinline__ double wave_min(double v) { for (int o = 32; o; o >>= 1) { double t = __shfl_xor(v, o); v = t < v ? t : v; } return v; }
__device__ __forceinline__ double wave_max(double v) { for (int o = 32; o; o >>= 1) { double t = __shfl_xor(v, o); v = t > v ? t : v; } return v; }

__global__ __launch_bounds__(256) void k_centroid_bounds(const double *__restrict__ verts, const uint32_t *__restrict__ vidx, uint32_t n,
                                                         double *__restrict__ partial /* gridDim.x x 6 */)
{
    __shared__ double sm[4][6];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const d3 c = centroid_of(verts, vidx, t);
        lo[0] = c.x < lo[0] ? c.x : lo[0]; hi[0] = c.x > hi[0] ? c.x : hi[0];
        lo[1] = c.y < lo[1] ? c.y : lo[1]; hi[1] = c.y > hi[1] ? c.y : hi[1];
        lo[2] = c.z < lo[2] ? c.z : lo[2]; hi[2] = c.z > hi[2] ? c.z : hi[2];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int a = 0; a < 3; ++a) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); }
    if (lane == 0) for (int a = 0; a < 3; ++a) { sm[w][a] = lo[a]; sm[w][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = sm[0][threadIdx.x];
        for (int ww = 1; ww < 4; ++ww) { const double t = sm[ww][threadIdx.x]; v = (threadIdx.x < 3) ? (t < v ? t : v) : (t > v ? t : v); }
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
__global__ void k_frame_from_bounds(const double *__restrict__ partial, uint32_t nblocks, double *__restrict__ frame)
{
    if (threadIdx.x < 3) {
        double lo = 1e300, hi = -1e300;
        for (uint32_t b = 0; b < nblocks; ++b) {
            const double l = partial[b * 6 + threadIdx.x], h = partial[b * 6 + 3 + threadIdx.x];
            lo = l < lo ? l : lo; hi = h > hi ? h : hi;
        }
        double span = (hi - lo) * (1.0 + 1.0 / 1048576.0);
        if (!(span > 0.0)) span = 1.0;
        frame[threadIdx.x] = lo;
        frame[3 + threadIdx.x] = span;
    }
}

// ---------------------------------------------------------------- Morton keys (load_obj.h:89-101, morton.h:70-89)
__global__ __launch_bounds__(256) void k_morton(const double *__restrict__ verts, const uint32_t *__restrict__ vidx, uint32_t n,
                                                const double *__restrict__ frame /* off[3], span[3] */,
                                                uint64_t *__restrict__ keys)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const d3 c = centroid_of(verts, vidx, t);
    keys[t] = morton3d(c.x, c.y, c.z, frame, frame + 3);
}

// ---------------------------------------------------------------- fillLeafNodes (bvh.cuh:125-144)
// leaf j <- triangle perm[j]; also resets the parent links and the refit arrival counters that the
// reference gets from zero-initialised cudaMalloc memory (main.cu:84-85).
__global__ __launch_bounds__(256) void k_fill_leaves(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ vidx,
                                                     const uint32_t *__restrict__ ids, uint32_t n,
                                                     LeafTri *__restrict__ leaf, int32_t *__restrict__ parent, uint32_t *__restrict__ bounded)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        const uint32_t t = perm[j];
        LeafTri lt;
        lt.id = ids ? ids[t] : t;
        lt.v0 = vidx[3 * (size_t)t]; lt.v1 = vidx[3 * (size_t)t + 1]; lt.v2 = vidx[3 * (size_t)t + 2];
        leaf[j] = lt;
        parent[(n - 1) + j] = -1;
        if (j < n - 1) { parent[j] = -1; bounded[j] = 0; }
    }
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

constexpr int REFIT_BLK = 512;     // leaves per workgroup

// Merge step shared by both refit phases: given my box, my sibling's box and which side I am, write the
// parent's 128-byte traversal record {bl, br, cl, cr} and return the parent's box (bvh.cuh:277).
__device__ __forceinline__ void store_rec32(NodeRec32 *__restrict__ r, const Box &bl, const Box &br, int2 ch)
{
    float4 *p = reinterpret_cast<float4 *>(r);
    p[0] = make_float4(__double2float_rd(bl.x1), __double2float_rd(bl.y1), __double2float_rd(bl.z1), __double2float_ru(bl.x2));
    p[1] = make_float4(__double2float_ru(bl.y2), __double2float_ru(bl.z2), __double2float_rd(br.x1), __double2float_rd(br.y1));
    p[2] = make_float4(__double2float_rd(br.z1), __double2float_ru(br.x2), __double2float_ru(br.y2), __double2float_ru(br.z2));
    reinterpret_cast<int4 *>(r)[3] = make_int4(ch.x, ch.y, 0, 0);      // ch already encoded: internal i >= 0, leaf j -> ~j
}

// Merge step shared by both refit phases: given my box, my sibling's box and which side I am, write the
// parent's 64-byte fp32 traversal record and return the parent's exact box (bvh.cuh:277 merge(childA, childB)).
__device__ __forceinline__ Box refit_merge(const Box &mine, const Box &other, bool left, int cl, int cr, NodeRec32 *__restrict__ rec32, int nleaf_base)
{
    Box bl, br;
    bl.x1 = left ? mine.x1 : other.x1; bl.x2 = left ? mine.x2 : other.x2;
    bl.y1 = left ? mine.y1 : other.y1; bl.y2 = left ? mine.y2 : other.y2;
    bl.z1 = left ? mine.z1 : other.z1; bl.z2 = left ? mine.z2 : other.z2;
    br.x1 = left ? other.x1 : mine.x1; br.x2 = left ? other.x2 : mine.x2;
    br.y1 = left ? other.y1 : mine.y1; br.y2 = left ? other.y2 : mine.y2;
    br.z1 = left ? other.z1 : mine.z1; br.z2 = left ? other.z2 : mine.z2;
    // child ids in the traversal record: internal node i >= 0, leaf j -> ~j (sign bit = leaf flag)
    store_rec32(rec32, bl, br, make_int2(cl >= nleaf_base ? ~(cl - nleaf_base) : cl, cr >= nleaf_base ? ~(cr - nleaf_base) : cr));
    return box_merge(bl, br);
}

constexpr int REFIT_SB = 64 * REFIT_BLK;     // leaves per super-block (phase 2 workgroup): 32 768

// Phase 1 -- block-local subtrees.  Workgroup b owns leaves [b*BLK, (b+1)*BLK).  An internal node whose
// leaf range lies inside that interval has both children finished by threads of this workgroup, so its
// arrival counter and the sibling-box hand-off live in LDS (workgroup-scope acq_rel: no cache maintenance).
// A thread that reaches a parent spanning workgroups stops and appends its node to the list of its
// super-block; the next phase continues from there after the kernel boundary has made every box visible.
__global__ __launch_bounds__(REFIT_BLK) void k_refit_local(const double *__restrict__ verts, const LeafTri *__restrict__ leaf, int n,
                                                           const NodeMeta *__restrict__ meta, const int32_t *__restrict__ parent,
                                                           double *__restrict__ boxes, uint32_t *__restrict__ bounded,
                                                           NodeRec32 *__restrict__ recs32,
                                                           int32_t *__restrict__ sb_list /* [n], region sb*REFIT_SB */, uint32_t *__restrict__ sb_count)
{
    __shared__ double lbox[REFIT_BLK][2][6];       // deposit slots: [local node][side] = child box, 48 KB
    __shared__ uint32_t lcnt[REFIT_BLK];
    const int b0 = blockIdx.x * REFIT_BLK;
    lcnt[threadIdx.x] = 0;
    __syncthreads();
    const int j = b0 + threadIdx.x;
    if (j >= n) return;
    const LeafTri lt = leaf[j];
    Box mine = box_set(load_vertex(verts, lt.v0), load_vertex(verts, lt.v1), load_vertex(verts, lt.v2));
    int me = (n - 1) + j;
    store_box(boxes, me, mine);
    int cur = parent[me];
    while (cur != -1) {
        const NodeMeta m = meta[cur];
        const int up = parent[cur];                                        // independent of m: both loads in flight together
        const int first = min(cur, m.z), last = max(cur, m.z);
        if (!(first >= b0 && last < b0 + REFIT_BLK)) {                     // parent spans workgroups: hand over to the next phase
            const int sb = b0 / REFIT_SB;
            sb_list[(size_t)sb * REFIT_SB + atomicAdd(&sb_count[sb], 1u)] = me;
            break;
        }
        const bool left = (m.x == me);
        const int slot = cur - b0;
        double *dst = lbox[slot][left ? 0 : 1];
        dst[0] = mine.x1; dst[1] = mine.x2; dst[2] = mine.y1; dst[3] = mine.y2; dst[4] = mine.z1; dst[5] = mine.z2;
        // Deposit and arrival are both LDS operations of this wave, and the LDS executes one wave's operations
        // in issue order: whoever observes the incremented counter also observes the deposit.  So the atomic can
        // be relaxed -- an acq_rel one would also drain this wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0))
        // at every level.  The wavefront-scope fences emit no instruction; they pin the compiler's order.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const uint32_t old = __hip_atomic_fetch_add(&lcnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (old == 0) break;                                               // first arriver leaves (bvh.cuh:270-272)
        const double *src = lbox[slot][left ? 1 : 0];
        const Box other{src[0], src[1], src[2], src[3], src[4], src[5]};
        bounded[cur] = 2;                                                  // Node::bounded: both arrivals seen
        mine = refit_merge(mine, other, left, m.x, m.y, recs32 + cur, n - 1);
        me = cur;
        store_box(boxes, me, mine);
        cur = up;
    }
}

// Phases 2 and 3 -- nodes that span phase-1 workgroups.  One workgroup per super-block climbs the nodes whose
// range stays inside [sb*span, (sb+1)*span): every arrival at such a node comes from THIS workgroup, so the
// arrival counters (global memory) and the box hand-offs need only workgroup-scope ordering -- same CU, same
// L1, no L2 write-back / invalidate per step.  A thread that reaches a parent leaving the super-block appends
// its node to out_list for the next phase.  Phase 3 is the same kernel with ONE workgroup and span = everything.
// (The reference's atomicAdd at bvh.cuh:270 has no ordering at all -- a race on real hardware.)
__global__ __launch_bounds__(1024) void k_refit_mid(int n, const NodeMeta *__restrict__ meta, const int32_t *__restrict__ parent,
                                                    double *boxes, uint32_t *bounded, NodeRec32 *__restrict__ recs32,
                                                    const int32_t *__restrict__ in_list, const uint32_t *__restrict__ in_count,
                                                    size_t in_stride, long long span,
                                                    int32_t *__restrict__ out_list, uint32_t *__restrict__ out_count)
{
    const uint32_t count = in_count[blockIdx.x];
    const long long lo = (long long)blockIdx.x * span, hi = lo + span;
    const int32_t *list = in_list + (size_t)blockIdx.x * in_stride;
    for (uint32_t k = threadIdx.x; k < count; k += blockDim.x) {
        int me = list[k];
        Box mine = load_box(boxes, me);
        int cur = parent[me];
        while (cur != -1) {
            const NodeMeta m = meta[cur];
            const int up = parent[cur];
            const int first = min(cur, m.z), last = max(cur, m.z);
            if (!(first >= lo && last < hi)) { out_list[atomicAdd(out_count, 1u)] = me; break; }
            const uint32_t old = __hip_atomic_fetch_add(&bounded[cur], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == 0) break;
            const bool left = (m.x == me);
            const Box other = load_box(boxes, left ? m.y : m.x);
            mine = refit_merge(mine, other, left, m.x, m.y, recs32 + cur, n - 1);
            me = cur;
            store_box(boxes, me, mine);
            cur = up;
        }
    }
}

// ---------------------------------------------------------------- verifier counters (check.cuh)
constexpr uint64_t BOX_UNINIT_BITS = 0xFFFFFFFFFFFFFFFFull;   // boxes are memset to 0xFF before refit

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
