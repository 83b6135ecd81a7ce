// cd_traverse.h -- BVH overlap traversal + exact triangle test (collision.cuh:19-88).
#pragma once
#include "cd_bvh.h"

namespace cd {

// ---------------------------------------------------------------- device-side counters
// A returning atomic on ONE address retires at only ~88 per microsecond on this chip (MI355X_MICROARCH.md,
// "dequeue" row), so nothing on the hot path funnels through a single word: statistics and candidate
// reservations are spread over NSHARD counters, each on its own 128-byte line (shard = workgroup & 63), and
// pairs are staged per workgroup in LDS and appended with one atomic per workgroup.
constexpr int NSHARD = 64;
struct alignas(128) CtrShard {
    unsigned long long pairs_tested;   // leaf AABB hits (reach neighborCount / SAT)
    unsigned long long node_visits;
    unsigned long long n_candidates;   // entries reserved in this shard of the candidate buffer (may exceed its capacity): low 32 bits the 8-byte candidates, from the shard's
                                       // front; high 32 bits the 32-byte SAT-ready pairs (FatPair below), from its end -- one atomic reserves both
    unsigned long long wave_steps;     // descent-loop iterations summed over waves (lane utilisation = node_visits / (64 * wave_steps))
    unsigned long long pad[12];        // [4] / [11]: the descent's own clock (earliest start as its complement, latest end); the rest: diagnostics (DIAG instances of the kernels)
};
// slots of 8 bytes the reservations of one shard's counter word take
__host__ __device__ __forceinline__ unsigned long long cand_slots(unsigned long long w) { return (w & 0xffffffffull) + 4ull * (w >> 32); }
__host__ __device__ __forceinline__ unsigned long long cand_entries(unsigned long long w) { return (w & 0xffffffffull) + (w >> 32); }
struct alignas(128) TravState {
    unsigned long long n_pairs;        // collision.cuh:40 `count`
    uint32_t n_deferred;               // (query, subtree) items the LDS stack could not hold (deep pass redoes them)
    uint32_t report_arrive;            // workgroups of k_report that have posted their part (polled completion, see k_report)
    unsigned long long pad[14];
    CtrShard shard[NSHARD];
};
static_assert(sizeof(CtrShard) == 128 && sizeof(TravState) == 128 * (NSHARD + 1), "counter layout");

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// What the host needs after a traversal, gathered into ONE 256-byte record that sits directly in front of the pair
// list, so that counters + sort flags + root box + the first pairs come back in a single device-to-host copy.
struct alignas(256) Report {
    unsigned long long n_pairs, pairs_tested, node_visits, max_shard_candidates, wave_steps, candidates;
    uint32_t n_deferred, pad0;
    uint32_t sort_flags[9]; uint32_t pad1;
    double root_box[6];
    unsigned long long clk_start_inv, clk_end;   // descent kernel, device wall clock (s_memrealtime ticks): ~(earliest wave start), latest wave end; 0 = not taken
    unsigned long long seq;                      // polled completion: the step's sequence number, stored LAST (0: this report does not take part)
};
static_assert(sizeof(Report) == 256, "report layout");

// `out` and `pairs_out` are PINNED HOST memory (zero-copy): the kernel posts the 256-byte record and the first
// min(n_pairs, spec_n) pairs straight over the host link, so the step ends with a stream synchronise instead of a
// DMA copy (whose set-up idles the GPU for ~12 us and runs ~5 us).  Block 0 / wave 0 writes the record.
// Polled completion (seq != 0): the host does not wait for the stream at all -- it spins on Report::seq in its own memory.  Every
// workgroup makes its part visible to the host (system-scope fence), then arrives on a device counter; the last one to arrive
// stores the sequence number with a system-scope release (the memory-model argument: DESIGN.md section 6), after putting the counter back
// to zero for the next report (of this step -- the deep pass, whose grid may differ -- or of the next one).
constexpr int REPORT_THREADS = 256;
__global__ __launch_bounds__(REPORT_THREADS) void k_report(const TravState *__restrict__ st, const uint32_t *__restrict__ sort_flags /* 9 words */,
                                                           const double *__restrict__ root_box, Report *__restrict__ out,
                                                           const uint32_t *__restrict__ pairs, uint32_t *__restrict__ pairs_out, unsigned long long spec_n,
                                                           unsigned long long seq)
{
    const unsigned long long np = st->n_pairs;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const int lane = threadIdx.x;                                   // NSHARD == 64: lane = shard
        const CtrShard sh = st->shard[lane];
        const unsigned long long tested = wave_sum_u64(sh.pairs_tested), visits = wave_sum_u64(sh.node_visits);
        const unsigned long long steps = wave_sum_u64(sh.wave_steps), cands = wave_sum_u64(cand_entries(sh.n_candidates));
        unsigned long long mx = cand_slots(sh.n_candidates), c0 = sh.pad[4], c1 = sh.pad[11];      // (max_shard_candidates: in 8-byte slots, what the capacity is counted in)
        for (int o = 32; o; o >>= 1) {
            const unsigned long long u = __shfl_xor(mx, o); mx = u > mx ? u : mx;
            const unsigned long long u0 = __shfl_xor(c0, o); c0 = u0 > c0 ? u0 : c0;
            const unsigned long long u1 = __shfl_xor(c1, o); c1 = u1 > c1 ? u1 : c1;
        }
        if (lane == 0) {
            out->n_pairs = np; out->pairs_tested = tested; out->node_visits = visits; out->max_shard_candidates = mx;
            out->wave_steps = steps; out->candidates = cands; out->n_deferred = st->n_deferred;
            out->clk_start_inv = c0; out->clk_end = c1;
        }
        if (lane < 9) out->sort_flags[lane] = sort_flags[lane];
        if (lane < 6) out->root_box[lane] = root_box[lane];
    }
    const unsigned long long take = np < spec_n ? np : spec_n;          // pairs are 8 bytes; move them as 16-byte quads, an odd last pair on its own
    const unsigned long long quads = take >> 1;                         // (pairs_out may be the caller's buffer: nothing is written past the list)
    const uint4 *src = reinterpret_cast<const uint4 *>(pairs);
    uint4 *dst = reinterpret_cast<uint4 *>(pairs_out);
    for (unsigned long long i = (unsigned long long)blockIdx.x * REPORT_THREADS + threadIdx.x; i < quads; i += (unsigned long long)gridDim.x * REPORT_THREADS) dst[i] = src[i];
    if ((take & 1ull) && blockIdx.x == 0 && threadIdx.x == 64) reinterpret_cast<uint2 *>(pairs_out)[take - 1] = reinterpret_cast<const uint2 *>(pairs)[take - 1];
    if (seq == 0ull) return;                                            // (uniform over the grid)
    __threadfence_system();                                             // this thread's host writes (a fence orders the calling thread's stores: EVERY thread fences)
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t old = atomicAdd(const_cast<uint32_t *>(&st->report_arrive), 1u);
        if (old == gridDim.x - 1u) {                                    // the last workgroup to arrive
            atomicExch(const_cast<uint32_t *>(&st->report_arrive), 0u);  // a second report of the same step (deep pass, another grid size) starts from zero
            __threadfence_system();
            __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

constexpr int TRAV_THREADS = 256;
constexpr int TRAV_STACK   = 32;       // variant A: LDS entries per lane (the reference's private stack is 32, collision.cuh:21)
constexpr int DEEP_STACK   = 192;      // global-memory entries per item in the overflow pass (tree height <= 96)

__device__ __forceinline__ bool band(bool a, bool b) { return a ? b : false; }
__device__ __forceinline__ bool bor(bool a, bool b) { return a ? true : b; }

struct QuerySrc {
    const LeafTri *leaf;          // local: sorted leaves
    const double  *boxes;         // local: node boxes (query box = boxes[(n-1)+j])
    const LeafBox32 *qbox;        // local: fp32 query boxes + flags (cd_bvh.h), what the fp32 descents read
    const int32_t *root;          // name (split) of the root record, written by the refit
    const void    *ext;           // external: cd_query records (88 B)
    const uint2   *list;          // deep pass: deferred (query index, subtree root) work items
    const uint32_t *sort_flags;   // 9 words, non-zero = the sort of this pipeline run failed (look-back time-out, or a half-key
                                  // run longer than the fix-up handles): the tree behind it is not a tree, do not walk it
};

// A fused call enqueues the traversal before the host has seen the sort's flags; keys that are not sorted can give
// child links with cycles, and a descent over them would never end.  Every traversal kernel starts with this
// (wave-uniform, 9 scalar loads); the host reads the same flags afterwards and redoes the whole step.
// The same as a value (0: fine): a kernel that has in-bounds work to request first reads the flags at its start and looks at them later --
// their round trip then runs beside that work instead of in front of it (k_descend_half).
__device__ __forceinline__ uint32_t sort_flags_or(const QuerySrc &src)
{
    uint32_t any = 0;                                   // (no null test: every QuerySrc is built with the context's flag words, and a branch around the loads would wait for them inside it)
#pragma unroll
    for (int i = 0; i < 9; ++i) any |= src.sort_flags[i];
    return any;
}
__device__ __forceinline__ bool sort_failed(const QuerySrc &src)
{
    if (!src.sort_flags) return false;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) any |= src.sort_flags[i];
    return any != 0;
}

struct ExtQuery { double v[9]; uint32_t id; uint32_t vidx[3]; };
static_assert(sizeof(ExtQuery) == 88, "cd_query layout");

// ====================================================================================================
// Variant A ("lane-private"): the reference's shape (collision.cuh:19-71) -- one query per lane, FP64 boxes
// (node topology record + two 48-byte child boxes per visit), the exact test inline in the descent.
// Kept as the in-process A/B baseline and as an independent second implementation for the tests.
// ====================================================================================================
__device__ __forceinline__ void leaf_hit(uint32_t q_id, uint32_t qa, uint32_t qb, uint32_t qc,
                                         const d3 &P1, const d3 &P2, const d3 &P3,
                                         const LeafTri lt, const double *__restrict__ verts,
                                         uint32_t *__restrict__ pairs, unsigned long long cap, TravState *__restrict__ st,
                                         uint32_t vbase)
{
    // vbase: global id of local vertex 0 -- non-zero only for external (cross-rank) queries, whose
    // vertex ids are global; local queries compare local indices with local indices (vbase = 0).
    if (neighbor_count(qa, qb, qc, lt.v0 + vbase, lt.v1 + vbase, lt.v2 + vbase) < 1) {   // collision.cuh:38
        if (q_id < lt.id) {                                               // tri_contact.cuh:81
            if (tri_contact(P1, P2, P3, load_vertex(verts, lt.v0), load_vertex(verts, lt.v1), load_vertex(verts, lt.v2))) {
                const unsigned long long cur = atomicAdd(&st->n_pairs, 1ull);   // collision.cuh:40
                if (cur < cap) { pairs[2 * cur] = q_id; pairs[2 * cur + 1] = lt.id; }
            }
        }
    }
}

template <bool EXTERNAL, bool DEEP>
__global__ __launch_bounds__(TRAV_THREADS) void k_traverse(QuerySrc src, uint32_t nq, int n,
                                                           const NodeMeta *__restrict__ meta, const double *__restrict__ boxes,
                                                           const LeafTri *__restrict__ leaf, const double *__restrict__ verts,
                                                           uint32_t *__restrict__ pairs, unsigned long long cap,
                                                           TravState *__restrict__ st,
                                                           uint2 *__restrict__ defer_list, uint32_t defer_cap,
                                                           int32_t *__restrict__ deep_stacks, uint32_t vbase)
{
    if (sort_failed(src)) return;
    __shared__ int32_t lds_stack[DEEP ? 1 : TRAV_STACK][TRAV_THREADS];
    const uint32_t tid = threadIdx.x;
    const uint32_t gi = blockIdx.x * TRAV_THREADS + tid;
    unsigned long long tested = 0, visits = 0;
    if (gi < nq) {
        const uint32_t qi = DEEP ? src.list[gi].x : gi;
        uint32_t q_id, qa, qb, qc; d3 P1, P2, P3; Box qbox;
        if (EXTERNAL) {
            const ExtQuery *q = reinterpret_cast<const ExtQuery *>(src.ext) + qi;
            P1 = d3{q->v[0], q->v[1], q->v[2]}; P2 = d3{q->v[3], q->v[4], q->v[5]}; P3 = d3{q->v[6], q->v[7], q->v[8]};
            q_id = q->id; qa = q->vidx[0]; qb = q->vidx[1]; qc = q->vidx[2];
            qbox = box_set(P1, P2, P3);
        } else {
            const LeafTri lt = src.leaf[qi];
            q_id = lt.id; qa = lt.v0; qb = lt.v1; qc = lt.v2;
            P1 = load_vertex(verts, qa); P2 = load_vertex(verts, qb); P3 = load_vertex(verts, qc);
            qbox = load_box(src.boxes, (n - 1) + (int)qi);                 // collision.cuh:86 &leaves[i].box
        }
        int32_t *gstack = DEEP ? deep_stacks + (size_t)gi * DEEP_STACK : nullptr;
        int sptr = 0;
        int32_t node = DEEP ? (int32_t)src.list[gi].y : ((n > 1) ? 0 : -1); // root = internal[0]
        while (node != -1) {
            ++visits;
            const NodeMeta m = meta[node];
            const int32_t cl = m.x, cr = m.y;
            const Box bl = load_box(boxes, cl), br = load_box(boxes, cr);
            const bool ol = box_overlap(qbox, bl);                         // collision.cuh:31-32
            const bool orr = box_overlap(qbox, br);
            int32_t next = -1;
            if (ol) {
                if (cl >= n - 1) { ++tested; leaf_hit(q_id, qa, qb, qc, P1, P2, P3, leaf[cl - (n - 1)], verts, pairs, cap, st, vbase); }
                else next = cl;
            }
            if (orr) {
                if (cr >= n - 1) { ++tested; leaf_hit(q_id, qa, qb, qc, P1, P2, P3, leaf[cr - (n - 1)], verts, pairs, cap, st, vbase); }
                else if (next == -1) next = cr;
                else {                                                     // both internal: descend left, push right
                    const int cap_s = DEEP ? DEEP_STACK : TRAV_STACK;
                    if (sptr < cap_s) { if (DEEP) gstack[sptr] = cr; else lds_stack[sptr][tid] = cr; ++sptr; }
                    else {
                        // Stack full: hand the right subtree to the deep pass as its own work item and go on.
                        // Each subtree is still traversed exactly once, so no pair is reported twice.
                        const uint32_t k = atomicAdd(&st->n_deferred, 1u);
                        if (k < defer_cap) defer_list[k] = make_uint2(qi, (uint32_t)cr);
                    }
                }
            }
            if (next != -1) node = next;
            else if (sptr > 0) { --sptr; node = DEEP ? gstack[sptr] : lds_stack[sptr][tid]; }
            else node = -1;
        }
    }
    tested = wave_sum_u64(tested); visits = wave_sum_u64(visits);
    if ((tid & 63) == 0) {
        CtrShard *sh = &st->shard[blockIdx.x & (NSHARD - 1)];
        if (tested) atomicAdd(&sh->pairs_tested, tested);
        if (visits) atomicAdd(&sh->node_visits, visits);
    }
}

// ====================================================================================================
// Variant B ("split", default): the traversal the north star describes, as two kernels.
//   k_descend : pure fp32 + integer descent over 64-byte NodeRec32 lines (conservative boxes, ~48 VGPRs).
//       * the wave first walks the root path of its first leaf through the scalar cache and every lane only tests
//         the siblings hanging off it (self-collision queries are leaves of the tree they query); the private
//         descent starts from the siblings a lane overlaps;
//       * per-lane stack in LDS ([depth][thread], bank-conflict free); overflow hands the subtree to the deep pass;
//       * busy lanes hand pending subtrees, with their query, to idle lanes of the wave (work sharing);
//       * every leaf the fp32 test cannot rule out becomes a CANDIDATE (query, leaf) on a wavefront-shared LDS
//         queue, compacted over the active lanes with __ballot / popcount prefixes;
//       * a full batch of 64 candidates is handed over one per lane: candidates the descent decided exactly (boxes
//         that are fp32 values) are counted as tested and pass the neighbour filter and the ID rule right there;
//         the survivors, and everything that still needs the FP64 box test, go to the workgroup's shard of a
//         global candidate buffer, reserved by ONE lane with a single atomicAdd;
//       * with CD_OPT_QUERIES_PER_WAVE > 64, lanes whose query has finished take the next query of the wave's
//         chunk (dynamic refill) instead.
//   k_exact   : exact FP64 leaf-AABB test for the candidates that need it (box.cuh:40-43 -- this is what "pairs
//       tested" counts), neighbour filter (collision.cuh:38), ID rule (tri_contact.cuh:81), then the 17-axis SAT
//       (tri_contact.cuh:19-78) on full batches of survivors, no descent state kept alive.  Pairs are staged in
//       LDS and appended with one global atomic per workgroup.
// Candidate-shard overflow is detected from the reserved counts (nothing is written past a shard's capacity)
// and handled by the host: grow, redo.
// ====================================================================================================
constexpr int WQ_STACK = 12;                 // LDS stack entries per lane (12 KB + 6 KB queue per workgroup -> 8 workgroups = 32 waves per CU)
constexpr int WQ_QCAP  = 192;                // queue slots per wave: < 64 left over + at most 128 new per step
constexpr int WQ_WAVES = TRAV_THREADS / 64;
constexpr int SHARE_MIN_IDLE = 16;           // idle lanes in a wave before busy lanes hand subtrees over (65 = never); 4 .. 16 measured equal, 1 and 32 worse

// leaf: bit 31 set = both boxes are CERTAIN (cd_bvh.h: every bound in an unambiguous cell), so the fp32 overlap found by the descent IS
// the exact leaf-AABB test and k_exact need not fetch the two FP64 boxes again
struct Candidates { uint32_t q, leaf; };
constexpr uint32_t CAND_CERTAIN = 0x80000000u;
// bit 30 (only together with bit 31): the descent has also applied the neighbour filter and the ID rule, and counted
// the pair as tested -- k_exact sends it straight to the SAT
constexpr uint32_t CAND_FILTERED = 0x40000000u;
constexpr uint32_t CAND_LEAF_MASK = 0x3fffffffu;
// A pair the half traversal has decided, counted and filtered leaves the descent with BOTH leaves' records (which its filter had to fetch anyway): k_exact starts the six
// vertex loads of the SAT straight from the entry -- one dependent round trip (the two leaf records by index) less in a kernel that is a chain of them.  32-byte entries,
// reserved from the END of the workgroup's shard downwards (entry k: slots cap - 4 (k + 1) .. cap - 4 k - 1; the 8-byte candidates grow from the front; the two meet
// only in a step whose shard overflowed, which is redone).  shard_cap is a multiple of 4 (mi355cd.hip), the buffer 256-byte aligned.
struct alignas(16) FatPair { LeafTri a, b; };
static_assert(sizeof(FatPair) == 4 * sizeof(Candidates), "a SAT-ready pair takes four candidate slots");
__device__ __forceinline__ FatPair *fat_slot(Candidates *shard, unsigned long long shard_cap, unsigned long long k) { return reinterpret_cast<FatPair *>(shard + shard_cap) - (k + 1); }
__device__ __forceinline__ const FatPair *fat_slot(const Candidates *shard, unsigned long long shard_cap, unsigned long long k) { return reinterpret_cast<const FatPair *>(shard + shard_cap) - (k + 1); }

// queries_per_wave: size of the contiguous chunk of queries one wave works through (multiple of 64).
// (one wave per workgroup, as k_descend_half: the waves of this kernel never meet)
constexpr int DESC_THREADS = 64, DESC_WAVES = DESC_THREADS / 64;
template <bool EXTERNAL, bool DEEP, bool REFILL>
__global__ __launch_bounds__(DESC_THREADS) void k_descend(QuerySrc src, uint32_t nq, int n, uint32_t queries_per_wave,
                                                          const NodeRec32 *__restrict__ recs, const double *__restrict__ boxes,
                                                          TravState *__restrict__ st,
                                                          Candidates *__restrict__ cand, unsigned long long shard_cap,
                                                          uint2 *__restrict__ defer_list, uint32_t defer_cap,
                                                          int32_t *__restrict__ deep_stacks, uint32_t vbase, uint32_t half /* see k_descend_half: deep pass of a half traversal */)
{
    if (sort_failed(src)) return;
    __shared__ int32_t lds_stack[DEEP ? 1 : WQ_STACK][DESC_THREADS];
    __shared__ Candidates queue[DESC_WAVES][WQ_QCAP];
    __shared__ uint8_t share_map[DESC_WAVES][64];      // work sharing: lane id of the k-th donor
    const int32_t root = (n > 1) ? *src.root : -1;     // records are named by split (cd_bvh.h): the root's name comes from the refit
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD and
    // its 4 MiB L2).  Give each XCD one CONTIGUOUS eighth of the Morton-ordered queries, so the subtrees its
    // waves walk overlap and stay in that L2, instead of every L2 seeing the whole tree.  Speed only: any
    // mapping that is a permutation of the workgroups gives the same results.
    const uint32_t nb = gridDim.x, per = nb >> 3;
    const uint32_t vblock = (blockIdx.x < (per << 3)) ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t wave_id = vblock * DESC_WAVES + w;
    CtrShard *sh = &st->shard[blockIdx.x & (NSHARD - 1)];
    Candidates *my_cand = cand + (size_t)(blockIdx.x & (NSHARD - 1)) * shard_cap;
    // this wave's chunk of work items (queries, or deferred (query, subtree) items in the deep pass)
    const uint32_t qpw_ = queries_per_wave & 0x3fffffffu;       // bit 30: debug switch (no shared root path)
    const unsigned long long c0 = (unsigned long long)wave_id * qpw_;
    const uint32_t chunk_begin = (uint32_t)(c0 < nq ? c0 : nq);
    const uint32_t chunk_end = (uint32_t)(c0 + qpw_ < nq ? c0 + qpw_ : nq);
    uint32_t next = chunk_begin;                        // wave-uniform: next unassigned work item
    uint32_t qcount = 0;                                // wave-uniform: candidates waiting in the queue
    uint32_t tested = 0, visits = 0, steps = 0;
    int32_t node = -1; int sptr = 0; uint32_t qi = 0, self_leaf = 0xffffffffu;
    // Hand the last `count` (<= 64) queued candidates over to k_exact, one per lane.  A candidate the descent has
    // decided exactly (CAND_CERTAIN) is a tested pair whatever follows, and all that stands between it and the SAT are
    // two cheap filters on 16-byte records -- so they run here, with full lanes, and only the ~2 % that survive (plus
    // everything that still needs the FP64 box test) travel through memory to k_exact.
    auto flush = [&](uint32_t count) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        Candidates cd0 = Candidates{0, 0};
        bool keep = lane < count;
        if (keep) cd0 = queue[w][qcount - count + lane];
        if (keep && (cd0.leaf & CAND_CERTAIN)) {
            tested += half ? 2u : 1u;                                          // collision.cuh:31-32 decided (exactly) by the descent (half: both directions)
            const LeafTri lt = src.leaf[cd0.leaf & CAND_LEAF_MASK];
            uint32_t q_id, qa, qb, qc;
            if (EXTERNAL) {
                const ExtQuery *q = reinterpret_cast<const ExtQuery *>(src.ext) + cd0.q;
                q_id = q->id; qa = q->vidx[0]; qb = q->vidx[1]; qc = q->vidx[2];
            } else {
                const LeafTri ql = src.leaf[cd0.q];
                q_id = ql.id; qa = ql.v0; qb = ql.v1; qc = ql.v2;
            }
            keep = neighbor_count(qa, qb, qc, lt.v0 + vbase, lt.v1 + vbase, lt.v2 + vbase) < 1 && (half ? q_id != lt.id : q_id < lt.id);   // collision.cuh:38, tri_contact.cuh:81
            cd0.leaf |= CAND_FILTERED;
        }
        qcount -= count;
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(keep);
        if (mk != 0ull) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&sh->n_candidates, (unsigned long long)__popcll(mk));
            base = __shfl(base, 0) + __popcll(mk & lt_mask);
            if (keep && base < shard_cap) my_cand[base] = cd0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    float qlo0 = 0, qlo1 = 0, qlo2 = 0, qhi0 = 0, qhi1 = 0, qhi2 = 0;
    uint32_t qcertain = 0;                              // CAND_CERTAIN if the query box is exact in fp32
    int32_t *gstack = DEEP ? deep_stacks + ((size_t)wave_id * 64 + lane) * DEEP_STACK : nullptr;
    const int stack_cap = DEEP ? DEEP_STACK : WQ_STACK;

    // shared root path (below): only for a wave whose queries are 64 consecutive leaves of the tree itself
    constexpr bool SHARED_PATH = !EXTERNAL && !DEEP && !REFILL;
    bool shared_path_pending = SHARED_PATH && n > 1 && !(queries_per_wave & 0x40000000u) && chunk_begin < chunk_end;
    bool first_round = true;
    while (true) {
        if (REFILL || first_round) {   // ---- (re)fill idle lanes with the next work items of the chunk
            first_round = false;
            const bool idle = (node == -1);
            const unsigned long long mi = __ballot(idle);
            const uint32_t remaining = chunk_end - next;
            if (mi != 0ull && remaining != 0u) {
                const uint32_t rank = __popcll(mi & lt_mask);
                if (idle && rank < remaining) {
                    const uint32_t item = next + rank;
                    if (DEEP) { qi = src.list[item].x; node = (int32_t)src.list[item].y; }
                    else { qi = item; node = root; }
                    if (EXTERNAL) {
                        const ExtQuery *q = reinterpret_cast<const ExtQuery *>(src.ext) + qi;
                        const Box qb = box_set(d3{q->v[0], q->v[1], q->v[2]}, d3{q->v[3], q->v[4], q->v[5]}, d3{q->v[6], q->v[7], q->v[8]});
                        self_leaf = 0xffffffffu;
                        qlo0 = __double2float_rd(qb.x1); qhi0 = __double2float_ru(qb.x2);
                        qlo1 = __double2float_rd(qb.y1); qhi1 = __double2float_ru(qb.y2);
                        qlo2 = __double2float_rd(qb.z1); qhi2 = __double2float_ru(qb.z2);
                        qcertain = box_is_fp32(qb) ? CAND_CERTAIN : 0u;     // (a query from another mesh: certain against EXACT leaves only, see the step below)
                    } else {
                        const float4 *qp = reinterpret_cast<const float4 *>(src.qbox + qi);
                        const float4 q0 = qp[0], q1 = qp[1];
                        const uint32_t qfl = __float_as_uint(q1.z);
                        qlo0 = q0.x; qlo1 = q0.y; qlo2 = q0.z; qhi0 = q0.w; qhi1 = q1.x; qhi2 = q1.y;
                        qcertain = (qfl & LB_CERTAIN) ? CAND_CERTAIN : 0u;
                        self_leaf = qi;
                        // the query meets its own leaf in every traversal: that hit was decided by the refit, exactly, once
                        // (not counted in the deep pass, which only continues a traversal that already did)
                        if (!DEEP && n > 1 && (qfl & LB_SELF)) ++tested;           // (a single triangle has no tree to walk: nothing is tested)
                    }
                    sptr = 0;
                }
                const uint32_t taken = __popcll(mi);
                next += (taken < remaining) ? taken : remaining;
            }
        }
        if (SHARED_PATH && shared_path_pending) {
            // ---- shared root path.  The 64 queries of this wave are the leaves [g0, g0 + 64) of the very tree they
            // query, so each of them would walk down (nearly) the same ~20 ancestors, one divergent 64-byte fetch and
            // ~50 instructions per lane and level.  Instead the WAVE walks from the root to leaf g0 once, fetching
            // each record through the scalar cache, and every lane only tests its box against the SIBLING hanging off
            // that path: the siblings plus leaf g0 partition all leaves, child boxes nest (also after the monotone
            // outward rounding), so the lanes reach exactly the subtrees a private descent from the root would.
            shared_path_pending = false;
            const bool valid = (node != -1);
            // (readfirstlane: the compiler cannot see that chunk_begin is wave-uniform, and would select per lane)
            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk_begin);
            int32_t pn = __builtin_amdgcn_readfirstlane(root);              // wave-uniform path node
            node = -1;
            for (int lev = 0; lev < 128; ++lev) {                            // (tree height <= 96: the bound only matters for a corrupt tree)
                const int4 *rql = reinterpret_cast<const int4 *>(rec_left(recs, n, (uint32_t)pn)), *rqr = reinterpret_cast<const int4 *>(rec_right(recs, n, (uint32_t)pn));
                const int4 r0 = rql[0], r1 = rql[1], r2 = rqr[0], r3 = rqr[1];
                const int32_t chl = r1.z, chr = r3.z;
                const uint32_t split = (uint32_t)pn;                           // a record's name IS its split position
                const bool go_left = g0 <= split;                             // uniform
                const int32_t sib = go_left ? chr : chl, onp = go_left ? chl : chr;
                const uint32_t sib_exact = (uint32_t)r3.w & (go_left ? REC_R_CERTAIN : REC_L_CERTAIN), onp_exact = (uint32_t)r3.w & (go_left ? REC_L_CERTAIN : REC_R_CERTAIN);
                // record layout: left = lo(r0.x,r0.y,r0.z) hi(r0.w,r1.x,r1.y), right = lo(r2.x,r2.y,r2.z) hi(r2.w,r3.x,r3.y).
                // The selects are done on the raw words, so that they stay scalar (s_cselect) like their condition.
                const float slo0 = __int_as_float(go_left ? r2.x : r0.x), slo1 = __int_as_float(go_left ? r2.y : r0.y), slo2 = __int_as_float(go_left ? r2.z : r0.z);
                const float shi0 = __int_as_float(go_left ? r2.w : r0.w), shi1 = __int_as_float(go_left ? r3.x : r1.x), shi2 = __int_as_float(go_left ? r3.y : r1.y);
                const bool hit = valid & (qlo0 < shi0) & (slo0 < qhi0) & (qlo1 < shi1) & (slo1 < qhi1) & (qlo2 < shi2) & (slo2 < qhi2);
                visits += valid ? 1u : 0u;
                bool cnd = false; uint32_t cleaf = 0;
                if (sib >= 0) {                                               // uniform branch
                    if (hit) {
                        if (sptr < stack_cap) { lds_stack[sptr][tid] = sib; ++sptr; }
                        else { const uint32_t k = atomicAdd(&st->n_deferred, 1u); if (k < defer_cap) defer_list[k] = make_uint2(qi, (uint32_t)sib); }
                    }
                } else { cleaf = (uint32_t)~sib; cnd = hit & (cleaf != self_leaf); }
                if (onp < 0) {                                                // the path ends at leaf g0 (uniform): its box is the on-path child's
                    const float plo0 = __int_as_float(go_left ? r0.x : r2.x), plo1 = __int_as_float(go_left ? r0.y : r2.y), plo2 = __int_as_float(go_left ? r0.z : r2.z);
                    const float phi0 = __int_as_float(go_left ? r0.w : r2.w), phi1 = __int_as_float(go_left ? r1.x : r3.x), phi2 = __int_as_float(go_left ? r1.y : r3.y);
                    const bool ph = valid & (qi != g0) & (qlo0 < phi0) & (plo0 < qhi0) & (qlo1 < phi1) & (plo1 < qhi1) & (qlo2 < phi2) & (plo2 < qhi2);
                    const unsigned long long mP = __builtin_amdgcn_ballot_w64(ph);
                    if (mP) { if (ph) queue[w][qcount + __popcll(mP & lt_mask)] = Candidates{qi, g0 | (onp_exact ? qcertain : 0u)}; qcount += __popcll(mP); }
                }
                const unsigned long long mC = __builtin_amdgcn_ballot_w64(cnd);
                if (mC) { if (cnd) queue[w][qcount + __popcll(mC & lt_mask)] = Candidates{qi, cleaf | (sib_exact ? qcertain : 0u)}; qcount += __popcll(mC); }
                while (qcount >= 64) flush(64);
                if (onp < 0) break;
                pn = __builtin_amdgcn_readfirstlane(onp);
            }
            if (valid & (sptr > 0)) { --sptr; node = lds_stack[sptr][tid]; }
        }
        // ---- work sharing inside the wave.  After the shared path a lane only has the few subtrees its own query
        // overlaps: lanes finish at very different times (24 of 64 busy on average; the busiest wave ran 78 steps
        // against a mean of 26, and the kernel ends with its last wave).  A busy lane that still has a subtree on
        // its stack hands the top one, together with its query (index, fp32 box, flags), to an idle lane, which
        // descends it as if it were its own: every (query, subtree) is still descended exactly once, by someone.
        if (!DEEP && SHARE_MIN_IDLE <= 64) {
            const bool idle = (node == -1), donor = (node != -1) & (sptr > 0);
            const unsigned long long m_idle = __builtin_amdgcn_ballot_w64(idle), m_don = __builtin_amdgcn_ballot_w64(donor);
            if (m_don != 0ull && __popcll(m_idle) >= SHARE_MIN_IDLE && (REFILL ? next >= chunk_end : true)) {   // (wave-uniform)
                const uint32_t nx = min((uint32_t)__popcll(m_don), (uint32_t)__popcll(m_idle));
                const uint32_t rank_d = __popcll(m_don & lt_mask), rank_r = __popcll(m_idle & lt_mask);
                const bool give = donor & (rank_d < nx), take = idle & (rank_r < nx);
                int32_t top = -1;
                // (giving the OLDEST entry instead -- the highest in the tree, the biggest piece of work -- measured the same)
                if (give) { share_map[w][rank_d] = (uint8_t)lane; --sptr; top = lds_stack[sptr][tid]; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int src = take ? (int)share_map[w][rank_r] : (int)lane;
                const int32_t e = __shfl(top, src);
                const uint32_t s_qi = __shfl(qi, src), s_self = __shfl(self_leaf, src), s_cert = __shfl(qcertain, src);
                const float f0 = __shfl(qlo0, src), f1 = __shfl(qlo1, src), f2 = __shfl(qlo2, src);
                const float f3 = __shfl(qhi0, src), f4 = __shfl(qhi1, src), f5 = __shfl(qhi2, src);
                if (take) { node = e; qi = s_qi; self_leaf = s_self; qcertain = s_cert; qlo0 = f0; qlo1 = f1; qlo2 = f2; qhi0 = f3; qhi1 = f4; qhi2 = f5; sptr = 0; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        const bool active = (node != -1);
        if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;            // chunk exhausted and every lane finished
        ++steps;

        // ---- one descent step per active lane (fp32, conservative).  Written as straight-line selects: the
        // kernel is instruction-issue bound (SQ_ACTIVE_INST_ANY ~ 70 % of its duration), so every exec-mask
        // save/restore the compiler does not have to emit is time.  Idle lanes fetch the root record and ignore it.
        const uint32_t rn = active ? (uint32_t)node : 0u;
        const float4 *rpl = rec_left(recs, n, rn), *rpr = rec_right(recs, n, rn);
        const float4 a = rpl[0], b = rpl[1], c = rpr[0], d = rpr[1];
        // child links: internal -> its split (>= 0), leaf j -> ~j; ch.z: the leaf children's flags -- CERTAIN (in `last`) for this mesh's own
        // queries, EXACT (in `first`) for queries from another mesh
        const int4 ch = make_int4(__float_as_int(b.z), __float_as_int(d.z), (int)(__float_as_uint(EXTERNAL ? b.w : d.w) >> 30), 0);
        // left: lo = (a.x, a.y, a.z) hi = (a.w, b.x, b.y); right: lo = (c.x, c.y, c.z) hi = (c.w, d.x, d.y)
        // (bitwise & on purpose: && would be lowered to short-circuit branches with exec-mask juggling)
        bool ol  = active & (qlo0 < a.w) & (a.x < qhi0) & (qlo1 < b.x) & (a.y < qhi1) & (qlo2 < b.y) & (a.z < qhi2);
        bool orr = active & (qlo0 < c.w) & (c.x < qhi0) & (qlo1 < d.x) & (c.y < qhi1) & (qlo2 < d.y) & (c.z < qhi2);
        if (EXTERNAL) {
            // The query's bounds are not in this mesh's cell table (cd_bvh.h): its box is rounded outward, and a node's hi' is only
            // known to be above every bound of THIS mesh in its cell.  lo_node < hi_query stays conservative as it is (hi_query is
            // rounded up); lo_query vs hi' needs the tie: equal fp32 values mean "maybe" -- unless both are fp32 values themselves
            // (the query box exact, the child an EXACT leaf), where equal means touching, exactly.
            const bool tmL = !band(qcertain != 0u, band(ch.x < 0, (ch.z & 1) != 0)), tmR = !band(qcertain != 0u, band(ch.y < 0, (ch.z & 2) != 0));
            const bool olt  = active & ((qlo0 < a.w) | ((qlo0 == a.w) & tmL)) & (a.x < qhi0) & ((qlo1 < b.x) | ((qlo1 == b.x) & tmL)) & (a.y < qhi1) &
                              ((qlo2 < b.y) | ((qlo2 == b.y) & tmL)) & (a.z < qhi2);
            const bool ort = active & ((qlo0 < c.w) | ((qlo0 == c.w) & tmR)) & (c.x < qhi0) & ((qlo1 < d.x) | ((qlo1 == d.x) & tmR)) & (c.y < qhi1) &
                              ((qlo2 < d.y) | ((qlo2 == d.y) & tmR)) & (c.z < qhi2);
            ol = olt; orr = ort;
        }
        visits += active ? 1u : 0u;
        // (selects on i1, not integer & of promoted bools: the masks then stay in SGPR pairs instead of being
        // materialised as 0 / 1 in VGPRs and compared again)
        const bool intL = band(ol, ch.x >= 0), intR = band(orr, ch.y >= 0);
        const uint32_t leafL = (uint32_t)~ch.x, leafR = (uint32_t)~ch.y;
        const bool candL = band(band(ol, ch.x < 0), leafL != self_leaf), candR = band(band(orr, ch.y < 0), leafR != self_leaf);
        int32_t nxt = intL ? ch.x : (intR ? ch.y : -1);
        if (band(intL, intR)) {                                            // both internal: descend left, push right
            if (sptr < stack_cap) { if (DEEP) gstack[sptr] = ch.y; else lds_stack[sptr][tid] = ch.y; ++sptr; }
            else {
                // Stack full: hand the right subtree to the deep pass as its own work item and go on.
                // Each subtree is still traversed exactly once, so no pair is reported twice.
                const uint32_t k = atomicAdd(&st->n_deferred, 1u);
                if (k < defer_cap) defer_list[k] = make_uint2(qi, (uint32_t)ch.y);
            }
        }
        if (band(band(active, !bor(intL, intR)), sptr > 0)) { --sptr; nxt = DEEP ? gstack[sptr] : lds_stack[sptr][tid]; }   // (nxt < 0, said with the masks at hand)
        node = active ? nxt : -1;
        // ---- enqueue candidates, compacted over the active lanes (skipped wave-uniformly when there are none)
        // (the builtin takes the i1 as it is; __ballot(int) makes the compiler materialise 0 / 1 and compare again)
        const unsigned long long mL = __builtin_amdgcn_ballot_w64(candL), mR = __builtin_amdgcn_ballot_w64(candR);
        if ((mL | mR) == 0ull) continue;
        {
            const uint32_t nL = __popcll(mL);
            if (candL) queue[w][qcount + __popcll(mL & lt_mask)] = Candidates{qi, leafL | ((ch.z & 1) ? qcertain : 0u)};
            if (candR) queue[w][qcount + nL + __popcll(mR & lt_mask)] = Candidates{qi, leafR | ((ch.z & 2) ? qcertain : 0u)};
            qcount += nL + __popcll(mR);
        }
        while (qcount >= 64) flush(64);                          // full batch: one candidate per lane
    }
    if (qcount > 0) flush(qcount);                               // final partial batch
    const unsigned long long t64 = wave_sum_u64(tested), v64 = wave_sum_u64(visits);
    if (lane == 0) {
        if (t64) atomicAdd(&sh->pairs_tested, t64);
        if (v64) atomicAdd(&sh->node_visits, v64);
        if (steps) atomicAdd(&sh->wave_steps, (unsigned long long)steps);
    }
}

// ====================================================================================================
// Variant D ("half", default for self-collision): the same split as variant B, but a query only looks to its RIGHT.
//   Self-collision queries are the leaves of the tree they query, and every step of the reference's decision for a
//   (query a, leaf b) hit is symmetric in a and b up to the ID rule: the strict product-form box test
//   (box.cuh:40-43: the two factors swap, IEEE multiplication commutes), neighborCount (triangle.cuh:18-30), and
//   tri_contact.cuh:81, which lets exactly the direction with the smaller ID in front through.  So the reference's
//   traversal meets every overlapping pair of leaf boxes twice, once from each side, and reports it at most once.
//   Here the query at Morton position i meets only leaves at positions > i; each such hit counts as two tested
//   pairs, and k_exact runs the SAT with the smaller-ID triangle in front, exactly as the reference would.
//   * No walk from the root.  The leaves (i, n-1] are exactly the subtrees hanging off the root path of leaf i to
//     the right, and with records named by split (cd_bvh.h) that chain is reached bottom-up from the leaf itself:
//     s = i; { test the right child of recs[s]; s = recs[s].last; } until last == n-1 -- one 32-byte read and one
//     box test per hop, about half the tree height of hops, the lowest (the ones that hit) first.
//   * The wave owns the 64 consecutive leaves [g0, g_last].  A hop whose cursor s is below g_last reads the right half
//     of recs[s] with s inside that range -- a record one of the wave's own lanes was given at the start (one coalesced
//     fetch of rec_right[g0 .. g0 + 63]): phase 1a reads it out of that lane's registers through the LDS crossbar
//     (ds_bpermute), touching neither memory nor LDS storage.
//   * A lane whose cursor reaches g_last or beyond is on the chain of leaf g_last (the `last` values along a root
//     path are exactly that chain's cursors), which all lanes of the wave share from their joining point upwards:
//     phase 1b walks it once with SCALAR loads and every lane that has joined tests the wave-uniform box.
//   * Phase 1 (both parts) only records what it hits (internal sibling -> per-lane LDS stack, leaf -> candidate
//     queue); phase 2 is variant B's private descent of those subtrees, with work sharing inside the wave.  Nothing
//     in a sibling subtree is to the left of the query, so phase 2 needs no position test.
//   What was measured on the way (profiles/r02_experiments/): the kernel's time tracks the instructions it issues
//   (SQ_ACTIVE_INST_ANY x 4 cycles / 1024 SIMDs ~ its duration in every version), then occupancy; staging a 256-leaf
//   block of records in LDS made the hops cheap but cost 3 of 8 workgroups per CU (no gain); running the chain inside
//   the descent loop to overlap the two latencies added more instructions than it hid (slower).
// ====================================================================================================
constexpr int HALF_STACK = 8;                // LDS stack entries per lane
constexpr int HALF_QCAP = 192;               // candidate queue slots per wave
constexpr uint32_t HALF_FLUSH_AT = HALF_QCAP - 64;   // an enqueue adds at most 64 candidates: drain before it when more than this are waiting

// (one wave per workgroup: the waves of this kernel never meet -- no barrier, wave-private LDS -- and a workgroup holds its
//  slot until its SLOWEST wave is done; with the work per query as uneven as it is, single-wave workgroups give the slots
//  back sooner: 53.8 -> 52.4 us at 1 M triangles.  Two waves: 53.2 us)
constexpr int HALF_THREADS = 64;

constexpr int HALF_SMALL_N = 1 << 27;        // up to here a record's byte offset (32 bytes a node) fits 32 bits: the !BIG instances
template <bool DIAG, bool TIES /* the mesh has a cell table (cd_bvh.h): hits between boxes that are not both CERTAIN are looked at comparison by comparison */,
          bool BIG /* n > HALF_SMALL_N: phase 2 forms 64-bit record addresses */>
__global__ __launch_bounds__(HALF_THREADS, 8) void k_descend_half(QuerySrc src, int n, const NodeRec32 *__restrict__ recs,
                                                                  TravState *__restrict__ st,
                                                                  Candidates *__restrict__ cand, unsigned long long shard_cap,
                                                                  uint2 *__restrict__ defer_list, uint32_t defer_cap,
                                                                  const uint32_t *__restrict__ order /* NULL, or the order hint: workgroup -> group of 64 leaves */,
                                                                  const uint32_t *__restrict__ perm, uint8_t *__restrict__ tri_cost /* both NULL, or: sorted position -> triangle, and per TRIANGLE
                                                                                   how long its wave took, a class of 1.28 us (the next step's hint is made from these: cd_bvh.h) */)
{
    const uint32_t bad_sort = sort_flags_or(src);      // looked at after phase 0 (whose loads are in bounds whatever the tree is): see sort_flags_or
    // The kernel times ITSELF with the device's constant-rate wall clock (s_memrealtime): first wave start -> last wave end, two
    // sharded atomicMax per wave (the start as its complement, so that the zeroed counters need no initial value).  A HIP time
    // stamp on the dispatch packet costs the step ~7 us of idle GPU around the kernel; this costs it nothing measurable.
    const unsigned long long clk0 = __builtin_amdgcn_s_memrealtime();
    __shared__ int32_t lds_stack[HALF_STACK][HALF_THREADS];
    __shared__ Candidates queue[HALF_QCAP];
    __shared__ uint8_t share_map[64];                  // work sharing: lane id of the k-th donor
    const uint32_t lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t dg_p1a = 0, dg_p1 = 0, dg_hops_in = 0, dg_hops_out = 0, dg_vis = 0;   // diagnostics (DIAG)
    // XCD-aware mapping of workgroups to groups of 64 leaves: half_vblock (cd_bvh.h) -- or, when this step's build has left one, the ORDER HINT: the same
    // groups per XCD, the ones that took longest in the previous step first (cd_bvh.h, build_half_order).  Any bijection gives the same results.
    const uint32_t nb = ((uint32_t)n + 63u) / 64u;                          // (= gridDim.x, without the load of the hidden argument)
    const uint32_t vblock = order ? order[blockIdx.x] : half_vblock(blockIdx.x, nb);
    CtrShard *sh = &st->shard[blockIdx.x & (NSHARD - 1)];
    Candidates *my_cand = cand + (size_t)(blockIdx.x & (NSHARD - 1)) * shard_cap;
    const uint32_t nq = (uint32_t)n, last_leaf = nq - 1u;
    constexpr uint32_t END = 0xffffffffu;
    // the wave owns the 64 consecutive leaves [g0, g_last]  (wave-uniform; readfirstlane tells the compiler)
    const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(vblock * 64u));
    const uint32_t g_last = (g0 + 63u < last_leaf) ? g0 + 63u : last_leaf;
    uint32_t qi = g0 + lane;
    const bool valid = qi < nq && n > 1;
    const uint32_t my_tri = (tri_cost && qi < nq) ? perm[qi] : 0u;           // (asked for now, used at the very end: no wait of its own)
    unsigned long long tm0 = 0, tm1 = 0, tm2 = 0, tm3 = 0, tm4 = 0;           // DIAG: s_memtime stamps at the phase boundaries
    if constexpr (DIAG) tm0 = __builtin_amdgcn_s_memtime();
    uint32_t qcount = 0;                                // wave-uniform: candidates waiting in the queue
    uint32_t tested = 0, steps = 0;
    uint32_t wvisits = 0;                               // wave-uniform: node visits of the whole wave (a popcount of the active mask per step: two scalar instructions, no per-lane add)
    int sptr = 0;
    // Hand-over of queued candidates to k_exact, as in k_descend, with the half traversal's counting (both directions)
    // and ID rule (either order; k_exact puts the smaller ID in front).  The queue is drained only when it might not
    // hold the next enqueue, and once at the end: a wave usually hands everything over in one go.
    auto flush = [&](uint32_t count) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        Candidates cd0 = Candidates{0, 0};
        bool keep = lane < count, fat = false;
        if (keep) cd0 = queue[qcount - count + lane];
        FatPair fp;
        if (keep && (cd0.leaf & CAND_CERTAIN)) {
            tested += 2u;                                                      // collision.cuh:31-32, decided exactly by the descent, both directions
            fp.b = src.leaf[cd0.leaf & CAND_LEAF_MASK];
            fp.a = src.leaf[cd0.q];
            keep = fat = neighbor_count(fp.a.v0, fp.a.v1, fp.a.v2, fp.b.v0, fp.b.v1, fp.b.v2) < 1 && fp.a.id != fp.b.id;   // collision.cuh:38, tri_contact.cuh:81
        }
        qcount -= count;
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(keep);
        if (mk != 0ull) {
            // survivors of the filter go as SAT-ready pairs (FatPair) to the end of the shard, what still needs the FP64 box test as a candidate to its front
            const unsigned long long mf = __builtin_amdgcn_ballot_w64(fat), mt = mk & ~mf;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&sh->n_candidates, (unsigned long long)__popcll(mt) | ((unsigned long long)__popcll(mf) << 32));
            const uint32_t base_t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), base_f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
            if (fat) { const uint32_t k = base_f + (uint32_t)__popcll(mf & lt_mask); if (4ull * ((unsigned long long)k + 1ull) <= shard_cap) *fat_slot(my_cand, shard_cap, k) = fp; }
            if (mt != 0ull) {                                                     // (wave-uniform; rare: a mesh whose vertices are fp32 values has none)
                const uint32_t k = base_t + (uint32_t)__popcll(mt & lt_mask);
                if (keep && !fat && k < shard_cap) my_cand[k] = cd0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    // one candidate per lane where `c` holds (compacted over the wave with ballot / popcount prefixes)
    auto enqueue = [&](bool c, uint32_t q, uint32_t leafword) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(c);
        if (m == 0ull) return;
        while (qcount > HALF_FLUSH_AT) flush(64);
        if (c) queue[qcount + __popcll(m & lt_mask)] = Candidates{q, leafword};
        qcount += __popcll(m);
    };
    float qlo0, qlo1, qlo2, qhi0, qhi1, qhi2; uint32_t qcertain;
    // Is a leaf hit DECIDED by the fp32 test?  Yes when both boxes are CERTAIN (the leaf's flag from its parent's record, the query's
    // from phase 0).  If not, the six comparisons themselves still tell: an fp32 '<' is the FP64 '<' unless the two bounds lie in one
    // AMBIGUOUS cell (cd_bvh.h), and then hi' is exactly one ulp above lo' -- so a hit none of whose comparisons has next_up(lo') == hi'
    // is exact as well.  Only hits that fail this too go to k_exact for the FP64 box test.  (TIES instance only: a mesh without a
    // table -- every coordinate an fp32 value -- runs the instance without this code.)
    auto cand_word = [&](bool cnd, bool leaf_certain, float nlo0, float nlo1, float nlo2, float nhi0, float nhi1, float nhi2) -> uint32_t {
        uint32_t cw = leaf_certain ? qcertain : 0u;
        if constexpr (!TIES) return cw;                                       // (no table: every box is certain, nothing to look at)
        if (__builtin_amdgcn_ballot_w64(band(cnd, cw == 0u)) != 0ull) {
            const bool risky = (f32_next_up(qlo0) == nhi0) | (f32_next_up(nlo0) == qhi0) | (f32_next_up(qlo1) == nhi1) | (f32_next_up(nlo1) == qhi1) |
                               (f32_next_up(qlo2) == nhi2) | (f32_next_up(nlo2) == qhi2);
            if (!risky) cw = CAND_CERTAIN;
        }
        return cw;
    };
    // an internal sibling / child that was hit (`c`, per lane): onto the lane's stack, descended in phase 2
    auto note_subtree = [&](bool c, int32_t link) {
        if (c) {
            if (sptr < HALF_STACK) { lds_stack[sptr][lane] = link; ++sptr; }
            else { const uint32_t k = atomicAdd(&st->n_deferred, 1u); if (k < defer_cap) defer_list[k] = make_uint2(qi, (uint32_t)link); }
        }
    };
    // ---- the query box comes out of the RECORDS, not out of qbox[]: leaf j is the left child of recs[j] (then that
    // record's left link is ~j) or else the right child of recs[j - 1] -- every s is the split of exactly one node
    // (cd_bvh.h) -- and a leaf child's box in a record is the leaf's own fp32 box with its exact bit.  The lane needs the
    // right half of recs[j] anyway (phase 1a), the right half of recs[j - 1] is in the neighbour lane, and the left
    // halves are what phase 2 reads next: 32 bytes per leaf less from HBM than with a separate query array.
    // The last leaf has no record of its own: it is the right child of recs[n - 2].
    float4 rc = make_float4(0.f, 0.f, 0.f, 0.f), rd = rc;
    {
        float4 la = rc, lb = rc, pc = rc, pd = rc;
        const bool own = valid && qi < last_leaf;
        if (own) { const float4 *rp = rec_right(recs, n, qi); rc = rp[0]; rd = rp[1]; const float4 *lp = rec_left(recs, n, qi); la = lp[0]; lb = lp[1]; }
        {   // the right half of recs[qi - 1]: the neighbour lane's registers (DPP wave_shr:1, a VALU move); lane 0 gets the
            // record before the wave's first one through the scalar cache (wave-uniform address)
            int4 e0 = make_int4(0, 0, 0, 0), e1 = e0;
            if (g0 > 0u && g0 <= last_leaf && n > 1) { const int4 *pp = reinterpret_cast<const int4 *>(rec_right(recs, n, g0 - 1u)); e0 = pp[0]; e1 = pp[1]; }
            auto shr1 = [](float v, int edge) { return __int_as_float(__builtin_amdgcn_update_dpp(edge, __float_as_int(v), 0x138, 0xf, 0xf, false)); };
            pc.x = shr1(rc.x, e0.x); pc.y = shr1(rc.y, e0.y); pc.z = shr1(rc.z, e0.z); pc.w = shr1(rc.w, e0.w);
            pd.x = shr1(rd.x, e1.x); pd.y = shr1(rd.y, e1.y); pd.z = shr1(rd.z, e1.z); pd.w = shr1(rd.w, e1.w);
        }
        const bool is_left = own && __float_as_int(lb.z) == (int32_t)~qi;
        qlo0 = is_left ? la.x : pc.x; qlo1 = is_left ? la.y : pc.y; qlo2 = is_left ? la.z : pc.z;
        qhi0 = is_left ? la.w : pc.w; qhi1 = is_left ? lb.x : pd.x; qhi2 = is_left ? lb.y : pd.y;
        const bool exact = is_left ? (__float_as_uint(rd.w) & REC_L_CERTAIN) != 0u : (__float_as_uint(pd.w) & REC_R_CERTAIN) != 0u;   // (the leaf's box is CERTAIN)
        qcertain = exact ? CAND_CERTAIN : 0u;
        // the query's own leaf, which every traversal of the reference meets once (collision.cuh:31-32): box.cuh:40-43 with
        // a == b is (x1 - x2)^2 > 0 per axis -- for a CERTAIN box that is lo' < hi' (equal fp32 copies are equal doubles, different
        // ones differ by at least the spacing of two fp32 cells, whose square does not underflow in FP64), otherwise the FP64
        // box decides (stored for every leaf that is not EXACT)
        // (a box that is not certain: lo' < hi' is still the FP64 answer on an axis unless hi' is exactly one ulp above lo' -- cand_word below)
        const bool own_risky = !exact & ((f32_next_up(qlo0) == qhi0) | (f32_next_up(qlo1) == qhi1) | (f32_next_up(qlo2) == qhi2));
        bool self = !own_risky & (qlo0 < qhi0) & (qlo1 < qhi1) & (qlo2 < qhi2);
        if (valid && own_risky) { const Box b = load_box(src.boxes, (n - 1) + (int)qi); self = box_overlap(b, b); }
        if (valid && self) ++tested;
    }
    if constexpr (DIAG) tm1 = __builtin_amdgcn_s_memtime();
    if (bad_sort) return;                               // (wave-uniform) keys not sorted: the links may have cycles or point anywhere; the host redoes the step
    // ---- phase 1a: hops below g_last
    uint32_t s = valid ? qi : END;                                            // cursor; >= g_last: joined the shared chain (or has none: g_last == n-1)
    for (int hop = 0; hop < 128; ++hop) {                                     // tree height <= 96: the bound only matters for a corrupt tree
        const bool act = s < g_last;
        const unsigned long long m_act1 = __builtin_amdgcn_ballot_w64(act);
        if (m_act1 == 0ull) break;
        ++steps; wvisits += (uint32_t)__popcll(m_act1);
        const int from = (int)((act ? (s - g0) : lane) << 2);
        float4 c, d;
        c.x = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rc.x))); c.y = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rc.y)));
        c.z = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rc.z))); c.w = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rc.w)));
        d.x = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rd.x))); d.y = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rd.y)));
        d.z = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rd.z))); d.w = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(rd.w)));
        const bool hit = act & (qlo0 < c.w) & (c.x < qhi0) & (qlo1 < d.x) & (c.y < qhi1) & (qlo2 < d.y) & (c.z < qhi2);
        const int32_t link = __float_as_int(d.z);
        const uint32_t lw = __float_as_uint(d.w);
        if constexpr (DIAG) dg_hops_in += act ? 1u : 0u;
        note_subtree(band(hit, link >= 0), link);
        s = act ? (lw & REC_LAST_MASK) : s;
        const bool lh = band(hit, link < 0);
        enqueue(lh, qi, (uint32_t)~link | cand_word(lh, (lw & REC_R_CERTAIN) != 0u, c.x, c.y, c.z, c.w, d.x, d.y));
    }
    dg_p1a = steps;
    if constexpr (DIAG) tm2 = __builtin_amdgcn_s_memtime();
    // ---- phase 1b: a lane whose cursor has reached g_last or beyond is on the chain of leaf g_last (the `last` values
    // along a root path are exactly that chain's cursors), which all lanes share from their joining point upwards: the
    // wave walks it once with SCALAR loads, and every lane that has joined tests the wave-uniform box.
    {
        uint32_t t = g_last;
        for (int hop = 0; hop < 128 && t < last_leaf; ++hop) {
            ++steps;
            const int4 *rq = reinterpret_cast<const int4 *>(rec_right(recs, n, t));    // wave-uniform address: scalar loads
            const int4 c = rq[0], d = rq[1];
            const bool act = s <= t;                                          // (s == END for lanes without a query: never)
            const bool hit = act & (qlo0 < __int_as_float(c.w)) & (__int_as_float(c.x) < qhi0) & (qlo1 < __int_as_float(d.x)) &
                             (__int_as_float(c.y) < qhi1) & (qlo2 < __int_as_float(d.y)) & (__int_as_float(c.z) < qhi2);
            const int32_t link = d.z;                                         // wave-uniform
            const uint32_t lw = (uint32_t)d.w;
            wvisits += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(act));
            if constexpr (DIAG) dg_hops_out += act ? 1u : 0u;
            if (link >= 0) note_subtree(hit, link);
            else enqueue(hit, qi, (uint32_t)~link | cand_word(hit, (lw & REC_R_CERTAIN) != 0u, __int_as_float(c.x), __int_as_float(c.y), __int_as_float(c.z),
                                                               __int_as_float(c.w), __int_as_float(d.x), __int_as_float(d.y)));
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lw & REC_LAST_MASK));
        }
    }
    dg_p1 = steps;
    if constexpr (DIAG) tm3 = __builtin_amdgcn_s_memtime();
    {
        // ---- phase 2: descend the sibling subtrees that were hit
        int32_t node = -1;
        if (sptr > 0) { --sptr; node = lds_stack[sptr][lane]; }
        while (true) {
            // work sharing inside the wave (see k_descend): a busy lane hands the top of its stack, with its query, to an idle lane
            if (SHARE_MIN_IDLE <= 64) {
                // (a lane without a node has an empty stack -- it pops before it goes idle, and a taker starts with none: "donor" is sptr > 0, ONE compare, which the
                //  ballot takes as it is; an i1 made of ANDs and ORs it would first write out as 0 / 1 and compare again)
                const bool idle = (node == -1), donor = (sptr > 0);
                const unsigned long long m_idle = __builtin_amdgcn_ballot_w64(idle), m_don = __builtin_amdgcn_ballot_w64(donor);
                if (m_don != 0ull && __popcll(m_idle) >= SHARE_MIN_IDLE) {        // (wave-uniform)
                    const uint32_t nx = min((uint32_t)__popcll(m_don), (uint32_t)__popcll(m_idle));
                    const uint32_t rank_d = __popcll(m_don & lt_mask), rank_r = __popcll(m_idle & lt_mask);
                    const bool give = donor & (rank_d < nx), take = idle & (rank_r < nx);
                    int32_t top = -1;
                    if (give) { share_map[rank_d] = (uint8_t)lane; --sptr; top = lds_stack[sptr][lane]; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const int from = take ? (int)share_map[rank_r] : (int)lane;
                    const int32_t e = __shfl(top, from);
                    const uint32_t s_qi = __shfl(qi, from), s_cert = __shfl(qcertain, from);
                    const float f0 = __shfl(qlo0, from), f1 = __shfl(qlo1, from), f2 = __shfl(qlo2, from);
                    const float f3 = __shfl(qhi0, from), f4 = __shfl(qhi1, from), f5 = __shfl(qhi2, from);
                    if (take) { node = e; qi = s_qi; qcertain = s_cert; qlo0 = f0; qlo1 = f1; qlo2 = f2; qhi0 = f3; qhi1 = f4; qhi2 = f5; sptr = 0; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
            const bool active = (node != -1);
            const unsigned long long m_act2 = __builtin_amdgcn_ballot_w64(active);
            if (m_act2 == 0ull) break;
            ++steps; wvisits += (uint32_t)__popcll(m_act2);
            // one descent step per active lane, straight-line selects (see k_descend)
            const uint32_t rn = (uint32_t)node;
            // (!BIG: a 32-bit byte offset from the two arrays' wave-uniform bases -- the loads take it beside the base in scalar registers, three address instructions a step less)
            const float4 *rpl, *rpr;
            if constexpr (BIG) { rpl = rec_left(recs, n, rn); rpr = rec_right(recs, n, rn); }
            else {
                const uint32_t roff = rn << 5;
                rpl = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(rec_left(recs, n, 0)) + roff);
                rpr = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(rec_right(recs, n, 0)) + roff);
            }
            // (only the lanes with a node load: the others' a .. d stay indeterminate and every use below is ANDed with `active` -- no select for a harmless address,
            //  no register clears, and the texture unit steps through 18 lanes' addresses a step on average instead of 64)
            float4 a, b, c, d;
            if (active) { a = rpl[0]; b = rpl[1]; c = rpr[0]; d = rpr[1]; }
            const int32_t cl = __float_as_int(b.z), cr = __float_as_int(d.z);
            const uint32_t lw = __float_as_uint(d.w);
            const bool ol  = active & (qlo0 < a.w) & (a.x < qhi0) & (qlo1 < b.x) & (a.y < qhi1) & (qlo2 < b.y) & (a.z < qhi2);
            const bool orr = active & (qlo0 < c.w) & (c.x < qhi0) & (qlo1 < d.x) & (c.y < qhi1) & (qlo2 < d.y) & (c.z < qhi2);
            if constexpr (DIAG) dg_vis += active ? 1u : 0u;
            const bool intL = band(ol, cl >= 0), intR = band(orr, cr >= 0);
            int32_t nxt = intL ? cl : (intR ? cr : -1);
            note_subtree(band(intL, intR), cr);                                // both internal: descend left, push right
            if (band(band(active, !bor(intL, intR)), sptr > 0)) { --sptr; nxt = lds_stack[sptr][lane]; }
            node = active ? nxt : -1;
            const bool candL = band(ol, cl < 0), candR = band(orr, cr < 0);
            // (most steps meet no leaf: one test instead of two -- written as the compare of a register for the ballot's sake, see `donor`: a leaf link is negative)
            if (__builtin_amdgcn_ballot_w64(((ol ? cl : 0) | (orr ? cr : 0)) < 0) != 0ull) {
                enqueue(candL, qi, (uint32_t)~cl | cand_word(candL, (lw & REC_L_CERTAIN) != 0u, a.x, a.y, a.z, a.w, b.x, b.y));
                enqueue(candR, qi, (uint32_t)~cr | cand_word(candR, (lw & REC_R_CERTAIN) != 0u, c.x, c.y, c.z, c.w, d.x, d.y));
            }
        }
    }
    if constexpr (DIAG) tm4 = __builtin_amdgcn_s_memtime();
    while (qcount > 0) flush(qcount < 64u ? qcount : 64u);
    const unsigned long long t64 = wave_sum_u64(tested), v64 = wvisits;
    if (lane == 0) {
        if (t64) atomicAdd(&sh->pairs_tested, t64);
        if (v64) atomicAdd(&sh->node_visits, v64);
        if (steps) atomicAdd(&sh->wave_steps, (unsigned long long)steps);
        if (blockIdx.x < 256u) atomicMax(&sh->pad[4], ~clk0);                  // (the earliest start is among the first workgroups dispatched)
        atomicMax(&sh->pad[11], __builtin_amdgcn_s_memrealtime());
    }
    if (tri_cost) {                                                          // (wave-uniform) the wave's time goes with each of its triangles
        const unsigned long long cl = (__builtin_amdgcn_s_memrealtime() - clk0) >> ORDER_SHIFT;
        if (g0 + lane < nq) tri_cost[my_tri] = (uint8_t)(cl < (unsigned long long)(ORDER_CLASSES - 1) ? cl : (unsigned long long)(ORDER_CLASSES - 1));
    }
    if constexpr (DIAG) {
        const unsigned long long hi = wave_sum_u64(dg_hops_in), ho = wave_sum_u64(dg_hops_out), vi = wave_sum_u64(dg_vis);
        const unsigned long long tm5 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            atomicAdd(&sh->pad[0], (unsigned long long)(dg_p1 - dg_p1a)); atomicAdd(&sh->pad[1], hi); atomicAdd(&sh->pad[2], ho);
            atomicAdd(&sh->pad[3], vi); atomicAdd(&sh->pad[5], (unsigned long long)dg_p1a);
            atomicAdd(&sh->pad[6], tm1 - tm0); atomicAdd(&sh->pad[7], tm2 - tm1); atomicAdd(&sh->pad[8], tm3 - tm2);
            atomicAdd(&sh->pad[9], tm4 - tm3); atomicAdd(&sh->pad[10], tm5 - tm4);
        }
    }
}

constexpr int EXACT_THREADS = 256;
constexpr int EXACT_PB = 2048;               // pairs staged in LDS per workgroup before the single global append
constexpr int EXACT_ITEMS = 4;               // candidates per lane per round (independent loads in flight together)
constexpr int EXACT_SQ = 256 * (EXACT_ITEMS + 1);   // SAT queue slots per workgroup: < 256 left over + at most 256*ITEMS new per round

struct SatItem { uint32_t q, leaf; };

// k_exact, two stages per workgroup round so that the expensive part runs with full lanes:
//   stage 1 (one candidate per lane): exact FP64 leaf-AABB test (box.cuh:40-43) -> counts as "tested";
//            neighbour filter (collision.cuh:38), ID rule (tri_contact.cuh:81).  Cheap, rejects ~90 %.
//   stage 2: survivors wait on a workgroup-shared LDS queue; whenever >= 256 are queued every thread takes one
//            and runs the 17-axis SAT (tri_contact.cuh:19-78).  Running the SAT inside stage 1's branch kept
//            ~8 % of the lanes busy for ~1100 FP64 operations per wave -- it was most of this kernel's time.
template <bool EXTERNAL>
__global__ __launch_bounds__(EXACT_THREADS) void k_exact(QuerySrc src, int n, const LeafTri *__restrict__ leaf, const double *__restrict__ boxes,
                                                         const double *__restrict__ verts, uint32_t vbase,
                                                         const Candidates *__restrict__ cand, unsigned long long shard_cap,
                                                         uint32_t *__restrict__ pairs, unsigned long long cap, TravState *__restrict__ st,
                                                         uint32_t half /* candidates of a half traversal (k_descend_half): q and leaf are an unordered pair */,
                                                         uint32_t *__restrict__ post /* NULL, or PINNED HOST memory: the first post_n pairs are also stored there, at their positions in the
                                                                                        list -- the report kernel behind this one then posts the counters only (the pairs' 160 KB through
                                                                                        32 workgroups of a kernel of their own were 4.6 us of the step; here they leave with the appends) */,
                                                         unsigned long long post_n)
{
    __shared__ uint2 pbuf[EXACT_PB];
    __shared__ SatItem sq[EXACT_SQ];
    __shared__ uint32_t pcount, sqcount;
    __shared__ unsigned long long pbase;
    __shared__ uint32_t wtested[EXACT_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) { pcount = 0; sqcount = 0; }
    // (round 5) A workgroup OWNS a shard of the candidate buffer -- blockIdx & 63, the shard the descent's workgroups of that number filled -- and takes every
    // (gridDim / 64)-th chunk of 256 of it: where its candidates lie does not depend on the other shards' counts, so the first round's candidates are requested
    // TOGETHER with the counts (unconditional loads inside the shard's area, masked by the count afterwards) -- one round trip where the prefix over the 64 counts
    // and the search for a chunk's shard were two.  Every wave reads the 64 counts itself (one load instruction): no LDS, no barrier in front of the first fetch.
    const uint32_t my_shard = blockIdx.x & (NSHARD - 1), my_slot = blockIdx.x / NSHARD, slots = gridDim.x / NSHARD;   // (the host launches a multiple of NSHARD workgroups)
    const Candidates *my_cand = cand + (size_t)my_shard * shard_cap;
    // (as two arrays of words, not an array of structs: round 5's Candidates c_first[] stayed in scratch memory -- 48 bytes a lane and four stores that waited for the loads)
    uint32_t cfq[EXACT_ITEMS], cfl[EXACT_ITEMS];
#pragma unroll
    for (int j = 0; j < EXACT_ITEMS; ++j) {
        const unsigned long long k = ((unsigned long long)j * slots + my_slot) * EXACT_THREADS + tid;
        const uint2 w = reinterpret_cast<const uint2 *>(my_cand)[k < shard_cap ? k : 0];
        cfq[j] = w.x; cfl[j] = w.y;
    }
    // ... and so is the workgroup's first chunk of SAT-ready pairs (k_descend_half's survivors: FatPair, from the end of the shard)
    uint4 ff0, ff1;
    {
        const unsigned long long k = (unsigned long long)my_slot * EXACT_THREADS + tid;
        const uint4 *fp = reinterpret_cast<const uint4 *>(fat_slot(my_cand, shard_cap, 4ull * (k + 1ull) <= shard_cap ? k : 0ull));
        ff0 = fp[0]; ff1 = fp[1];
    }
    unsigned long long total, total_fat;                        // this shard's candidates and SAT-ready pairs
    {
        unsigned long long c = st->shard[lane].n_candidates;
        const bool over = cand_slots(c) > shard_cap;            // the host will grow the buffer and redo
        if (__ballot(over) != 0ull) c = 0;
        total = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)c, (int)my_shard);
        total_fat = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(c >> 32), (int)my_shard);
    }
    __syncthreads();                                            // (pcount, sqcount)
    uint32_t tested = 0;

    // a contact -> LDS pair staging (or direct append when the staging area is full)
    auto contact = [&](uint32_t q_id, uint32_t l_id) __attribute__((always_inline)) {
        const uint32_t slot = atomicAdd(&pcount, 1u);                          // LDS
        if (slot < EXACT_PB) pbuf[slot] = make_uint2(q_id, l_id);
        else {                                                                 // staging full: append directly (collision.cuh:40)
            const unsigned long long cur = atomicAdd(&st->n_pairs, 1ull);
            if (cur < cap) { pairs[2 * cur] = q_id; pairs[2 * cur + 1] = l_id; }
            if (post && cur < post_n) reinterpret_cast<uint2 *>(post)[cur] = make_uint2(q_id, l_id);
        }
    };
    // SAT of two leaves of this mesh.  Half traversal: the pair is unordered; the reference tests it with the smaller ID as the query
    // (tri_contact.cuh:81 lets only that direction through), so that triangle goes in front
    auto sat_leaves = [&](LeafTri ql, LeafTri lt) __attribute__((always_inline)) {
        if (half && ql.id > lt.id) { const LeafTri t = ql; ql = lt; lt = t; }
        const d3 P1 = load_vertex(verts, ql.v0), P2 = load_vertex(verts, ql.v1), P3 = load_vertex(verts, ql.v2);
        if (tri_contact_fast(P1, P2, P3, load_vertex(verts, lt.v0), load_vertex(verts, lt.v1), load_vertex(verts, lt.v2))) contact(ql.id, lt.id);
    };
    // SAT of one queued survivor
    auto run_sat = [&](const SatItem it) __attribute__((always_inline)) {
        const LeafTri lt = leaf[it.leaf];
        if (EXTERNAL) {
            const ExtQuery *q = reinterpret_cast<const ExtQuery *>(src.ext) + it.q;
            const d3 P1 = d3{q->v[0], q->v[1], q->v[2]}, P2 = d3{q->v[3], q->v[4], q->v[5]}, P3 = d3{q->v[6], q->v[7], q->v[8]};
            const uint32_t q_id = q->id;
            if (tri_contact_fast(P1, P2, P3, load_vertex(verts, lt.v0), load_vertex(verts, lt.v1), load_vertex(verts, lt.v2))) contact(q_id, lt.id);
        } else sat_leaves(leaf[it.q], lt);
    };

    // ---- the SAT-ready pairs first (the usual step has nothing else): chunks of 256 dealt round-robin over the shard's workgroups, every lane one SAT
    if (!EXTERNAL) {
        const unsigned long long nfchunks = (total_fat + EXACT_THREADS - 1) / EXACT_THREADS;
        for (unsigned long long cf = my_slot; cf < nfchunks; cf += slots) {
            const unsigned long long k = cf * EXACT_THREADS + tid;
            uint4 e0 = ff0, e1 = ff1;
            if (cf != my_slot) { const uint4 *fp = reinterpret_cast<const uint4 *>(fat_slot(my_cand, shard_cap, k < total_fat ? k : 0ull)); e0 = fp[0]; e1 = fp[1]; }   // (workgroup-uniform branch)
            if (k < total_fat) sat_leaves(LeafTri{e0.x, e0.y, e0.z, e0.w}, LeafTri{e1.x, e1.y, e1.z, e1.w});
        }
    }

    // Chunks of 256 candidates are dealt round-robin over the workgroups, EXACT_ITEMS chunks per workgroup and round: a few
    // thousand survivors (the usual case: the descent has filtered the rest) spread over as many workgroups as they fill, one
    // SAT batch each, instead of queueing four batches deep in a few workgroups; a million candidates still give every lane
    // EXACT_ITEMS independent loads.
    const unsigned long long nchunks = (total + EXACT_THREADS - 1) / EXACT_THREADS;                                 // of this shard
    for (unsigned long long c0 = my_slot; c0 < nchunks; c0 += (unsigned long long)slots * EXACT_ITEMS) {          // uniform trip count per workgroup
        // stage 1 on EXACT_ITEMS candidates per lane at once: their loads are independent and in flight together
        // (the stage is a chain of two dependent round trips per candidate -- latency, not bandwidth)
        Candidates c[EXACT_ITEMS]; bool ok[EXACT_ITEMS];
#pragma unroll
        for (int j = 0; j < EXACT_ITEMS; ++j) {
            const unsigned long long k = (c0 + (unsigned long long)j * slots) * EXACT_THREADS + tid;
            ok[j] = k < total;
            if (c0 == my_slot) c[j] = Candidates{cfq[j], cfl[j]};            // (the first round's were requested with the counts; workgroup-uniform branch)
            else c[j] = my_cand[ok[j] ? k : 0];
            if (!ok[j]) c[j] = Candidates{0, 0};
        }
        LeafTri lt[EXACT_ITEMS]; Box lb[EXACT_ITEMS], qbox[EXACT_ITEMS]; bool certain[EXACT_ITEMS], filtered[EXACT_ITEMS]; uint32_t q_id[EXACT_ITEMS], qa[EXACT_ITEMS], qb[EXACT_ITEMS], qc[EXACT_ITEMS];
#pragma unroll
        for (int j = 0; j < EXACT_ITEMS; ++j) {
            const uint32_t qi = c[j].q, lj = c[j].leaf & CAND_LEAF_MASK;  // (0, 0) for lanes past the end: harmless in-bounds loads
            certain[j] = (c[j].leaf & CAND_CERTAIN) != 0;
            filtered[j] = (c[j].leaf & CAND_FILTERED) != 0;                 // counted and filtered by the descent: nothing to fetch here
            c[j].leaf = lj;
            if (filtered[j]) { lt[j] = LeafTri{0, 0, 0, 0}; lb[j] = qbox[j] = Box{0, 0, 0, 0, 0, 0}; q_id[j] = qa[j] = qb[j] = qc[j] = 0; continue; }
            lt[j] = leaf[lj];
            lb[j] = qbox[j] = Box{0, 0, 0, 0, 0, 0};
            if (!certain[j]) lb[j] = leaf_box64(boxes, src.qbox, n, (int)lj);
            if (EXTERNAL) {
                const ExtQuery *q = reinterpret_cast<const ExtQuery *>(src.ext) + qi;
                q_id[j] = q->id; qa[j] = q->vidx[0]; qb[j] = q->vidx[1]; qc[j] = q->vidx[2];
                if (!certain[j]) qbox[j] = box_set(d3{q->v[0], q->v[1], q->v[2]}, d3{q->v[3], q->v[4], q->v[5]}, d3{q->v[6], q->v[7], q->v[8]});
            } else {
                const LeafTri ql = leaf[qi];
                q_id[j] = ql.id; qa[j] = ql.v0; qb[j] = ql.v1; qc[j] = ql.v2;
                if (!certain[j]) qbox[j] = leaf_box64(boxes, src.qbox, n, (int)qi);
            }
        }
#pragma unroll
        for (int j = 0; j < EXACT_ITEMS; ++j) {
            if (ok[j] && filtered[j]) { sq[atomicAdd(&sqcount, 1u)] = SatItem{c[j].q, c[j].leaf}; continue; }
            if (ok[j] && (certain[j] || box_overlap(qbox[j], lb[j]))) {        // collision.cuh:31-32, exact (certain: already decided exactly by the descent)
                tested += half ? 2u : 1u;                                      // (half traversal: the reference meets the pair from both sides)
                const bool survive = neighbor_count(qa[j], qb[j], qc[j], lt[j].v0 + vbase, lt[j].v1 + vbase, lt[j].v2 + vbase) < 1   // collision.cuh:38
                                     && (half ? q_id[j] != lt[j].id : q_id[j] < lt[j].id);   // tri_contact.cuh:81
                if (survive) sq[atomicAdd(&sqcount, 1u)] = SatItem{c[j].q, c[j].leaf};   // < 256 left over + <= 256*ITEMS new: fits EXACT_SQ
            }
        }
        __syncthreads();
        // The count is read ONCE, between two barriers with no push in between, so that every wave sees the same value
        // and the trip count below is workgroup-uniform whatever the skew between the waves (the loop holds barriers).
        uint32_t cnt = sqcount;
        while (cnt >= EXACT_THREADS) {                                         // full batch, every lane runs one SAT
            cnt -= EXACT_THREADS;
            const SatItem it = sq[cnt + tid];
            run_sat(it);
        }
        __syncthreads();                                                       // every wave has read the count and its items ...
        if (tid == 0) sqcount = cnt;                                           // ... before the count moves and the next round pushes
        __syncthreads();
    }
    __syncthreads();
    if (tid < sqcount) run_sat(sq[tid]);                                       // final partial batch (< 256)
    const unsigned long long t64 = wave_sum_u64(tested);
    if (lane == 0) wtested[tid >> 6] = (uint32_t)t64;
    __syncthreads();
    const uint32_t staged = pcount < EXACT_PB ? pcount : EXACT_PB;
    if (tid == 0) {
        if (staged) pbase = atomicAdd(&st->n_pairs, (unsigned long long)staged);
        uint32_t t = 0;
        for (int i = 0; i < EXACT_THREADS / 64; ++i) t += wtested[i];
        if (t) atomicAdd(&st->shard[blockIdx.x & (NSHARD - 1)].pairs_tested, (unsigned long long)t);
    }
    __syncthreads();
    for (uint32_t i = tid; i < staged; i += EXACT_THREADS) {
        const unsigned long long cur = pbase + i;
        if (cur < cap) reinterpret_cast<uint2 *>(pairs)[cur] = pbuf[i];
        if (post && cur < post_n) reinterpret_cast<uint2 *>(post)[cur] = pbuf[i];
    }
}

// ---------------------------------------------------------------- brute force (check.cuh:117-141)
// Tile of 256 "j" triangles staged in LDS per step; thread i tests its triangle against the tile.
__global__ __launch_bounds__(256) void k_brute_force(const double *__restrict__ verts, const uint32_t *__restrict__ vidx,
                                                     const uint32_t *__restrict__ ids, uint32_t n, int box_filter,
                                                     uint32_t *__restrict__ pairs, unsigned long long cap, TravState *__restrict__ st)
{
    __shared__ double sv[256][9];
    __shared__ uint32_t si[256][4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    uint32_t ia = 0, a0 = 0, a1 = 0, a2 = 0; d3 P1{}, P2{}, P3{}; Box bi{};
    if (live) {
        a0 = vidx[3 * (size_t)i]; a1 = vidx[3 * (size_t)i + 1]; a2 = vidx[3 * (size_t)i + 2];
        ia = ids ? ids[i] : i;
        P1 = load_vertex(verts, a0); P2 = load_vertex(verts, a1); P3 = load_vertex(verts, a2);
        bi = box_set(P1, P2, P3);
    }
    unsigned long long tested = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 256) {
        __syncthreads();
        const uint32_t j = j0 + threadIdx.x;
        if (j < n) {
            const uint32_t b0 = vidx[3 * (size_t)j], b1 = vidx[3 * (size_t)j + 1], b2 = vidx[3 * (size_t)j + 2];
            si[threadIdx.x][0] = ids ? ids[j] : j; si[threadIdx.x][1] = b0; si[threadIdx.x][2] = b1; si[threadIdx.x][3] = b2;
            const d3 Q1 = load_vertex(verts, b0), Q2 = load_vertex(verts, b1), Q3 = load_vertex(verts, b2);
            double *s = sv[threadIdx.x];
            s[0] = Q1.x; s[1] = Q1.y; s[2] = Q1.z; s[3] = Q2.x; s[4] = Q2.y; s[5] = Q2.z; s[6] = Q3.x; s[7] = Q3.y; s[8] = Q3.z;
        }
        __syncthreads();
        if (!live) continue;
        const uint32_t lim = min(256u, n - j0);
        for (uint32_t k = 0; k < lim; ++k) {
            const double *s = sv[k];
            const d3 Q1{s[0], s[1], s[2]}, Q2{s[3], s[4], s[5]}, Q3{s[6], s[7], s[8]};
            if (box_filter && !box_overlap(bi, box_set(Q1, Q2, Q3))) continue;
            ++tested;
            if (neighbor_count(a0, a1, a2, si[k][1], si[k][2], si[k][3]) < 1 && ia < si[k][0]) {
                if (tri_contact(P1, P2, P3, Q1, Q2, Q3)) {
                    const unsigned long long cur = atomicAdd(&st->n_pairs, 1ull);
                    if (cur < cap) { pairs[2 * cur] = ia; pairs[2 * cur + 1] = si[k][0]; }
                }
            }
        }
    }
    tested = wave_sum_u64(tested);
    if ((threadIdx.x & 63) == 0 && tested) atomicAdd(&st->shard[blockIdx.x & (NSHARD - 1)].pairs_tested, tested);
}

// tri_contact.cuh:80-87 over explicit index pairs, with the collision.cuh:38 neighbour gate.
__global__ __launch_bounds__(256) void k_test_pairs(const double *__restrict__ verts, const uint32_t *__restrict__ vidx,
                                                    const uint32_t *__restrict__ ids, const uint32_t *__restrict__ pr,
                                                    unsigned long long np, uint8_t *__restrict__ out)
{
    const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= np) return;
    const uint32_t a = pr[2 * k], b = pr[2 * k + 1];
    const uint32_t a0 = vidx[3 * (size_t)a], a1 = vidx[3 * (size_t)a + 1], a2 = vidx[3 * (size_t)a + 2];
    const uint32_t b0 = vidx[3 * (size_t)b], b1 = vidx[3 * (size_t)b + 1], b2 = vidx[3 * (size_t)b + 2];
    const uint32_t ia = ids ? ids[a] : a, ib = ids ? ids[b] : b;
    bool r = false;
    if (neighbor_count(a0, a1, a2, b0, b1, b2) < 1 && ia < ib)
        r = tri_contact(load_vertex(verts, a0), load_vertex(verts, a1), load_vertex(verts, a2),
                        load_vertex(verts, b0), load_vertex(verts, b1), load_vertex(verts, b2));
    out[k] = r ? 1 : 0;
}

// Leaves whose AABB strictly overlaps a peer's root box -> cd_query records (cross-rank pass), for ALL peers in one
// launch.  roots: n_boxes x 6 doubles in device memory (the all-gathered root AABBs, or one caller-supplied box);
// box p is skipped when p == skip or when it does not strictly overlap my_root (box.cuh:40-43) -- a peer whose root
// misses this rank's root cannot overlap any of its leaves.  Records for box p go to out + p * cap, their number to
// counts[p] (may exceed cap: nothing is written past it).  A returning atomic on one word retires at ~88 per
// microsecond on this chip, so hits are compacted with __ballot per wave and an LDS prefix over the 16 waves of the
// workgroup, and reserved with ONE global atomic per workgroup and box.
constexpr int PACK_THREADS = 1024;
__global__ __launch_bounds__(PACK_THREADS) void k_pack_queries(const double *__restrict__ verts, const LeafTri *__restrict__ leaf,
                                                              const double *__restrict__ boxes, const LeafBox32 *__restrict__ qbox32, int n,
                                                              const double *__restrict__ roots, int n_boxes, int skip, const double *__restrict__ my_root,
                                                              ExtQuery *__restrict__ out, unsigned long long cap, unsigned long long *__restrict__ counts,
                                                              uint32_t vbase)
{
    __shared__ uint32_t wcount[PACK_THREADS / 64];
    __shared__ unsigned long long wg_base;
    const int j = blockIdx.x * PACK_THREADS + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool live = j < n;
    // the leaf's fp32 box (rounded outward) rules most leaves out without touching the 48-byte FP64 box
    float lo0 = 0, lo1 = 0, lo2 = 0, hi0 = 0, hi1 = 0, hi2 = 0; bool exact32 = false;
    if (live) { const LeafBox32 lb = qbox32[j]; lo0 = lb.lo[0]; lo1 = lb.lo[1]; lo2 = lb.lo[2]; hi0 = lb.hi[0]; hi1 = lb.hi[1]; hi2 = lb.hi[2]; exact32 = (lb.flags & LB_EXACT) != 0; }
    Box mine{0, 0, 0, 0, 0, 0}; bool have_box = false;
    const Box me = my_root ? load_box(my_root, 0) : Box{0, 0, 0, 0, 0, 0};
    for (int p = 0; p < n_boxes; ++p) {                                       // (wave-uniform loop and skips)
        if (p == skip) continue;
        const Box rb = load_box(roots, p);
        if (my_root && !box_overlap(me, rb)) continue;
        // conservative: the exact strict overlap implies this one.  The peer's box is not in this mesh's cell table (cd_bvh.h), so
        // its lo against the leaf's hi' is '<=' (a tie may hide a bound of the peer below the leaf's); the leaf's lo' against the
        // peer's hi rounded up can stay strict
        bool hit = live && lo0 < __double2float_ru(rb.x2) && __double2float_rd(rb.x1) <= hi0 && lo1 < __double2float_ru(rb.y2) &&
                   __double2float_rd(rb.y1) <= hi1 && lo2 < __double2float_ru(rb.z2) && __double2float_rd(rb.z1) <= hi2;
        if (hit) {
            if (!have_box) {                                                  // (an exact-in-fp32 box is its query box: leaf_box64, cd_bvh.h)
                mine = exact32 ? Box{(double)lo0, (double)hi0, (double)lo1, (double)hi1, (double)lo2, (double)hi2} : load_box(boxes, (n - 1) + j);
                have_box = true;
            }
            hit = box_overlap(mine, rb);                                      // box.cuh:40-43, exact
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (lane == 0) wcount[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < PACK_THREADS / 64; ++k) { const uint32_t v = wcount[k]; before += (k < (int)w) ? v : 0u; total += v; }
        if (total != 0u) {                                                    // (workgroup-uniform)
            if (threadIdx.x == 0) wg_base = atomicAdd(&counts[p], (unsigned long long)total);
            __syncthreads();
            const unsigned long long k = wg_base + before + __popcll(m & ((1ull << lane) - 1ull));
            if (hit && k < cap) {
                const LeafTri lt = leaf[j];
                const d3 A = load_vertex(verts, lt.v0), B = load_vertex(verts, lt.v1), C = load_vertex(verts, lt.v2);
                ExtQuery q;
                q.v[0] = A.x; q.v[1] = A.y; q.v[2] = A.z; q.v[3] = B.x; q.v[4] = B.y; q.v[5] = B.z; q.v[6] = C.x; q.v[7] = C.y; q.v[8] = C.z;
                q.id = lt.id; q.vidx[0] = lt.v0 + vbase; q.vidx[1] = lt.v1 + vbase; q.vidx[2] = lt.v2 + vbase;
                out[(size_t)p * cap + k] = q;
            }
        }
        __syncthreads();                                                      // wcount / wg_base are reused by the next box
    }
}

// The same selection straight from the TRIANGLES, in their original order, before there is a tree: a query is a triangle
// (vertices, ID, global vertex indices) whatever leaf order its owner's tree will have, and which triangles overlap a
// peer's box does not depend on that order either.  The multi-GPU step packs with this kernel first thing, so that the
// counts and the records travel while the rank sorts and builds (cd_multi.h).  The box test is the exact one
// (box.cuh:13-22 + box.cuh:40-43) on the FP64 box of the triangle's vertices.
__global__ __launch_bounds__(PACK_THREADS) void k_pack_triangles(const double *__restrict__ verts, const uint32_t *__restrict__ vidx, const uint32_t *__restrict__ ids,
                                                                int n, const double *__restrict__ roots, int n_boxes, int skip, const double *__restrict__ my_root,
                                                                ExtQuery *__restrict__ out, unsigned long long cap, unsigned long long *__restrict__ counts,
                                                                uint32_t vbase)
{
    __shared__ uint32_t wcount[PACK_THREADS / 64];
    __shared__ unsigned long long wg_base;
    const int t = blockIdx.x * PACK_THREADS + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool live = t < n;
    uint32_t ia = 0, ib = 0, ic = 0;
    d3 A{0, 0, 0}, Bv{0, 0, 0}, Cv{0, 0, 0};
    Box mine{0, 0, 0, 0, 0, 0};
    if (live) {
        ia = vidx[3 * (size_t)t]; ib = vidx[3 * (size_t)t + 1]; ic = vidx[3 * (size_t)t + 2];
        A = load_vertex(verts, ia); Bv = load_vertex(verts, ib); Cv = load_vertex(verts, ic);
        mine = box_set(A, Bv, Cv);
    }
    const Box me = load_box(my_root, 0);
    for (int p = 0; p < n_boxes; ++p) {                                       // (wave-uniform loop and skips)
        if (p == skip) continue;
        const Box rb = load_box(roots, p);
        if (!box_overlap(me, rb)) continue;                                   // a peer whose box misses this rank's box overlaps none of its triangles
        const bool hit = live && box_overlap(mine, rb);                       // box.cuh:40-43, exact
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (lane == 0) wcount[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < PACK_THREADS / 64; ++k) { const uint32_t v = wcount[k]; before += (k < (int)w) ? v : 0u; total += v; }
        if (total != 0u) {                                                    // (workgroup-uniform)
            if (threadIdx.x == 0) wg_base = atomicAdd(&counts[p], (unsigned long long)total);
            __syncthreads();
            const unsigned long long k = wg_base + before + __popcll(m & ((1ull << lane) - 1ull));
            if (hit && k < cap) {
                ExtQuery q;
                q.v[0] = A.x; q.v[1] = A.y; q.v[2] = A.z; q.v[3] = Bv.x; q.v[4] = Bv.y; q.v[5] = Bv.z; q.v[6] = Cv.x; q.v[7] = Cv.y; q.v[8] = Cv.z;
                q.id = ids ? ids[t] : (uint32_t)t; q.vidx[0] = ia + vbase; q.vidx[1] = ib + vbase; q.vidx[2] = ic + vbase;
                out[(size_t)p * cap + k] = q;
            }
        }
        __syncthreads();                                                      // wcount / wg_base are reused by the next box
    }
}

}  // namespace cd
