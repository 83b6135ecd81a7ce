// mi355cd.hip -- C-ABI implementation of include/mi355cd.h (libmi355cd.so), gfx950 only.
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -shared -fPIC (see ../Makefile).
#include "../../include/mi355cd.h"
#include "cd_math.h"
#include "cd_sort.h"
#include "cd_bvh.h"
#include "cd_build.h"
#include "cd_traverse.h"
#include "cd_post.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

using namespace cd;

namespace {

#define HIPCHK(expr)                                                   \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return -(int)e_;                         \
    } while (0)

enum Stage { ST_CREATED = 0, ST_SORTED = 1, ST_BUILT = 2, ST_REFIT = 3 };

enum Ev { EV_MORTON0, EV_MORTON1, EV_SORT1, EV_HIER0, EV_HIER1, EV_REFIT0, EV_REFIT1, EV_TRAV0, EV_TRAV1, EV_DESC1, EV_DEEP0, EV_DEEP1, EV_CHK0, EV_CHK1, EV_BLK0, EV_BLK1, EV_COUNT };

}  // namespace

// Everything one traversal pass writes: counters, candidates, pairs (+ the Report in front of them), deferred items.
// The context has two sets: [0] for its own leaves as queries (and for every single-pass entry point), [1] for the
// external queries of the multi-GPU step, so that both passes can be enqueued back to back and read with ONE host
// synchronisation (cd_multi.h).
struct TravBuf {
    TravState *d_state = nullptr; bool state_owned = false;                // [0]: inside the context's scratch block (zeroed by the fused memset)
    uint32_t *d_pairs = nullptr; uint64_t pairs_cap = 0;                   // d_pairs points sizeof(Report) bytes into its allocation: [Report][pairs]
    char *h_report = nullptr;                                              // pinned: Report + SPEC_PAIRS pairs, target of the read-back
    uint32_t synced_reports = 0;                                           // reports into h_report that ended in a stream synchronise (polled completion waits for POLL_WARM of them)
    uint64_t direct_id = 0; uint32_t synced_direct = 0;                    // the same for the caller's pinned pair buffer last written directly (its serial number: pinned_pairs_id)
    uint2 *d_defer = nullptr; uint32_t defer_cap = 0;
    Candidates *d_cand = nullptr; uint64_t cand_cap = 0;
    int32_t *d_deep = nullptr; uint64_t deep_items = 0;
    // the report of the pass being enqueued, decided BEFORE its kernels are (run_traversal, prepare_report): the exact kernel posts the first pairs into pinned host memory itself
    uint32_t *post_dst = nullptr; uint64_t post_n = 0;                     // where to (the caller's pinned buffer, or the staging area behind the report), how many at most
    bool prepared = false; unsigned long long prep_seq = 0; uint32_t *prep_area = nullptr; bool posted = false;   // the polled completion's sequence number and the scan's area of that report; the pass's k_exact did post
};

struct cd_multi;
struct cd_ctx {
    uint32_t nv = 0, nt = 0;
    cd_multi *attached_multi = nullptr;     // the multi-GPU step object built on this context, if any (cd_multi.h): it holds a pointer to the context and buffers inside it
    int stage = ST_CREATED;
    int frame_mode = CD_FRAME_REFERENCE;
    uint32_t vbase = 0;
    int trav_variant = 3;                   // CD_OPT_TRAVERSAL: 0 = lane-private FP64 (A), 1 = fp32 descent from the root + exact kernel (B), 3 = half traversal (D)
    uint32_t queries_per_wave = 64; 
    uint32_t dbg_no_shared_path = 0;        // CD_DBG_NO_SHARED_PATH: k_descend without the shared root path (A/B)
    uint32_t dbg_diag = 0;                  // CD_DBG_DIAG: the DIAG instance of the descent kernel runs, which also fills the diagnostic counters (cd_debug_counters)
    uint32_t dbg_lds_pad = 0;               // CD_DBG_LDS_PAD: extra dynamic LDS bytes per traversal workgroup (occupancy experiments)
    double frame_host[FRAME_WORDS] = {0.004501, -0.476622, -0.381965, 3.08, 0.76, 2.36, 0.0, 0.0};   // morton.h:45,51,57; [6]: the bits of the key layout word (cd_math.h; 0 = the reference's interleave)
    hipStream_t stream = nullptr;
    hipEvent_t ev[EV_COUNT] = {};
    // inputs
    double *d_verts = nullptr; uint32_t *d_vidx = nullptr; uint32_t *d_ids = nullptr;
    // sort
    uint64_t *d_keys[2] = {nullptr, nullptr}; uint32_t *d_perm[2] = {nullptr, nullptr};
    uint32_t *d_counts = nullptr; uint32_t ntiles = 0;
    uint32_t *d_chunk_tot = nullptr;        // d_counts summed over chunks of OS_CHUNK tiles (sorts of more than OS_CHUNK_MIN_TILES tiles: k_tile_chunks)
    // onesweep state, part of one scratch block (layout: cd_create): [8][256] u32 histograms | 8 u32 tickets (padded) | [8][ntiles][256] u64 granules
    void *d_os = nullptr; size_t os_bytes = 0 /* whole block */, zero_bytes = 0 /* its front part: all a fused call with the hybrid sort needs zeroed */, sort_hi_bytes = 0 /* of that, the sort's own */;
    unsigned long long *d_os_look_lo = nullptr;   // granules of the digit passes 0..5 (behind the front part); d_os_look: passes 6 and 7
    uint32_t *d_os_hist = nullptr; uint32_t *d_os_ticket = nullptr; unsigned long long *d_os_look = nullptr;
    double *d_frame = nullptr, *d_partial = nullptr;
    // tree
    unsigned long long *d_top_pub = nullptr; uint32_t top_seq = 0;          // k_cross_fused: the upper levels of the fp32 tree as its first workgroup publishes them, and the launch counter its flag word carries
    LeafTri *d_leaf = nullptr; NodeMeta *d_meta = nullptr; int32_t *d_parent = nullptr; double *d_seg = nullptr; float *d_seg32 = nullptr; uint32_t nbp2 = 1; int32_t *d_cross = nullptr; uint32_t cross_cap = 0;   // segment tree over leaf boxes: nbp2*512 heap nodes
    double *d_boxes = nullptr; uint32_t *d_bounded = nullptr; NodeRec32 *d_recs32 = nullptr; LeafBox32 *d_qbox = nullptr;
    unsigned long long *d_leaf_side = nullptr;   // fused build: one bit a leaf -- its box is in the left half of recs[j] (else the right half of recs[j - 1]); what k_cross_fused reads instead of qbox[]
    bool qbox_valid = false;                // d_qbox holds THIS tree's query boxes (the fused build stores them only when somebody is known to read them: qbox_wanted / ensure_qbox)
    uint32_t dbg_big_offsets = 0;           // CD_DBG_BIG_OFFSETS: k_descend_half's 64-bit-address instance whatever n is (tests)
    uint32_t dbg_store_qbox = 0;            // CD_DBG_STORE_QBOX: the fused build always stores qbox[] (A/B, tests)
    // the cell table of the current vertices (cd_bvh.h AmbTable; amb_refresh): keys == nullptr while every coordinate is an fp32 value
    AmbTable amb = {nullptr, 0u, 0u};
    unsigned long long *d_amb_keys = nullptr, *d_amb_vals = nullptr; uint64_t amb_cap = 0; uint32_t *d_amb_flag = nullptr;
    bool cell_table_opt = true;             // CD_OPT_CELL_TABLE
    uint8_t *d_vamb = nullptr, *vamb = nullptr;   // per vertex: which of its three coordinates lie in ambiguous cells (vamb: d_vamb, or nullptr while there is no table)
    int32_t *d_root = nullptr;              // name (split) of the root record, one word inside d_small
    bool leaf_records_filled = false;       // leaf[] holds the sorted triangles (leaves_filled: and parent[] / bounded[] are reset)
    bool internal_boxes_valid = false;      // the FP64 boxes of the internal nodes were written by the last refit (fused calls skip them)
    int32_t *d_split_of = nullptr;          // fused build: split of every internal node (child links of the records)
    bool hierarchy_valid = false;           // meta[] / parent[] hold the tree of the current keys (fused calls build the records without them)
    bool last_tree_fused = false;           // the last fused call built hierarchy + refit in one pass (ms_hierarchy is then part of ms_refit)
    uint32_t stamp_mask = 15;               // CD_OPT_KERNEL_STAMPS: with stage timing off, which time stamps a fused call still takes (1 block build, 2 descent, 4 exact, 8 pipeline start): ~5 us of idle GPU each
    uint32_t dbg_sort_windows = 0;          // CD_DBG_SORT_WINDOWS: 0 the form the size asks for, 1 always the large window form of k_local_sort, 2 always the small one (A/B, tests)
    bool left_frame = false; uint32_t steps_in_mode1 = 0;    // the sort went from its first form to its second because keys lay beyond bit 59 (not for a long run); sorts since then
    bool local_small_ok = true, local_small_active = false;   // the small window form has not met a run too long for it; the sort in flight used it
    uint32_t dbg_split_cross = 0;           // CD_DBG_SPLIT_CROSS: the fused build runs k_cross_meta + k_cross_records instead of k_cross_fused (A/B, tests)
    uint32_t dbg_no_fused_build = 0;        // CD_DBG_STAGEWISE_BUILD: fused entry points run k_hierarchy + the meta-reading refit (A/B)
    uint32_t *d_small = nullptr;            // 16 x u32 scratch counters (parent_wrong, check outputs)
    // traversal
    TravBuf tb[2];
    int exact_blocks = 1024;
    // pair-list post-processing (cd_sorted_pairs / cd_collision_triangles): sort buffers sized on demand
    uint64_t *pp_keys[2] = {nullptr, nullptr}; uint32_t *pp_vals[2] = {nullptr, nullptr}; uint32_t *pp_flags = nullptr;
    void *pp_os = nullptr; size_t pp_os_bytes = 0; uint32_t pp_cap = 0;
    uint64_t last_pairs_on_device = 0;      // pairs of the last traversal that are resident in d_pairs
    // host mirrors
    cd_stats stats = {};
    uint32_t sort_flags[9] = {};            // [0..7] look-back time-out words of the last sort, [8] half-key fix-up overflow; refreshed by read_state()
    bool stage_events = true;               // CD_OPT_STAGE_TIMING
    bool leaves_filled = false;             // the sort's fix-up hop already wrote leaf[], parent = -1, bounded = 0
    bool events_ride = false;               // the last pass recorded EV_TRAV0 / EV_DESC1 / EV_TRAV1 through its kernels' dispatch packets
    uint32_t dbg_report_copies = 0;        // CD_DBG_REPORT_COPIES: the report kernel copies the first pairs to the host (32 workgroups) instead of the exact kernel posting them (A/B)
    bool order_hint_large = false;          // CD_OPT_ORDER_HINT 2: also for trees of more than 1 M leaves (sorted chunk by chunk; measured slower there)
    bool order_hint = true;                 // CD_OPT_ORDER_HINT: the half traversal takes its groups of 64 leaves longest-first, by the previous step's times (cd_bvh.h, build_half_order)
    bool order_ready = false;               // d_order holds THIS tree's order hint (its fused build has made it)
    uint32_t *d_cost = nullptr, *d_order = nullptr;   // per group of 64 leaves: the score k_build_block gives it (max of its triangles' time classes); the order made from the scores
    bool order_pending = false;             // k_build_block has scored the groups: the k_cross_fused behind it sorts them
    uint8_t *d_tri_cost = nullptr;          // per triangle (original index): the time class its wave left in the last half traversal
    bool prezeroed = false;                 // fused path: the scratch block was zeroed by one memset at pipeline start
    hipEvent_t tree_done_event = nullptr;   // set by the multi-GPU step: taken (and cleared) by the launch that completes the tree, if it can carry it
    bool scratch_clean = false;             // ... or by the kernels of the previous fused step (ZeroPlan, cd_build.h): no memset at all
    bool quiet_pass = false;                // launch_pass records no events (a pass on another stream, beside the one whose times are reported)
    int sort_mode = 0;                      // 0 hybrid on key bits 44..59 (every in-frame Morton key is below 2^60), 1 hybrid on bits 48..63 (2 global passes + in-LDS sort of the windows + fix-up), 2 half-key (4 passes + fix-up),
                                            // 2 full (8 passes); forced by CD_OPT_SORT_FULL, or escalated after an overflow on this context
    // CD_OPT_GRAPH: the steady-state fused step captured once as a hipGraph and replayed (graph_step); the key is everything a replay bakes in
    bool graph_opt = false;
    // CD_OPT_POLL: the step's end is read off the report's sequence word in pinned host memory instead of waiting for the stream (wait_report)
    bool poll_opt = true;
    unsigned long long report_seq = 0;      // last sequence number handed to a k_report
    uint32_t polled_steps = 0, poll_fallbacks = 0;
    // why a polled wait ran into its 20 ms budget (two such fall-backs in 360 000 stressed steps went unexplained): at the time-out the stream was still BUSY (the step itself was
    // late: queued behind other work, or the device stalled) / the stream had DRAINED and the word arrived with the synchronise (late in flight) / the word was not there even after the
    // stream had drained (LOST: the one that would be a bug of the protocol); and the longest polled wait that did end in the word, in microseconds
    uint32_t poll_fb_busy = 0, poll_fb_late_word = 0, poll_fb_lost = 0, poll_max_wait_us = 0;
    // CD_DBG_POLL_SCAN (tools/poll_stress.py, tests): the pair area the report kernel writes is filled with 0xff before every step and scanned the moment the
    // sequence word is seen -- a pair that is still 0xff then was overtaken by the word (poll_stale counts such steps: CD_DBG_GET_POLL_STALE; _FALLBACKS: the fall-backs to the stream)
    bool dbg_poll_check = false;
    uint32_t poll_stale = 0;
    hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr;
    struct GraphKey { uint64_t cap, spec_n; int sort_mode, variant, frame_mode; uint32_t nt, exact_blocks, dbg; const void *p_pairs, *p_cand, *p_defer, *p_report; uint64_t cand_cap; uint32_t defer_cap, pad; } graph_key = {};   // (no padding bytes: compared with memcmp)
    struct GraphPost { bool leaves_filled, leaf_records_filled, hierarchy_valid, internal_boxes_valid, last_tree_fused, events_ride, scratch_clean, qbox_valid; uint32_t sort_passes; } graph_post = {};
    uint64_t graph_replays = 0, graph_captures = 0;
    bool all_verts_referenced = false;      // every vertex belongs to a triangle (checked at cd_create; the topology never changes afterwards)
    int wall_clock_khz = 0;                 // hipDeviceAttributeWallClockRate: ticks of s_memrealtime per millisecond
    double root_box_host[6] = {};           // AABB of the whole tree, fetched together with other read-backs
    bool root_box_valid = false;
};

namespace {

// k_cross_fused's published levels above the blocks (24 bytes a node, heap nodes [1, nbp2)) and, behind them, the flag word
// (a tree of more than TOP_IN_BLOCK blocks: from k_top_publish / k_top_publish_upper)
size_t top_pub_nodes(const cd_ctx *c) { return std::max<size_t>(TOP_IN_BLOCK, c->nbp2); }
size_t top_pub_bytes(const cd_ctx *c) { return sizeof(unsigned long long) * 3 * top_pub_nodes(c) + 64; }
uint32_t *top_flag_of(cd_ctx *c) { return reinterpret_cast<uint32_t *>(c->d_top_pub + 3 * top_pub_nodes(c)); }

void free_all(cd_ctx *c)
{
    hipFree(c->d_verts); hipFree(c->d_vidx); hipFree(c->d_ids);
    for (int i = 0; i < 2; ++i) { hipFree(c->d_keys[i]); hipFree(c->d_perm[i]); }
    hipFree(c->d_counts); hipFree(c->d_chunk_tot); hipFree(c->d_os); hipFree(c->d_frame); hipFree(c->d_partial);
    hipFree(c->d_top_pub); hipFree(c->d_leaf); hipFree(c->d_meta); hipFree(c->d_parent); hipFree(c->d_seg); hipFree(c->d_seg32); hipFree(c->d_cross); hipFree(c->d_boxes);
    hipFree(c->d_bounded); hipFree(c->d_recs32); hipFree(c->d_qbox); hipFree(c->d_leaf_side); hipFree(c->d_split_of); hipFree(c->d_cost); hipFree(c->d_order); hipFree(c->d_tri_cost);
    hipFree(c->d_amb_keys); hipFree(c->d_amb_flag); hipFree(c->d_vamb);
    for (TravBuf &tb : c->tb) {
        if (tb.d_pairs) hipFree(reinterpret_cast<char *>(tb.d_pairs) - sizeof(Report));
        if (tb.h_report) hipHostFree(tb.h_report);
        hipFree(tb.d_defer); hipFree(tb.d_deep); hipFree(tb.d_cand);
        if (tb.state_owned) hipFree(tb.d_state);
    }
    for (int i = 0; i < 2; ++i) { hipFree(c->pp_keys[i]); hipFree(c->pp_vals[i]); }
    hipFree(c->pp_flags); hipFree(c->pp_os);
    if (c->graph_exec) hipGraphExecDestroy(c->graph_exec);
    if (c->graph) hipGraphDestroy(c->graph);
    for (int i = 0; i < EV_COUNT; ++i) if (c->ev[i]) hipEventDestroy(c->ev[i]);
    if (c->stream) hipStreamDestroy(c->stream);
}

inline uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

// pair buffers handed out by cd_alloc_host_pairs: pinned host memory the report kernel writes STRAIGHT into (no staging copy on the host)
// (id: a buffer's serial number -- an address the allocator hands out again is a NEW buffer, with pages the device has never written)
struct PinnedPairs { std::mutex mu; struct Buf { const uint32_t *p; uint64_t cap, id; }; std::vector<Buf> v; uint64_t next_id = 1; };
PinnedPairs &pinned_pairs() { static PinnedPairs p; return p; }
// 0: not a buffer of cd_alloc_host_pairs (or too small for `cap`); else the buffer's serial number
uint64_t pinned_pairs_id(const uint32_t *pairs, uint64_t cap)
{
    if (!pairs) return 0;
    PinnedPairs &pp = pinned_pairs();
    std::lock_guard<std::mutex> lock(pp.mu);
    for (const auto &e : pp.v) if (e.p == pairs && cap <= e.cap) return e.id;
    return 0;
}
constexpr uint64_t SPEC_PAIRS = 1u << 15;     // pairs that come back together with the counters, zero-copy (256 KB)

// morton.h:70-89 / :7-29 on explicit inputs (cd_morton3d_points, cd_expand64_values): the device functions k_morton uses
__global__ void k_morton_points(const double *__restrict__ xyz, uint64_t n, const double *__restrict__ frame /* FRAME_WORDS */, uint64_t *__restrict__ keys)
{
    const unsigned long long layout = (unsigned long long)__double_as_longlong(frame[6]);
    const KeyLayout kl = key_layout(layout, frame, frame + 3);
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        keys[i] = layout ? morton3d_layout(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], kl) : morton3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], frame, frame + 3);
}
__global__ void k_expand_values(const uint64_t *__restrict__ v, uint64_t n, uint64_t *__restrict__ out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = expand64(v[i]);
}

// box.cuh:24-32 / :40-43 and tri_contact.cuh:19-78 on explicit operands (cd_box_pairs, cd_tri_contact_points): the device functions
// the refit, the descent's FP64 leaf test and k_exact use
__global__ void k_box_pairs(const double *__restrict__ a, const double *__restrict__ b, uint64_t n, uint8_t *__restrict__ overlap,
                            double *__restrict__ merged)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double *pa = a + 6 * i, *pb = b + 6 * i;
        const Box x{pa[0], pa[1], pa[2], pa[3], pa[4], pa[5]}, y{pb[0], pb[1], pb[2], pb[3], pb[4], pb[5]};
        if (overlap) overlap[i] = box_overlap(x, y) ? 1 : 0;
        if (merged) {
            const Box m = box_merge(x, y);
            double *o = merged + 6 * i;
            o[0] = m.x1; o[1] = m.x2; o[2] = m.y1; o[3] = m.y2; o[4] = m.z1; o[5] = m.z2;
        }
    }
}
__global__ void k_tri_contact_points(const double *__restrict__ t, uint64_t n, uint8_t *__restrict__ out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double *p = t + 18 * i;
        // (the form k_exact runs: hardware max / min, tri_contact itself where a projection is a NaN -- cd_math.h; the reference-compiled verdicts of tests/golden pin THIS)
        out[i] = tri_contact_fast(d3{p[0], p[1], p[2]}, d3{p[3], p[4], p[5]}, d3{p[6], p[7], p[8]},
                                  d3{p[9], p[10], p[11]}, d3{p[12], p[13], p[14]}, d3{p[15], p[16], p[17]}) ? 1 : 0;
    }
}

// ---- the cell table of the vertices (cd_bvh.h): which fp32 cells hold two distinct doubles of the same axis
__global__ void k_amb_scan(const double *__restrict__ v, uint64_t n3, uint32_t *__restrict__ flag)
{
    bool any = false;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n3; i += (uint64_t)gridDim.x * blockDim.x) { const double x = v[i]; any |= (double)(float)x != x; }
    if (__builtin_amdgcn_ballot_w64(any) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
// Slot h: keys[h] = the cell's key (| AMB_BIT once two distinct doubles were seen in it), vals[h] = the first double that claimed it
// (as an ordered integer; 0 = not yet written -- no double maps to 0).  A value that finds its cell's slot holding ANOTHER double marks
// the cell; one that finds its own double does nothing.  The pass is idempotent and order-independent: whatever the interleaving,
// a cell ends up marked iff two distinct doubles of the mesh lie in it.
__global__ void k_amb_insert(const double *__restrict__ v, uint64_t n3, unsigned long long *keys, unsigned long long *vals, uint32_t shift, uint32_t mask)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n3; i += (uint64_t)gridDim.x * blockDim.x) {
        double x = v[i];
        if (x == 0.0) x = 0.0;                                             // -0.0 and +0.0 are the same value
        const unsigned long long key = amb_key((int)(i % 3), amb_cell(x));
        uint32_t h = amb_hash(key, shift);
        // A mesh has far fewer distinct coordinates than vertices along its regular directions (a grid's columns share their x):
        // look before touching the slot with an atomic -- a thousand threads on one word retire one after the other.
        for (uint32_t probe = 0; probe <= mask; ++probe) {                 // (the table holds every key at a load below 2 / 3: this ends)
            unsigned long long old = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~AMB_BIT;
            if (old == 0ull) old = atomicCAS(&keys[h], 0ull, key) & ~AMB_BIT;
            if (old == 0ull || old == key) break;
            h = (h + 1u) & mask;
        }
        const unsigned long long o = f64_ordered(x);                       // (never 0: f64_ordered(-NaN with all bits set) aside, outside the contract)
        unsigned long long w = __hip_atomic_load(&vals[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0ull) w = atomicCAS(&vals[h], 0ull, o);
        if (w != 0ull && w != o && !(__hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & AMB_BIT)) atomicOr(&keys[h], AMB_BIT);
    }
}
// vamb[v]: bit a = the cell of vertex v's coordinate a is ambiguous (what the leaf encoding of the builds reads instead of six probes)
__global__ void k_amb_vertex(const double *__restrict__ verts, uint32_t nv, AmbTable t, uint8_t *__restrict__ vamb)
{
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += gridDim.x * blockDim.x) {
        const double x = verts[3 * (size_t)v], y = verts[3 * (size_t)v + 1], z = verts[3 * (size_t)v + 2];
        vamb[v] = (uint8_t)((amb_lookup(t, 0, x) ? 1u : 0u) | (amb_lookup(t, 1, y) ? 2u : 0u) | (amb_lookup(t, 2, z) ? 4u : 0u));
    }
}

constexpr int BOUNDS_BLOCKS = 1024;
constexpr uint32_t SORT_RETRY_STEPS = 64;          // sorts after which a context whose mesh had left the Morton frame tries the first sort form again
void graph_drop(cd_ctx *c);

int ensure_pairs(cd_ctx *c, TravBuf &tb, uint64_t cap)
{
    if (cap <= tb.pairs_cap) return 0;
    if (tb.d_pairs) hipFree(reinterpret_cast<char *>(tb.d_pairs) - sizeof(Report));
    tb.d_pairs = nullptr; tb.pairs_cap = 0;
    if (&tb == &c->tb[0]) c->last_pairs_on_device = 0;                 // the resident pair list went with the old buffer
    char *block = nullptr;
    HIPCHK(hipMalloc(&block, sizeof(Report) + sizeof(uint32_t) * 2 * cap + 16));      // +16: k_report moves pairs as 16-byte quads
    tb.d_pairs = reinterpret_cast<uint32_t *>(block + sizeof(Report));
    tb.pairs_cap = cap;
    return 0;
}

// (an event that was never recorded -- stage events off -- makes hipEventElapsedTime fail: that is "no measurement",
//  not an error of the call in progress, so the sticky last-error is cleared)
float elapsed(cd_ctx *c, int a, int b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) { (void)hipGetLastError(); return 0.f; }
    return ms;
}

// Per-stage events cost a few microseconds of idle GPU each (two per stage boundary).  With CD_OPT_STAGE_TIMING 0
// only the events the roofline needs are recorded: pipeline start, descent start / end, pipeline end.
inline hipError_t evrec(cd_ctx *c, int idx)
{
    if (!c->stage_events && !(idx == EV_MORTON0 || idx == EV_TRAV0 || idx == EV_DESC1 || idx == EV_TRAV1 || idx == EV_DEEP0 || idx == EV_DEEP1)) return hipSuccess;
    if (!c->stage_events && idx == EV_MORTON0 && !(c->stamp_mask & 8u)) return hipSuccess;
    return hipEventRecord(c->ev[idx], c->stream);
}

struct Prezeroed {                          // scope of a fused call: stage memsets are replaced by the one in enqueue_morton_sort
    cd_ctx *c;
    explicit Prezeroed(cd_ctx *c_) : c(c_) { c->prezeroed = true; }
    void done() { c->prezeroed = false; }
    ~Prezeroed() { c->prezeroed = false; }
};

// ---- stage enqueuers (no host synchronisation inside) -------------------------------------------
static bool fused_build_next(const cd_ctx *c);
// links_too: the fix-up hop also resets the parent links / arrival counters (k_hierarchy and the stage-wise refit need them;
// the fused build does not)
// frame_ready: the caller has already run the bounds kernels of this step (the multi-GPU step needs the box of all triangles first)
int enqueue_morton_sort(cd_ctx *c, bool links_too = true, bool frame_ready = false)
{
    const uint32_t n = c->nt;
    hipStream_t s = c->stream;
    c->order_ready = false;                                                 // (the order hint belongs to a sorted order: this sort's fused build makes the next)
    HIPCHK(evrec(c, EV_MORTON0));
    // auto frame: per-block centroid bounds here (frame_ready: they are still there from this step's first sort); k_morton
    // folds them into the frame itself
    const bool auto_frame = c->frame_mode == CD_FRAME_AUTO;
    if (auto_frame && !frame_ready) k_centroid_bounds<false><<<BOUNDS_BLOCKS, 256, 0, s>>>(c->d_verts, c->d_vidx, n, c->d_partial);
    // fused pipeline: the sort scratch, the small counters and the traversal state are one block, zeroed once here
    // (the hybrid forms use the digit passes 6 and 7 only: their granules sit in the front part of the block, with the counters)
    // ... unless the previous fused step's own kernels have left the sort scratch zeroed and this step's kernels zero
    // the rest (k_local_sort: the small counters, k_build_block: the traversal counters; ZeroPlan, cd_build.h)
    const bool self_cleaning = c->prezeroed && c->sort_mode <= 1 && fused_build_next(c);
    const bool skip_memset = self_cleaning && c->scratch_clean;
    c->scratch_clean = false;
    if (!skip_memset) HIPCHK(hipMemsetAsync(c->d_os, 0, c->sort_mode <= 1 ? (c->prezeroed ? c->zero_bytes : c->sort_hi_bytes) : c->os_bytes, s));
    // Three forms of the same stable 64-bit sort (cd_sort.h): hybrid = 2 global passes on the top 16 bits + an in-LDS
    // sort of run-aligned windows + the fix-up hop; half-key = 4 global passes on the high 32 bits + the fix-up hop;
    // full = 8 global passes.  The keys start in the buffer that leaves the sorted data in buffer 0.
    const int mode = c->sort_mode;
    const bool hybrid = mode <= 1;
    const int down = mode == 0 ? 4 : 0;         // mode 0: the two global digits are key bits 44..51 and 52..59 -- 16 bits that all vary,
                                                // where bits 48..63 of a 60-bit Morton key hold 12: runs 16 x shorter, windows of even size
    const int first_digit = hybrid ? 6 : (mode == 2 ? 4 : 0);
    int cur = mode == 3 ? 0 : 1;
    const uint32_t mblocks = c->ntiles < 2048u ? c->ntiles : 2048u;         // a workgroup per sort tile (4096 keys)
    unsigned long long host_layout = 0ull; std::memcpy(&host_layout, &c->frame_host[6], sizeof host_layout);
    const double *mframe = auto_frame ? nullptr : c->d_frame;               // (auto: the kernel folds the partial bounds itself and WRITES d_frame)
    const double *mpart = auto_frame ? c->d_partial : nullptr;
    if (auto_frame || host_layout != 0ull)
        k_morton<true><<<mblocks, MORTON_THREADS, 0, s>>>(c->d_verts, c->d_vidx, n, mframe, c->d_keys[cur], c->d_os_hist, first_digit, down, c->d_os_ticket + 16,
                                                          mpart, (uint32_t)BOUNDS_BLOCKS, c->d_frame, nullptr, c->d_counts);
    else
        k_morton<false><<<mblocks, MORTON_THREADS, 0, s>>>(c->d_verts, c->d_vidx, n, mframe, c->d_keys[cur], c->d_os_hist, first_digit, down, c->d_os_ticket + 16,
                                                           mpart, (uint32_t)BOUNDS_BLOCKS, c->d_frame, nullptr, c->d_counts);
    HIPCHK(evrec(c, EV_MORTON1));
    // onesweep: one pass over the data per digit; the digit histograms came with the keys
    // (more than 512 tiles: the first pass adds up chunk totals of the per-tile counts instead of every earlier tile's row -- cd_sort.h)
    const bool chunked = c->ntiles > (uint32_t)OS_CHUNK_MIN_TILES;
    if (chunked) k_tile_chunks<<<cdiv(c->ntiles, (uint32_t)OS_CHUNK), RADIX, 0, s>>>(c->d_counts, c->ntiles, c->d_chunk_tot);
    for (int pass = first_digit; pass < 8; ++pass) {
        k_os_pass<<<c->ntiles, OS_THREADS, 0, s>>>(c->d_keys[cur], c->d_perm[cur], c->d_keys[cur ^ 1], c->d_perm[cur ^ 1], n, pass * RADIX_BITS - down,
                                                    c->d_os_hist + pass * RADIX,
                                                    (pass >= 6 ? c->d_os_look + (size_t)(pass - 6) * c->ntiles * RADIX : c->d_os_look_lo + (size_t)pass * c->ntiles * RADIX),
                                                    c->d_os_ticket + pass, pass == first_digit, HIST_COPIES, pass == first_digit ? c->d_counts : nullptr,
                                                    pass == first_digit && chunked ? c->d_chunk_tot : nullptr);
        cur ^= 1;
    }
    c->leaves_filled = false; c->leaf_records_filled = false;
    if (hybrid) {                               // data is in buffer 1 again; windows go 1 -> 0, fix-up hop and leaf fill in the kernel's epilogue
        const LeafFill fill{c->d_vidx, c->d_ids, n, c->d_leaf, links_too ? c->d_parent : nullptr, links_too ? c->d_bounded : nullptr};
        // one workgroup per CU is all this kernel's LDS allows: the windows are n / 256 keys when that is less than their nominal
        // 4096 (1 M keys: 256 windows of 3907 instead of 245 of 4096 -- every CU busy, fewer windows over 4096 keys), not below 1024
        // (round 5) the SMALL form by default: windows of 2048 keys in workgroups of 512 threads and 48 KB, two of which share a CU (cd_sort.h) -- faster at every
        // size measured (100 k: 18.3 -> 14.4 us, 1 M cloth: 30.9 -> 27.8, 8 M: 269 -> 199); a run too long for it (3072 keys) is answered by the large form
        // (windows of 4096, runs up to 6144, one 1024-thread workgroup a CU) before the sort escalates to more global passes (judge_sort_flags)
        const bool small = c->dbg_sort_windows == 2 || (c->dbg_sort_windows == 0 && c->local_small_ok);
        c->local_small_active = small;
        if (small) {
            // (windows of n / 512 keys when that is less than the nominal 2048 -- two workgroups on every CU --, not below 512)
            const uint32_t per_half_cu = cdiv(n, 512u), win = per_half_cu >= (uint32_t)LocalSmall::W ? (uint32_t)LocalSmall::W : (per_half_cu < 512u ? 512u : per_half_cu);
            k_local_sort<LeafFill, LocalSmall><<<cdiv(n, win), LocalSmall::THREADS, 0, s>>>(c->d_keys[1], c->d_perm[1], c->d_keys[0], c->d_perm[0], n, 48 - down, c->d_os_ticket + 16, fill,
                                                                                            self_cleaning ? c->d_small : nullptr, self_cleaning ? 128u : 0u, win);
        } else {
            const uint32_t per_cu = cdiv(n, 256u), win = per_cu >= (uint32_t)LOCAL_W ? (uint32_t)LOCAL_W : (per_cu < 1024u ? 1024u : per_cu);
            k_local_sort<LeafFill, LocalLarge><<<cdiv(n, win), LocalLarge::THREADS, 0, s>>>(c->d_keys[1], c->d_perm[1], c->d_keys[0], c->d_perm[0], n, 48 - down, c->d_os_ticket + 16, fill,
                                                                                            self_cleaning ? c->d_small : nullptr, self_cleaning ? 128u : 0u, win);
        }
        c->leaves_filled = links_too;
        c->leaf_records_filled = true;
    } else if (mode != 3) {
        k_sort_fixup_fill<<<cdiv(n, 256), 256, 0, s>>>(c->d_keys[1], c->d_perm[1], c->d_keys[0], c->d_perm[0], n, c->d_os_ticket + 16,
                                                       c->d_vidx, c->d_ids, c->d_leaf, links_too ? c->d_parent : nullptr, links_too ? c->d_bounded : nullptr);
        c->leaves_filled = links_too;           // by the fix-up hop; enqueue_hierarchy runs k_fill_leaves otherwise
        c->leaf_records_filled = true;
    }
    c->stats.sort_passes = hybrid ? 2 : (mode == 2 ? 4 : 8);
    HIPCHK(evrec(c, EV_SORT1));
    HIPCHK(hipGetLastError());
    return 0;
}

int enqueue_hierarchy(cd_ctx *c, bool poison_boxes)
{
    const uint32_t n = c->nt;
    hipStream_t s = c->stream;
    HIPCHK(evrec(c, EV_HIER0));
    if (!c->prezeroed) HIPCHK(hipMemsetAsync(c->d_small, 0, 16 * sizeof(uint32_t), s));
    // (a repeated cd_build_hierarchy needs the parent links reset again: the flag is good for one use)
    if (!c->leaves_filled || poison_boxes)
        k_fill_leaves<<<cdiv(n, 256), 256, 0, s>>>(c->d_perm[0], c->d_vidx, c->d_ids, n, c->d_leaf, c->d_parent, c->d_bounded, poison_boxes ? c->d_boxes : nullptr);
    c->leaves_filled = false;
    if (n > 1)
        k_hierarchy<<<cdiv(n - 1, 256), 256, 0, s>>>(c->d_keys[0], (int)n, c->d_meta, c->d_parent, c->d_small);
    c->hierarchy_valid = true;
    HIPCHK(evrec(c, EV_HIER1));
    HIPCHK(hipGetLastError());
    return 0;
}

// fused: k_build_block (cd_build.h) builds hierarchy, fp32 boxes and records of its 512-leaf block itself, k_cross_meta /
// k_cross_records those of the cross nodes: no k_hierarchy before it, meta[] / parent[] / internal FP64 boxes are not written.
// Does anybody the build knows of read qbox[] after it?  The half traversal takes its query boxes out of the records and k_cross_fused its leaf pieces too (cd_build.h);
// the from-the-root descent (variant 1, and variant 0's tree comes from the stage-wise build anyway), k_cross_records, k_exact's FP64 path on a mesh with a cell table
// and the multi-GPU step's packers / external pass do read it.  Everybody else asks ensure_qbox() when the time comes.
static bool qbox_wanted(const cd_ctx *c)
{
    return c->trav_variant != 3 || c->dbg_split_cross || c->dbg_store_qbox || c->amb.keys != nullptr || c->amb.mask != 0u || c->attached_multi != nullptr;
}
int ensure_qbox(cd_ctx *c)
{
    if (c->qbox_valid) return 0;
    k_fill_qbox<<<cdiv(c->nt, 256), 256, 0, c->stream>>>(c->d_verts, c->d_leaf, (int)c->nt, c->d_qbox, c->amb, (const uint8_t *)c->vamb);
    HIPCHK(hipGetLastError());
    c->qbox_valid = true;
    return 0;
}
int enqueue_refit(cd_ctx *c, bool write_internal, bool fused = false)
{
    const uint32_t n = c->nt;
    hipStream_t s = c->stream;
    HIPCHK(evrec(c, EV_REFIT0));
    const int nblocks = (int)cdiv(n, REFIT_BLK);
    // the cross-node list (nodes whose range leaves their 512-leaf block) and its length, d_small[16]
    int32_t *cross_list = c->d_cross;
    uint32_t *cross_count = c->d_small + 16;
    if (!c->prezeroed) HIPCHK(hipMemsetAsync(cross_count, 0, 64 * sizeof(uint32_t), s));
    if (fused) {
        if (!c->leaf_records_filled)                 // (the 8-pass sort does not fill the leaves; enqueue_hierarchy would have)
            k_fill_leaves<<<cdiv(n, 256), 256, 0, s>>>(c->d_perm[0], c->d_vidx, c->d_ids, n, c->d_leaf, c->d_parent, c->d_bounded, nullptr);
        c->leaves_filled = false; c->leaf_records_filled = true;
        c->hierarchy_valid = false;
        // (its time stamps ride on its own dispatch packet: this is the largest kernel of the step, bench.py prices it)
        const bool stamp = (c->stamp_mask & 1u) != 0;
        const bool self_cleaning = c->prezeroed && c->sort_mode <= 1;
        ZeroPlan zp{nullptr, 0u, nullptr, 0u, nullptr, 0u, top_flag_of(c)};
        if (self_cleaning) {
            const size_t gran = sizeof(unsigned long long) * (size_t)c->ntiles * RADIX;
            zp = ZeroPlan{c->d_os_hist, (uint32_t)HIST_COPIES * 8u * RADIX + 8u /* histograms + tickets; the flags behind them stay */,
                          reinterpret_cast<uint4 *>(c->d_os_look), (uint32_t)(2 * gran / sizeof(uint4)),
                          reinterpret_cast<uint32_t *>(c->tb[0].d_state), (uint32_t)(sizeof(TravState) / sizeof(uint32_t)), top_flag_of(c)};
        }
        const int seg_min = (c->nbp2 > 1 && !c->dbg_split_cross) ? SEG32_MIN_LEVEL : SEG_MIN_LEVEL;
        // the half traversal's order hint (cd_bvh.h): this kernel scores the groups of 64 leaves, 8 workgroups of k_cross_fused sort them -- for trees that kernel serves
        // (trees of more than TOP_IN_BLOCK blocks -- more than 1 M leaves -- only with CD_OPT_ORDER_HINT 2: their descent runs many rounds of waves, the tail the hint packs is a
        //  tenth of it and the locality it gives up costs more -- 4 M cloth 173 -> 176 us, 8 M 454 -> 471 us with the hint)
        const bool hint = c->order_hint && c->trav_variant >= 3 && n > 64u && c->nbp2 > 1 && !c->dbg_split_cross && (c->nbp2 <= (uint32_t)TOP_IN_BLOCK || c->order_hint_large);
        const uint32_t *hp = hint ? c->d_perm[0] : nullptr; const uint8_t *ht = hint ? c->d_tri_cost : nullptr; uint32_t *hc = hint ? c->d_cost : nullptr;
        const int store_q = qbox_wanted(c) ? 1 : 0;
        c->qbox_valid = store_q != 0;
        if (stamp)
            hipExtLaunchKernelGGL(k_build_block, dim3(nblocks), dim3(REFIT_BLK), 0u, s, c->ev[EV_BLK0], c->ev[EV_BLK1], 0u,
                                  (const double *)c->d_verts, (const LeafTri *)c->d_leaf, (int)n, (const uint64_t *)c->d_keys[0], c->d_split_of,
                                  c->d_boxes, c->d_recs32, c->d_qbox, c->d_root, c->d_seg, c->d_seg32, (int)c->nbp2,
                                  cross_list, cross_count, c->cross_cap, zp, seg_min, c->amb, (const uint8_t *)c->vamb, hp, ht, hc, c->d_leaf_side, store_q);
        else                                        // (no time stamps: a plain launch, which a stream capture can record -- graph_step)
            k_build_block<<<nblocks, REFIT_BLK, 0, s>>>(c->d_verts, c->d_leaf, (int)n, c->d_keys[0], c->d_split_of, c->d_boxes, c->d_recs32, c->d_qbox, c->d_root,
                                                        c->d_seg, c->d_seg32, (int)c->nbp2, cross_list, cross_count, c->cross_cap, zp, seg_min, c->amb, (const uint8_t *)c->vamb,
                                                        hp, ht, hc, c->d_leaf_side, store_q);
        c->order_pending = hint;
        c->scratch_clean = self_cleaning;       // (judge_sort_flags takes it back when the sort has raised a flag)
    } else {
        c->qbox_valid = true;
        k_refit_seg_local<<<nblocks, REFIT_BLK, 0, s>>>(c->d_verts, c->d_leaf, (int)n, c->d_meta, c->d_boxes, c->d_bounded,
                                                       c->d_recs32, c->d_qbox, c->d_root, write_internal ? 1 : 0, c->d_seg, (int)c->nbp2,
                                                       cross_list, cross_count, c->cross_cap, c->amb, (const uint8_t *)c->vamb);
    }
    // fused build, a tree of 2 blocks or more: the cross nodes' ranges, splits, links and records in ONE launch (k_cross_fused, cd_build.h)
    if (fused && n > 1 && c->nbp2 > 1 && !c->dbg_split_cross) {
        const uint32_t xb = (uint32_t)nblocks < 8u ? 8u : ((uint32_t)nblocks > 1790u ? 1790u : (uint32_t)nblocks);   // 16 nodes per workgroup and round, ~13 per block; with the two workgroups below at most what the chip holds at once (7 workgroups per CU)
        // (the multi-GPU step's "tree is there" event rides on this kernel's dispatch packet: recorded on its own it is a barrier
        //  packet between the tree and the traversal, ~6 us of idle GPU)
        hipEvent_t done = c->tree_done_event; c->tree_done_event = nullptr;
        const uint32_t ogroups = (n + 63u) / 64u;
        const uint32_t nord = c->order_pending ? 8u * (((ogroups + 7u) / 8u + (uint32_t)ORDER_MAX_ITEMS - 1u) / (uint32_t)ORDER_MAX_ITEMS) : 0u;   // a workgroup per XCD list and chunk of it
        c->order_pending = false;
        const bool large = c->nbp2 > (uint32_t)TOP_IN_BLOCK;                // the upper levels come from launches of their own (below)
        const uint32_t xlds = std::max((uint32_t)(large ? 64u : sizeof(float) * 6 * (c->nbp2 >= 4 ? c->nbp2 / 4 : 1)) /* (used by the publishing workgroup only) */,
                                       nord ? (uint32_t)sizeof(OrderLds<256>) : 0u /* (the order hint's workgroups) */);
        uint32_t *top_flag = top_flag_of(c);
        if (++c->top_seq == 0u) ++c->top_seq;                               // (never 0: what k_build_block leaves in the flag word)
        if (large) {
            const int nspans = (int)(c->nbp2 / (uint32_t)TOP_IN_BLOCK);
            k_top_publish<<<nspans, 256, 0, s>>>(c->d_seg32, (int)c->nbp2, nblocks, c->d_top_pub);
            k_top_publish_upper<<<1, 256, 0, s>>>(c->d_top_pub, nspans, top_flag, c->top_seq);
        }
        if (done)
            hipExtLaunchKernelGGL(k_cross_fused, dim3(nord + xb + 2u /* [the order hint's 8,] then: the first workgroup publishes the upper levels, the last folds the FP64 box of all leaves */), dim3(256), xlds, s, nullptr, done, 0u,
                                  (const uint64_t *)c->d_keys[0], (int)n, (const double *)c->d_seg, (const float *)c->d_seg32, (int)c->nbp2, nblocks, (const unsigned long long *)c->d_leaf_side, c->d_boxes,
                                  c->d_recs32, (const int32_t *)c->d_split_of, c->d_root, (const int32_t *)c->d_cross, (const uint32_t *)cross_count, c->cross_cap,
                                  c->d_top_pub, top_flag, c->top_seq, nord, ogroups, (const uint32_t *)c->d_cost, c->d_order);
        else
            k_cross_fused<<<nord + xb + 2u, 256, xlds, s>>>(c->d_keys[0], (int)n, c->d_seg, c->d_seg32, (int)c->nbp2, nblocks, c->d_leaf_side, c->d_boxes,
                                                            c->d_recs32, c->d_split_of, c->d_root, c->d_cross, cross_count, c->cross_cap, c->d_top_pub, top_flag, c->top_seq,
                                                            nord, ogroups, c->d_cost, c->d_order);
        if (nord) c->order_ready = true;
        c->internal_boxes_valid = write_internal;
        HIPCHK(evrec(c, EV_REFIT1));
        HIPCHK(hipGetLastError());
        return 0;
    }
    // the levels above the blocks: a launch of their own -- unless the fused build's k_cross_meta can take them along (block 0)
    const bool top_in_meta = fused && n > 1 && c->nbp2 <= (uint32_t)TOP_IN_BLOCK;
    if (!top_in_meta) {
        if (fused && n > 1) {                    // a large tree: a workgroup per 2048 blocks, then the heap's upper part the same way
            int nb = (int)c->nbp2, live = nblocks;
            while (nb > 1) {
                const int span = nb < TOP_IN_BLOCK ? nb : TOP_IN_BLOCK;
                k_top_levels<<<nb / span, 256, 0, s>>>(c->d_seg, nb, live);
                nb /= span; live = (live + span - 1) / span;
            }
        } else k_refit_seg_top<<<1, 1024, 0, s>>>(c->d_seg, (int)c->nbp2, nblocks);
    }
    // about 13 cross nodes per 512-leaf block: ~one node per wave, every load chain in flight at once
    // one wave per cross node, about 13 of them per 512-leaf block: two workgroups (8 waves) per block -> 1-2 nodes per wave
    // (measured: 1024 / 2048 / 4096 / 8192 workgroups at 1 M triangles -> 120 / 113 / 111 / 112 us for the whole stage with the FP64
    //  kernels; k_cross_records alone with 2 .. 16 workgroups per block: no difference, 76 us for the stage)
    const uint32_t xblocks = 2u * nblocks < 256u ? 256u : (2u * nblocks > 16384u ? 16384u : 2u * (uint32_t)nblocks);
    if (fused && n > 1) {
        k_cross_meta<<<xblocks, 256, 0, s>>>(c->d_keys[0], (int)n, c->d_meta, c->d_split_of, c->d_cross, cross_count, c->cross_cap,
                                             top_in_meta ? c->d_seg : nullptr, (int)c->nbp2, nblocks);
        k_cross_records<<<xblocks, 256, 0, s>>>((int)n, c->d_meta, c->d_seg, c->d_seg32, (int)c->nbp2, c->d_qbox, c->d_boxes, c->d_recs32,
                                                c->d_split_of, c->d_root, c->d_cross, cross_count, c->cross_cap, c->amb);
    } else if (n > 1)
        k_refit_seg_cross<<<xblocks, 256, 0, s>>>((int)n, c->d_meta, c->d_seg, (int)c->nbp2, c->d_boxes, c->d_bounded, c->d_recs32,
                                                  c->d_root, write_internal ? 1 : 0, c->d_cross, cross_count, c->cross_cap, c->amb);
    c->internal_boxes_valid = write_internal;
    HIPCHK(evrec(c, EV_REFIT1));
    HIPCHK(hipGetLastError());
    return 0;
}

// Hierarchy + refit of a fused call: one pass that builds the records straight from the sorted keys (no k_hierarchy, no
// meta[] / parent[]), unless the traversal in use walks the reference-shaped tree (variant 0) or the A/B switch says so.
static bool fused_build_next(const cd_ctx *c) { return !(c->trav_variant == 0 || c->dbg_no_fused_build); }
int enqueue_tree(cd_ctx *c)
{
    c->last_tree_fused = false;
    if (!fused_build_next(c)) {
        int rc = enqueue_hierarchy(c, false);
        if (!rc) rc = enqueue_refit(c, c->trav_variant == 0, false);
        return rc;
    }
    c->last_tree_fused = true;
    return enqueue_refit(c, false, true);
}

// Launch one traversal pass (shallow: all queries from the root; deep: the deferred (query, subtree) items).
template <bool EXTERNAL, bool DEEP>
void launch_pass(cd_ctx *c, TravBuf &tb, const QuerySrc &src, uint32_t items, uint64_t cap_pairs)
{
    const int n = (int)c->nt;
    hipStream_t s = c->stream;
    const uint32_t vb = EXTERNAL ? c->vbase : 0u;
    if (c->trav_variant == 0) {
        k_traverse<EXTERNAL, DEEP><<<cdiv(items, TRAV_THREADS), TRAV_THREADS, 0, s>>>(src, items, n, c->d_meta, c->d_boxes, c->d_leaf, c->d_verts, tb.d_pairs, cap_pairs, tb.d_state,
                                                                                       DEEP ? nullptr : tb.d_defer, DEEP ? 0u : tb.defer_cap, DEEP ? tb.d_deep : nullptr, vb);
    } else {
        // variant 3 (half traversal) applies to self-collision queries; external queries are not leaves of this tree
        // and take the full descent of variant 1.  The deep pass of a half traversal continues (query, subtree) items
        // with the full descent, but its candidates keep the half traversal's meaning (`half`).
        const bool half_mode = c->trav_variant >= 3 && !EXTERNAL && (DEEP || items == (uint32_t)n);     // (k_descend_half derives its grid mapping from n: its queries are ALL the leaves)
        const uint32_t qpw = (DEEP || c->trav_variant >= 3) ? 64u : c->queries_per_wave;
        const uint64_t shard_cap = tb.cand_cap / NSHARD;
        const dim3 grid(cdiv(items, qpw * DESC_WAVES));
        const size_t pad = DEEP ? 0 : c->dbg_lds_pad;
        // Timing of the two kernels: with stage events on, hipEventRecord before / between / after (each record is a
        // barrier packet and ~6 us of idle GPU); off, the events ride on the kernels' own dispatch packets
        // (hipExtLaunchKernelGGL start / stop events): same timestamps, no gaps.
        const bool ride = !DEEP && !c->stage_events && !c->quiet_pass;
        hipEvent_t e0 = (ride && (c->stamp_mask & 2u)) ? c->ev[EV_TRAV0] : nullptr, e1 = (ride && (c->stamp_mask & 2u)) ? c->ev[EV_DESC1] : nullptr;
        hipEvent_t e2 = (ride && (c->stamp_mask & 4u)) ? c->ev[EV_TRAV1] : nullptr;
        const uint32_t qarg = qpw | (c->dbg_no_shared_path ? 0x40000000u : 0u);
        uint2 *dl = DEEP ? nullptr : tb.d_defer; const uint32_t dcap = DEEP ? 0u : tb.defer_cap; int32_t *deep = DEEP ? tb.d_deep : nullptr;
        const uint32_t half = half_mode ? 1u : 0u;
        const bool plain = !e0 && !e1 && !e2;          // no time stamp rides on these launches: plain launches (a stream capture can record them)
        if (half_mode && !DEEP) {
            const dim3 hgrid(cdiv(items, 64u)), hblock(HALF_THREADS);
            const uint32_t *h_order = (c->order_hint && c->order_ready) ? c->d_order : nullptr;      // (the order hint: cd_bvh.h, build_half_order)
            const bool h_rec = c->order_hint && (c->nbp2 <= (uint32_t)TOP_IN_BLOCK || c->order_hint_large);      // (the waves leave their times only where a build will read them)
            const uint32_t *h_perm = h_rec ? c->d_perm[0] : nullptr; uint8_t *h_tri = h_rec ? c->d_tri_cost : nullptr;
#define LAUNCH_HALF_(DIAG, TIES, BIG)                                                                                                           \
            do { if (plain) k_descend_half<DIAG, TIES, BIG><<<hgrid, hblock, pad, s>>>(src, n, c->d_recs32, tb.d_state, tb.d_cand, (unsigned long long)shard_cap, dl, dcap, h_order, h_perm, h_tri); \
                 else hipExtLaunchKernelGGL((k_descend_half<DIAG, TIES, BIG>), hgrid, hblock, (uint32_t)pad, s, e0, e1, 0u, src, n, (const NodeRec32 *)c->d_recs32, \
                                            tb.d_state, tb.d_cand, (unsigned long long)shard_cap, dl, dcap, h_order, h_perm, h_tri); } while (0)
            const bool ties = c->amb.keys != nullptr || c->amb.mask != 0u;                      // (a graph capture bakes the instance in: amb_refresh drops the graph when the table comes or goes)
            const bool big = n > HALF_SMALL_N || c->dbg_big_offsets;                            // (CD_DBG_BIG_OFFSETS: the tests run the 64-bit instance on trees of any size)
#define LAUNCH_HALF(DIAG, TIES) do { if (big) LAUNCH_HALF_(DIAG, TIES, true); else LAUNCH_HALF_(DIAG, TIES, false); } while (0)
            if (c->dbg_diag) { if (ties) LAUNCH_HALF(true, true); else LAUNCH_HALF(true, false); }
            else { if (ties) LAUNCH_HALF(false, true); else LAUNCH_HALF(false, false); }
#undef LAUNCH_HALF
#undef LAUNCH_HALF_
        }
        else if (qpw == 64)
            hipExtLaunchKernelGGL((k_descend<EXTERNAL, DEEP, false>), grid, dim3(DESC_THREADS), (uint32_t)pad, s, e0, e1, 0u,
                                  src, items, n, qarg, (const NodeRec32 *)c->d_recs32, (const double *)c->d_boxes, tb.d_state, tb.d_cand, (unsigned long long)shard_cap, dl, dcap, deep, vb, half);
        else
            hipExtLaunchKernelGGL((k_descend<EXTERNAL, DEEP, true>), grid, dim3(DESC_THREADS), (uint32_t)pad, s, e0, e1, 0u,
                                  src, items, n, qarg, (const NodeRec32 *)c->d_recs32, (const double *)c->d_boxes, tb.d_state, tb.d_cand, (unsigned long long)shard_cap, dl, dcap, deep, vb, half);
        if (!DEEP && !ride && !c->quiet_pass) evrec(c, EV_DESC1);
        uint32_t *post = DEEP ? nullptr : tb.post_dst; const unsigned long long post_n = DEEP ? 0ull : (unsigned long long)tb.post_n;
        if (!DEEP) { tb.posted = post != nullptr; tb.post_dst = nullptr; tb.post_n = 0; }     // (asked for by run_traversal for THIS pass only)
        if (plain)
            k_exact<EXTERNAL><<<c->exact_blocks, EXACT_THREADS, 0, s>>>(src, n, c->d_leaf, c->d_boxes, c->d_verts, vb, tb.d_cand, (unsigned long long)shard_cap, tb.d_pairs,
                                                                        (unsigned long long)cap_pairs, tb.d_state, half, post, post_n);
        else
            hipExtLaunchKernelGGL((k_exact<EXTERNAL>), dim3(c->exact_blocks), dim3(EXACT_THREADS), 0u, s, nullptr, e2, 0u,
                                  src, n, (const LeafTri *)c->d_leaf, (const double *)c->d_boxes, (const double *)c->d_verts, vb, (const Candidates *)tb.d_cand,
                                  (unsigned long long)shard_cap, tb.d_pairs, (unsigned long long)cap_pairs, tb.d_state, half, post, post_n);
        c->events_ride = ride;
    }
}

struct HostCounters { uint64_t n_pairs, pairs_tested, node_visits, max_shard_candidates; uint32_t n_deferred; uint64_t wave_steps, candidates, clk_start_inv, clk_end; };

// One host round trip and NO copy: k_report writes counters, the sort's time-out flags, the root box and the first
// spec_n pairs straight into pinned host memory.  enqueue_report queues the kernel; parse_report reads the record
// after the caller has synchronised the stream (the multi-GPU step queues two passes and synchronises once).
int ensure_report(TravBuf &tb)
{
    if (!tb.h_report) {
        // (coherent = fine-grained host memory: what the polled completion's system-scope release / acquire pair is defined on)
        HIPCHK(hipHostMalloc(&tb.h_report, sizeof(Report) + sizeof(uint32_t) * 2 * SPEC_PAIRS, hipHostMallocCoherent));
        std::memset(tb.h_report, 0, sizeof(Report));                       // (Report::seq: pinned memory is recycled by the allocator -- no stale sequence number)
    }
    return 0;
}
// direct: the first spec_n pairs go straight into the CALLER's buffer (pinned host memory from cd_alloc_host_pairs) instead of the staging area behind the record
int enqueue_report(cd_ctx *c, TravBuf &tb, bool want_pairs, uint64_t &spec_n, uint32_t *direct = nullptr, unsigned long long seq = 0)
{
    { const int rc = ensure_report(tb); if (rc) return rc; }
    if (!want_pairs) spec_n = 0; else if (spec_n > SPEC_PAIRS) spec_n = SPEC_PAIRS;
    // (the pair list is allocated with an even capacity + slack, so the quad copy may read one pair past `take`; a pinned buffer of
    //  cd_alloc_host_pairs has the same slack)
    constexpr int REPORT_BLOCKS = 32;
    // (the pass's exact kernel has posted the pairs itself: the report is the 256-byte record, one workgroup)
    const uint64_t copy_n = tb.posted ? 0 : spec_n;
    tb.posted = false;
    k_report<<<copy_n ? REPORT_BLOCKS : 1, REPORT_THREADS, 0, c->stream>>>(tb.d_state, c->d_os_ticket + 8, c->d_boxes, reinterpret_cast<Report *>(tb.h_report),
                                                              tb.d_pairs, direct ? direct : reinterpret_cast<uint32_t *>(tb.h_report + sizeof(Report)), (unsigned long long)copy_n, seq);
    return 0;
}
// Polled completion of a step (CD_OPT_POLL): spin on the sequence word k_report stores last.  Nothing that follows may need the
// stream's events (the caller checks: no stage events, no time stamps).  A step that takes longer than the spin budget, or never
// reports (a faulted kernel), ends in the ordinary stream synchronise, which also returns the error.  Every 64th polled step
// synchronises the stream as well, so that the runtime retires what it keeps per launch.
// Why the host may trust what it reads after seeing the word: DESIGN.md section 6 (system-scope release / acquire on fine-grained host memory,
// the HSA memory model's own guarantee -- the one the runtime's completion signal rests on too).  The first POLL_WARM reports into a report
// area (and into a caller's pinned buffer, told apart by serial number, not address) still end in a stream synchronise: not part of that
// argument, but free, and it makes the first steps of a context go through the path that also returns a launch's errors.
constexpr uint32_t POLL_WARM = 2;
// May this report be waited for by polling?  Also notes which pinned pair buffer the report goes to (a new one starts cold).
bool poll_this_report(const cd_ctx *c, TravBuf &tb, uint64_t direct_id)
{
    if (direct_id && direct_id != tb.direct_id) { tb.direct_id = direct_id; tb.synced_direct = 0; }     // (a step into an ordinary buffer in between does not make the pinned one cold again)
    return c->poll_opt && !c->stage_events && c->stamp_mask == 0 && tb.synced_reports >= POLL_WARM && (!direct_id || tb.synced_direct >= POLL_WARM);
}
int wait_report(cd_ctx *c, TravBuf &tb, unsigned long long seq)
{
    if (seq != 0ull) {
        const volatile unsigned long long *p = &reinterpret_cast<const volatile Report *>(tb.h_report)->seq;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t it = 1; *p != seq; ++it) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if ((it & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        if (*p == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            if ((c->polled_steps & 15u) == 0) {                               // (a clock read every 16th step: what the longest ordinary wait is)
                const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                if ((uint64_t)us > c->poll_max_wait_us) c->poll_max_wait_us = (uint32_t)us;
            }
            if ((++c->polled_steps & 63u) == 0) HIPCHK(hipStreamSynchronize(c->stream));
            return 0;
        }
        ++c->poll_fallbacks;
        const hipError_t q = hipStreamQuery(c->stream);                       // busy: the step is late; drained: the word is
        (void)hipGetLastError();
        HIPCHK(hipStreamSynchronize(c->stream));
        if (q == hipErrorNotReady) ++c->poll_fb_busy;
        else if (*p == seq) ++c->poll_fb_late_word;
        else ++c->poll_fb_lost;
        return 0;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
void parse_report(cd_ctx *c, TravBuf &tb, HostCounters &h, uint32_t *spec_pairs, uint64_t spec_n)
{
    const Report &r = *reinterpret_cast<const Report *>(tb.h_report);
    h = HostCounters{r.n_pairs, r.pairs_tested, r.node_visits, r.max_shard_candidates, r.n_deferred, r.wave_steps, r.candidates, r.clk_start_inv, r.clk_end};
    std::memcpy(c->sort_flags, r.sort_flags, sizeof c->sort_flags);
    std::memcpy(c->root_box_host, r.root_box, sizeof(double) * 6);
    if (spec_n && spec_pairs) {
        const uint64_t take = r.n_pairs < spec_n ? r.n_pairs : spec_n;
        std::memcpy(spec_pairs, tb.h_report + sizeof(Report), sizeof(uint32_t) * 2 * take);
    }
}
// What read_state needs decided before it enqueues the report -- is the step's end polled (its sequence number), is the pair area poisoned for the scan -- decided here, so that
// run_traversal can do it BEFORE the pass's kernels are enqueued: with `post` the pass's exact kernel stores the first pairs into the pinned area itself (k_exact), and a poison
// written after that would be written over the pairs.
int prepare_report(cd_ctx *c, TravBuf &tb, uint32_t *spec_pairs, uint64_t spec_n, uint64_t direct_id, bool post)
{
    if (ensure_report(tb)) return CD_ERR_ARG;
    const bool direct = direct_id != 0;
    if (spec_n > SPEC_PAIRS) spec_n = SPEC_PAIRS;
    tb.prep_seq = poll_this_report(c, tb, direct_id) ? ++c->report_seq : 0ull;
    tb.prep_area = nullptr;
    uint32_t *dst = direct ? spec_pairs : reinterpret_cast<uint32_t *>(tb.h_report + sizeof(Report));
    if (c->dbg_poll_check && tb.prep_seq && spec_pairs && spec_n) {
        tb.prep_area = dst;
        std::memset(dst, 0xff, sizeof(uint32_t) * 2 * spec_n);
    }
    tb.post_dst = (post && spec_pairs && spec_n) ? dst : nullptr;
    tb.post_n = tb.post_dst ? spec_n : 0;
    tb.prepared = true;
    return 0;
}
int read_state(cd_ctx *c, TravBuf &tb, HostCounters &h, uint32_t *spec_pairs = nullptr, uint64_t spec_n = 0, uint64_t direct_id = 0)
{
    if (ensure_report(tb)) return CD_ERR_ARG;
    const bool direct = direct_id != 0;
    if (!tb.prepared) prepare_report(c, tb, spec_pairs, spec_n, direct_id, false);
    const unsigned long long seq = tb.prep_seq;
    uint32_t *area = tb.prep_area;
    tb.prepared = false; tb.post_dst = nullptr; tb.post_n = 0;
    int rc = enqueue_report(c, tb, spec_pairs != nullptr, spec_n, direct ? spec_pairs : nullptr, seq);
    if (rc) return rc;
    const uint32_t fb0 = c->poll_fallbacks;
    if ((rc = wait_report(c, tb, seq))) return rc;
    if (seq == 0ull) { ++tb.synced_reports; if (direct) ++tb.synced_direct; }
    if (area && c->poll_fallbacks == fb0) {
        const Report &r = *reinterpret_cast<const Report *>(tb.h_report);
        const uint64_t take = r.n_pairs < spec_n ? r.n_pairs : spec_n;
        bool stale = false;
        for (uint64_t i = 0; i < 2 * take; ++i) stale |= reinterpret_cast<volatile uint32_t *>(area)[i] == 0xffffffffu;
        if (stale) ++c->poll_stale;
    }
    parse_report(c, tb, h, direct ? nullptr : spec_pairs, spec_n);          // (direct: the pairs are already where the caller wants them)
    return 0;
}

// Did a shard of the candidate buffer overflow in the pass `h` reports?  (Nothing is written past a shard; the step is redone.)
bool shards_overflowed(const TravBuf &tb, const HostCounters &h) { return h.max_shard_candidates > tb.cand_cap / NSHARD; }
int grow_candidates(cd_ctx *c, TravBuf &tb, uint64_t max_shard);
int grow_shards(cd_ctx *c, TravBuf &tb, const HostCounters &h) { return grow_candidates(c, tb, h.max_shard_candidates); }
int grow_candidates(cd_ctx *c, TravBuf &tb, uint64_t max_shard)
{
    hipFree(tb.d_cand); tb.d_cand = nullptr; tb.cand_cap = 0;
    const uint64_t want = ((max_shard + max_shard / 4 + 1024 + 3) & ~3ull) * NSHARD;          // (a shard: a multiple of 4 slots -- FatPair entries are counted from its end, cd_traverse.h)
    HIPCHK(hipMalloc(&tb.d_cand, sizeof(Candidates) * want));
    tb.cand_cap = want;
    return 0;
}

// Traversal (local leaves or external queries).  Blocks: reads the counters, runs the deep pass when needed.
int run_traversal(cd_ctx *c, TravBuf &tb, const void *d_ext, uint64_t nq_ext, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs)
{
    const uint32_t n = c->nt;
    hipStream_t s = c->stream;
    const bool external = d_ext != nullptr;
    const uint32_t nq = external ? (uint32_t)nq_ext : n;
    int rc = ensure_pairs(c, tb, cap_pairs > 0 ? cap_pairs : 1);
    if (rc) return rc;
    const int per_pass = c->trav_variant == 0 ? 1 : 2;             // descent + exact kernel
    if (c->trav_variant == 0 && !(c->internal_boxes_valid && c->hierarchy_valid)) {   // variant 0 walks meta[] and the FP64 boxes of the internal nodes
        if (!c->hierarchy_valid && (rc = enqueue_hierarchy(c, false))) return rc;
        if ((rc = enqueue_refit(c, true, false))) return rc;
    }
    // (the from-the-root descent reads its query boxes from qbox[], k_exact the leaf boxes of candidates that are not certain -- external queries against a tree whose
    //  build did not store them: filled here, once per tree)
    if (c->trav_variant != 0 && (external || c->trav_variant < 3) && (rc = ensure_qbox(c))) return rc;
    uint32_t launches = 0;
    float deep_ms = 0.f;
    HostCounters h = {};
    uint64_t spec_valid = 0;
    bool done = false;
    c->stats.stack_overflows = 0;
    for (int attempt = 0; attempt < 8 && !done; ++attempt) {
        launches = 0; deep_ms = 0.f;
        QuerySrc src{c->d_leaf, c->d_boxes, c->d_qbox, c->d_root, d_ext, nullptr, c->d_os_ticket + 8};
        const bool will_ride = !c->stage_events && (c->trav_variant == 1 || c->trav_variant >= 3) && nq > 0;   // see launch_pass: events on the dispatch packets
        c->events_ride = false;
        if (!will_ride) HIPCHK(evrec(c, EV_TRAV0));
        if (!(c->prezeroed && attempt == 0 && &tb == &c->tb[0])) HIPCHK(hipMemsetAsync(tb.d_state, 0, sizeof(TravState), s));
        const uint64_t spec_n = pairs ? (cap_pairs < SPEC_PAIRS ? cap_pairs : SPEC_PAIRS) : 0;
        // (the exact kernel posts the first pairs itself when it is the one that appends them: every traversal but variant 0's)
        if ((rc = prepare_report(c, tb, pairs, spec_n, pinned_pairs_id(pairs, cap_pairs), c->trav_variant != 0 && nq > 0 && !c->dbg_report_copies))) return rc;
        if (nq > 0) {
            if (external) launch_pass<true, false>(c, tb, src, nq, cap_pairs); else launch_pass<false, false>(c, tb, src, nq, cap_pairs);
            launches += per_pass;
        }
        if (!c->events_ride) HIPCHK(evrec(c, EV_TRAV1));      // device time of the kernels only: recorded before the read-back
        if ((rc = read_state(c, tb, h, pairs, spec_n, pinned_pairs_id(pairs, cap_pairs)))) return rc;
        spec_valid = spec_n;
        if (shards_overflowed(tb, h)) { if ((rc = grow_shards(c, tb, h))) return rc; continue; }
        if (h.n_deferred > tb.defer_cap) {               // the deferred list was too small: grow it and redo (pairs restart at 0)
            hipFree(tb.d_defer); tb.d_defer = nullptr; tb.defer_cap = 0;
            HIPCHK(hipMalloc(&tb.d_defer, sizeof(uint2) * (size_t)h.n_deferred));
            tb.defer_cap = h.n_deferred;
            continue;
        }
        if (h.n_deferred > 0) {                          // deep pass over the deferred (query, subtree) items
            const uint32_t nd = h.n_deferred;
            const uint64_t deep_lanes = (uint64_t)cdiv(nd, 64) * 64 + 256;
            if (deep_lanes > tb.deep_items) {
                hipFree(tb.d_deep); tb.d_deep = nullptr; tb.deep_items = 0;
                HIPCHK(hipMalloc(&tb.d_deep, sizeof(int32_t) * (size_t)DEEP_STACK * deep_lanes));
                tb.deep_items = deep_lanes;
            }
            if (c->trav_variant != 0 && (rc = ensure_qbox(c))) return rc;      // (the deep pass continues with the from-the-root descent)
            HIPCHK(evrec(c, EV_DEEP0));
            HIPCHK(hipMemsetAsync(&tb.d_state->n_deferred, 0, sizeof(uint32_t), s));
            // candidate shards restart from 0: the shallow pass's candidates have all been consumed by k_exact
            for (int i = 0; i < NSHARD; ++i) { /* one memset per shard would be 64 calls: clear them with a 2-D memset */ }
            HIPCHK(hipMemset2DAsync(&tb.d_state->shard[0].n_candidates, sizeof(CtrShard), 0, sizeof(unsigned long long), NSHARD, s));
            src.list = tb.d_defer;
            if (external) launch_pass<true, true>(c, tb, src, nd, cap_pairs); else launch_pass<false, true>(c, tb, src, nd, cap_pairs);
            launches += per_pass;
            HIPCHK(evrec(c, EV_DEEP1));
            if ((rc = read_state(c, tb, h))) return rc;
            if (h.n_deferred != 0) return CD_ERR_ARG;    // tree deeper than DEEP_STACK: cannot happen (height <= 96)
            if (shards_overflowed(tb, h)) { if ((rc = grow_shards(c, tb, h))) return rc; continue; }
            c->stats.stack_overflows = nd;
            deep_ms = elapsed(c, EV_DEEP0, EV_DEEP1);            // (always recorded: a deep pass is device time of this traversal)
        }
        done = true;
    }
    if (!done) return CD_ERR_ARG;
    HIPCHK(hipGetLastError());
    const uint64_t found = h.n_pairs;
    const uint64_t ncopy = found < cap_pairs ? found : cap_pairs;
    // the first spec_valid pairs came back with the counters -- unless a deep pass appended more afterwards
    const uint64_t have = (c->stats.stack_overflows == 0) ? spec_valid : 0;
    if (pairs && ncopy > have) {
        HIPCHK(hipMemcpyAsync(pairs + 2 * have, tb.d_pairs + 2 * have, sizeof(uint32_t) * 2 * (ncopy - have), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    // (a time stamp that was not taken in this call must not be read: the event holds an earlier call's)
    const bool have_d = !c->events_ride || (c->stamp_mask & 2u), have_x = !c->events_ride || (c->stamp_mask & 4u);
    c->stats.ms_traverse = (have_d && have_x) ? elapsed(c, EV_TRAV0, EV_TRAV1) + deep_ms : 0.f;
    c->stats.ms_descend = (c->trav_variant != 0 && nq > 0 && have_d) ? elapsed(c, EV_TRAV0, EV_DESC1) : 0.f;
    c->stats.ms_exact = (c->trav_variant != 0 && nq > 0 && have_d && have_x) ? elapsed(c, EV_DESC1, EV_TRAV1) : 0.f;
    c->stats.traverse_launches = launches;
    c->stats.n_pairs = found; c->stats.pairs_tested = h.pairs_tested; c->stats.node_visits = h.node_visits;
    c->stats.wave_steps = h.wave_steps; c->stats.candidates = h.candidates;
    // the descent kernel's own clock (k_descend_half): first wave start -> last wave end, in ticks of the device's constant wall clock
    c->stats.ms_descend_clock = (h.clk_end && h.clk_start_inv && c->wall_clock_khz > 0 && c->stats.stack_overflows == 0)
                                    ? (float)((double)(h.clk_end - ~h.clk_start_inv) / (double)c->wall_clock_khz) : 0.f;
    c->last_pairs_on_device = found <= cap_pairs ? found : cap_pairs;
    if (n_pairs) *n_pairs = found;
    return found > cap_pairs ? CD_OVERFLOW : CD_OK;
}


// ---- pair-list post-processing -----------------------------------------------------------------------
int pp_reserve(cd_ctx *c, uint32_t m)
{
    if (m <= c->pp_cap) return 0;
    for (int i = 0; i < 2; ++i) { hipFree(c->pp_keys[i]); hipFree(c->pp_vals[i]); c->pp_keys[i] = nullptr; c->pp_vals[i] = nullptr; }
    hipFree(c->pp_flags); hipFree(c->pp_os); c->pp_flags = nullptr; c->pp_os = nullptr; c->pp_cap = 0;
    const uint32_t cap = m + m / 4 + 4096;
    const uint32_t ntiles = cdiv(cap, SORT_TILE);
    for (int i = 0; i < 2; ++i) { HIPCHK(hipMalloc(&c->pp_keys[i], sizeof(uint64_t) * cap)); HIPCHK(hipMalloc(&c->pp_vals[i], sizeof(uint32_t) * cap)); }
    HIPCHK(hipMalloc(&c->pp_flags, sizeof(uint32_t) * cap));
    c->pp_os_bytes = sizeof(uint32_t) * 8 * RADIX + 64 + sizeof(unsigned long long) * 8 * (size_t)ntiles * RADIX;
    HIPCHK(hipMalloc(&c->pp_os, c->pp_os_bytes));
    c->pp_cap = cap;
    return 0;
}

// Ascending radix sort of pp_keys[0][0..m) (onesweep, cd_sort.h); result back in pp_keys[0].
int pp_sort(cd_ctx *c, uint32_t m)
{
    hipStream_t s = c->stream;
    const uint32_t ntiles = cdiv(m, SORT_TILE);
    uint32_t *hist = reinterpret_cast<uint32_t *>(c->pp_os);
    uint32_t *ticket = hist + 8 * RADIX;
    unsigned long long *look = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(c->pp_os) + sizeof(uint32_t) * 8 * RADIX + 64);
    HIPCHK(hipMemsetAsync(c->pp_os, 0, sizeof(uint32_t) * 8 * RADIX + 64 + sizeof(unsigned long long) * 8 * (size_t)ntiles * RADIX, s));
    k_os_hist<<<cdiv(m, SORT_THREADS * 8) < 1024u ? cdiv(m, SORT_THREADS * 8) : 1024u, SORT_THREADS, 0, s>>>(c->pp_keys[0], m, hist);
    int cur = 0;
    for (int pass = 0; pass < 8; ++pass) {
        k_os_pass<<<ntiles, OS_THREADS, 0, s>>>(c->pp_keys[cur], c->pp_vals[cur], c->pp_keys[cur ^ 1], c->pp_vals[cur ^ 1], m, pass * RADIX_BITS,
                                                hist + pass * RADIX, look + (size_t)pass * ntiles * RADIX, ticket + pass, pass == 0, 1, nullptr, nullptr);
        cur ^= 1;
    }
    return 0;
}

int judge_sort_flags(cd_ctx *c);
// ---- CD_OPT_GRAPH: the steady-state fused step as ONE hipGraph launch ------------------------------------------------
// Eligible: no stage events, no time stamps, hybrid sort, fused build, half traversal, and a previous step has left the
// scratch clean (so the captured sequence holds kernels only: no memset).  The capture records exactly what the stream path
// enqueues -- k_morton, 2 x k_os_pass, k_local_sort, k_build_block, k_cross_fused, k_descend_half, k_exact, k_report -- with
// the arguments of this call baked in; any change of those (capacity, buffers grown, options) captures again.  Whatever the
// report says that the stream path would answer with a retry -- a sort flag, a candidate shard that overflowed, a deferred
// subtree -- is handed to the stream path (handled = false, or run_traversal on the tree that is there).
void graph_drop(cd_ctx *c)
{
    if (c->graph_exec) { hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
}
// The cell table follows the VERTICES: rebuilt by cd_create / cd_update_vertices (after their copy), never inside a step.  Every
// coordinate an fp32 value -- the reference's loader produces those (load_obj.h:38) -- : one streaming pass, no table.
int amb_refresh(cd_ctx *c)
{
    hipStream_t s = c->stream;
    const uint64_t n3 = 3ull * c->nv;
    if (!c->d_amb_flag) HIPCHK(hipMalloc(&c->d_amb_flag, sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(c->d_amb_flag, 0, sizeof(uint32_t), s));
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n3 + 255) / 256, 4096);
    k_amb_scan<<<blocks, 256, 0, s>>>(c->d_verts, n3, c->d_amb_flag);
    uint32_t inexact = 0;
    HIPCHK(hipMemcpyAsync(&inexact, c->d_amb_flag, sizeof inexact, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const AmbTable before = c->amb;
    if (!inexact) { c->amb = AmbTable{nullptr, 0u, 0u}; c->vamb = nullptr; }
    else if (!c->cell_table_opt) { c->amb = AmbTable{nullptr, 0u, 1u}; c->vamb = nullptr; }      // every inexact coordinate counts as ambiguous: plain outward rounding
    else {
        uint64_t cap = 1024; uint32_t lg = 10;
        while (cap < n3 + n3 / 2) { cap <<= 1; ++lg; }                      // load below 2 / 3
        if (cap > (1ull << 32)) return CD_ERR_ARG;
        if (cap != c->amb_cap) {
            hipFree(c->d_amb_keys); c->d_amb_keys = c->d_amb_vals = nullptr; c->amb_cap = 0;
            HIPCHK(hipMalloc(&c->d_amb_keys, sizeof(unsigned long long) * 2 * cap));       // keys | vals, one allocation, one memset
            c->d_amb_vals = c->d_amb_keys + cap;
            c->amb_cap = cap;
        }
        HIPCHK(hipMemsetAsync(c->d_amb_keys, 0, sizeof(unsigned long long) * 2 * cap, s));
        const uint32_t shift = 64u - lg, mask = (uint32_t)(cap - 1);
        k_amb_insert<<<blocks, 256, 0, s>>>(c->d_verts, n3, c->d_amb_keys, c->d_amb_vals, shift, mask);
        c->amb = AmbTable{c->d_amb_keys, shift, mask};
        if (!c->d_vamb) HIPCHK(hipMalloc(&c->d_vamb, c->nv));
        k_amb_vertex<<<(uint32_t)std::min<uint64_t>((c->nv + 255ull) / 256, 4096), 256, 0, s>>>(c->d_verts, c->nv, c->amb, c->d_vamb);
        c->vamb = c->d_vamb;
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipGetLastError());
    }
    if (before.keys != c->amb.keys || before.shift != c->amb.shift || before.mask != c->amb.mask) graph_drop(c);       // (a captured step has the table baked into its launches)
    return CD_OK;
}
bool graph_eligible(const cd_ctx *c)
{
    return c->graph_opt && !c->stage_events && c->stamp_mask == 0 && c->sort_mode <= 1 && fused_build_next(c) && c->trav_variant == 3 &&
           c->scratch_clean && !c->dbg_diag && !c->dbg_lds_pad && c->nt > 1 && c->tree_done_event == nullptr;
}
int graph_step(cd_ctx *c, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs, bool &handled)
{
    handled = false;
    TravBuf &tb = c->tb[0];
    hipStream_t s = c->stream;
    int rc = ensure_pairs(c, tb, cap_pairs > 0 ? cap_pairs : 1);
    if (!rc) rc = ensure_report(tb);
    if (rc) return rc;
    uint64_t spec_n = pairs ? (cap_pairs < SPEC_PAIRS ? cap_pairs : SPEC_PAIRS) : 0;
    const bool direct = pinned_pairs_id(pairs, cap_pairs) != 0;
    const cd_ctx::GraphKey key{cap_pairs, spec_n, c->sort_mode, c->trav_variant, c->frame_mode, c->nt, (uint32_t)c->exact_blocks, c->dbg_no_shared_path | (c->dbg_split_cross << 1) | (c->order_hint ? 4u : 0u) | (c->order_ready ? 8u : 0u) | (c->local_small_ok ? 16u : 0u) | (c->dbg_sort_windows << 5) | (qbox_wanted(c) ? 128u : 0u) /* (whether the descent's launch reads the order hint is baked in) */,
                               tb.d_pairs, tb.d_cand, tb.d_defer, direct ? (const void *)pairs : (const void *)tb.h_report, tb.cand_cap, tb.defer_cap, 0u};
    static_assert(sizeof(cd_ctx::GraphKey) == 2 * 8 + 6 * 4 + 4 * 8 + 8 + 2 * 4, "GraphKey has no padding");
    if (!c->graph_exec || std::memcmp(&key, &c->graph_key, sizeof key) != 0) {
        graph_drop(c);
        HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        c->prezeroed = true;
        rc = enqueue_morton_sort(c, false);
        if (!rc) rc = enqueue_tree(c);
        if (!rc) {
            QuerySrc src{c->d_leaf, c->d_boxes, c->d_qbox, c->d_root, nullptr, nullptr, c->d_os_ticket + 8};
            launch_pass<false, false>(c, tb, src, c->nt, cap_pairs);
            rc = enqueue_report(c, tb, pairs != nullptr, spec_n, direct ? pairs : nullptr);
        }
        c->prezeroed = false;
        hipGraph_t g = nullptr;
        const hipError_t ee = hipStreamEndCapture(s, &g);
        if (rc || ee != hipSuccess || !g) { if (g) hipGraphDestroy(g); (void)hipGetLastError(); c->scratch_clean = false; return CD_OK; }   // (not handled: the stream path runs, with its memset)
        c->graph = g;
        if (hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0) != hipSuccess) { graph_drop(c); (void)hipGetLastError(); c->scratch_clean = false; return CD_OK; }
        std::memset(&c->graph_key, 0, sizeof c->graph_key);
        c->graph_key = key;
        c->graph_post = cd_ctx::GraphPost{c->leaves_filled, c->leaf_records_filled, c->hierarchy_valid, c->internal_boxes_valid, c->last_tree_fused, c->events_ride, c->scratch_clean, c->qbox_valid,
                                          c->stats.sort_passes};
        ++c->graph_captures;
    }
    // what the enqueue functions leave behind on the host side, as the capture left it
    const cd_ctx::GraphPost &gp = c->graph_post;
    c->leaves_filled = gp.leaves_filled; c->leaf_records_filled = gp.leaf_records_filled; c->hierarchy_valid = gp.hierarchy_valid;
    c->internal_boxes_valid = gp.internal_boxes_valid; c->last_tree_fused = gp.last_tree_fused; c->events_ride = gp.events_ride; c->scratch_clean = gp.scratch_clean; c->qbox_valid = gp.qbox_valid;
    c->stats.sort_passes = gp.sort_passes;
    HIPCHK(hipGraphLaunch(c->graph_exec, s));
    HIPCHK(hipStreamSynchronize(s));
    ++c->graph_replays;
    HostCounters h = {};
    parse_report(c, tb, h, direct ? nullptr : pairs, spec_n);
    c->stage = ST_REFIT; c->root_box_valid = true;
    { const int js = judge_sort_flags(c); if (js != CD_OK) return CD_OK; }                 // (not handled: the stream path redoes the step in the sort's next form)
    if (shards_overflowed(tb, h) || h.n_deferred > 0) {             // the tree is fine: the traversal again, with the stream path's retries / deep pass
        handled = true;
        c->scratch_clean = false;
        return run_traversal(c, tb, nullptr, 0, pairs, cap_pairs, n_pairs);
    }
    handled = true;
    const uint64_t found = h.n_pairs, ncopy = found < cap_pairs ? found : cap_pairs;
    if (pairs && ncopy > spec_n) {
        HIPCHK(hipMemcpyAsync(pairs + 2 * spec_n, tb.d_pairs + 2 * spec_n, sizeof(uint32_t) * 2 * (ncopy - spec_n), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    c->stats.ms_morton = c->stats.ms_sort = c->stats.ms_hierarchy = c->stats.ms_refit = c->stats.ms_traverse = c->stats.ms_descend = c->stats.ms_exact = 0.f;
    c->stats.ms_pipeline = c->stats.ms_build_block = 0.f;
    c->stats.traverse_launches = 0; c->stats.stack_overflows = 0;       // (0: this call launched no kernel of its own -- one graph launch)
    c->stats.n_pairs = found; c->stats.pairs_tested = h.pairs_tested; c->stats.node_visits = h.node_visits;
    c->stats.wave_steps = h.wave_steps; c->stats.candidates = h.candidates;
    c->stats.ms_descend_clock = (h.clk_end && h.clk_start_inv && c->wall_clock_khz > 0) ? (float)((double)(h.clk_end - ~h.clk_start_inv) / (double)c->wall_clock_khz) : 0.f;
    c->last_pairs_on_device = ncopy;
    if (n_pairs) *n_pairs = found;
    return found > cap_pairs ? CD_OVERFLOW : CD_OK;
}

}  // namespace

void multi_detach_from(cd_ctx *c);          // cd_multi.h

extern "C" {

const char *cd_version(void) { return "mi355cd 0.1 gfx950"; }

int cd_create(cd_ctx **out, const double *verts_xyz, uint32_t nv, const uint32_t *vidx3, const uint32_t *ids, uint32_t nt)
{
    if (!out || !verts_xyz || !vidx3 || nv == 0 || nt == 0) return CD_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return CD_ERR_NO_DEVICE;
    for (uint64_t k = 0; k < 3ull * nt; ++k) if (vidx3[k] >= nv) return CD_ERR_INDEX;
    bool all_ref = true;                                // is every vertex used by some triangle?  (then the box of the triangles is the box of the vertices: cd_multi.h)
    { std::vector<uint8_t> used(nv, 0);
      for (uint64_t k = 0; k < 3ull * nt; ++k) used[vidx3[k]] = 1;
      for (uint32_t v = 0; v < nv && all_ref; ++v) all_ref = used[v] != 0; }
    cd_ctx *c = new (std::nothrow) cd_ctx();
    if (!c) return CD_ERR_ARG;
    c->nv = nv; c->nt = nt;
    c->all_verts_referenced = all_ref;
    c->ntiles = cdiv(nt, SORT_TILE);
    { int dev = 0, khz = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess) c->wall_clock_khz = khz; (void)hipGetLastError(); }
    const size_t n = nt;
#define ALLOC(p, bytes) do { hipError_t e_ = hipMalloc((void **)&(p), (bytes)); if (e_ != hipSuccess) { free_all(c); delete c; return -(int)e_; } } while (0)
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return -(int)e; }
    for (int i = 0; i < EV_COUNT; ++i) { e = hipEventCreate(&c->ev[i]); if (e != hipSuccess) { free_all(c); delete c; return -(int)e; } }
    ALLOC(c->d_verts, sizeof(double) * 3 * (size_t)nv);
    ALLOC(c->d_vidx, sizeof(uint32_t) * 3 * n);
    if (ids) ALLOC(c->d_ids, sizeof(uint32_t) * n);
    for (int i = 0; i < 2; ++i) { ALLOC(c->d_keys[i], sizeof(uint64_t) * n); ALLOC(c->d_perm[i], sizeof(uint32_t) * n); }
    ALLOC(c->d_counts, sizeof(uint32_t) * RADIX * c->ntiles);
    ALLOC(c->d_chunk_tot, sizeof(uint32_t) * RADIX * cdiv(c->ntiles, (uint32_t)OS_CHUNK));
    // one scratch block so that the fused pipeline zeroes everything with ONE memset, and as little as the sort form needs:
    //   [onesweep: histograms | tickets, flags | look-back granules of passes 6, 7] [small counters: 128 words] [TravState]   <- hybrid sort: this much
    //   [look-back granules of passes 0..5]                                                                                  <- the other forms: all of it
    const size_t gran = sizeof(unsigned long long) * (size_t)c->ntiles * RADIX;           // one pass
    const size_t look_off = sizeof(uint32_t) * HIST_COPIES * 8 * RADIX + 128;              // hist (HIST_COPIES partial tables) | 8 tickets, 8 time-out flags, fix-up flag, pad
    c->sort_hi_bytes = look_off + 2 * gran;
    const size_t small_off = (c->sort_hi_bytes + 127) & ~(size_t)127, state_off = small_off + 512;
    c->zero_bytes = state_off + sizeof(TravState);
    const size_t lo_off = (c->zero_bytes + 127) & ~(size_t)127;
    c->os_bytes = lo_off + 6 * gran;
    ALLOC(c->d_os, c->os_bytes);
    c->d_small = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(c->d_os) + small_off);
    c->tb[0].d_state = reinterpret_cast<TravState *>(reinterpret_cast<char *>(c->d_os) + state_off);
    c->d_root = reinterpret_cast<int32_t *>(c->d_small + 96);
    c->d_os_hist = reinterpret_cast<uint32_t *>(c->d_os);
    c->d_os_ticket = c->d_os_hist + HIST_COPIES * 8 * RADIX;
    c->d_os_look = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(c->d_os) + look_off);
    c->d_os_look_lo = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(c->d_os) + lo_off);
    ALLOC(c->d_frame, sizeof(double) * FRAME_WORDS);
    ALLOC(c->d_partial, sizeof(double) * BOUNDS_STRIDE * BOUNDS_BLOCKS);
    ALLOC(c->d_leaf, sizeof(LeafTri) * n);
    // a failed sort may leave slots unwritten for one (discarded) run: keep their vertex ids in range
    { hipError_t e_ = hipMemset(c->d_leaf, 0, sizeof(LeafTri) * n); if (e_ != hipSuccess) { free_all(c); delete c; return -(int)e_; } }
    ALLOC(c->d_meta, sizeof(NodeMeta) * n);
    ALLOC(c->d_parent, sizeof(int32_t) * 2 * n);
    { uint32_t nb = cdiv(nt, REFIT_BLK); c->nbp2 = 1; while (c->nbp2 < nb) c->nbp2 <<= 1; }
    if (c->nbp2 / (uint32_t)TOP_IN_BLOCK > (uint32_t)TOP_MAX_SPANS) { free_all(c); delete c; return CD_ERR_ARG; }   // k_top_publish_upper folds the span roots in LDS sized for 2^30 leaves (cd_build.h)
    ALLOC(c->d_seg, sizeof(double) * 6 * (size_t)c->nbp2 * REFIT_BLK);
    ALLOC(c->d_top_pub, top_pub_bytes(c));
    if (hipMemset(c->d_top_pub, 0, top_pub_bytes(c)) != hipSuccess) { free_all(c); delete c; return -(int)hipGetLastError(); }
    ALLOC(c->d_seg32, sizeof(float) * 6 * ((size_t)c->nbp2 << (REFIT_LOG - SEG32_MIN_LEVEL + 1)));   // fused build: levels SEG32_MIN_LEVEL .. 9 of the blocks' fp32 trees
    c->cross_cap = nt;                                  // every internal node could be one (it never is: about 2 %)
    ALLOC(c->d_cross, sizeof(int32_t) * (size_t)c->cross_cap);
    ALLOC(c->d_boxes, sizeof(double) * 6 * 2 * n);
    ALLOC(c->d_bounded, sizeof(uint32_t) * n);
    ALLOC(c->d_recs32, sizeof(NodeRec32) * n);
    ALLOC(c->d_qbox, sizeof(LeafBox32) * n);
    ALLOC(c->d_leaf_side, sizeof(unsigned long long) * ((n + 63) / 64 + 1));
    ALLOC(c->d_split_of, sizeof(int32_t) * n);
    { const size_t groups = ((size_t)n + 63) / 64;
      ALLOC(c->d_cost, sizeof(uint32_t) * groups); ALLOC(c->d_order, sizeof(uint32_t) * groups); ALLOC(c->d_tri_cost, n);
      if (hipMemset(c->d_cost, 0, sizeof(uint32_t) * groups) != hipSuccess || hipMemset(c->d_tri_cost, 0, n) != hipSuccess) { free_all(c); delete c; return CD_ERR_ARG; } }   // (no times yet: the first hint is half_vblock's own order)
    c->tb[0].cand_cap = ((4 * n > (1u << 20) ? 4 * n : (1u << 20)) + 4 * NSHARD - 1) / (4 * NSHARD) * (4 * NSHARD);    // (a shard: a multiple of 4 slots, cd_traverse.h: FatPair)
    ALLOC(c->tb[0].d_cand, sizeof(Candidates) * c->tb[0].cand_cap);
    c->tb[0].defer_cap = 1u << 16;
    ALLOC(c->tb[0].d_defer, sizeof(uint2) * c->tb[0].defer_cap);
#undef ALLOC
    // main.cu:86-88 H2D
    bool ok = hipMemcpy(c->d_verts, verts_xyz, sizeof(double) * 3 * (size_t)nv, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(c->d_vidx, vidx3, sizeof(uint32_t) * 3 * n, hipMemcpyHostToDevice) == hipSuccess &&
              (!ids || hipMemcpy(c->d_ids, ids, sizeof(uint32_t) * n, hipMemcpyHostToDevice) == hipSuccess) &&
              hipMemcpy(c->d_frame, c->frame_host, sizeof(double) * FRAME_WORDS, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { free_all(c); delete c; return -(int)hipGetLastError(); }
    { const int rc = amb_refresh(c); if (rc) { free_all(c); delete c; return rc; } }
    *out = c;
    return CD_OK;
}

void cd_destroy(cd_ctx *c)
{
    if (!c) return;
    multi_detach_from(c);                   // a cd_multi still attached keeps no pointer into a freed context (its next step returns CD_ERR_ORDER)
    hipStreamSynchronize(c->stream);
    free_all(c);
    delete c;
}

int cd_update_vertices(cd_ctx *c, const double *verts_xyz)
{
    if (!c || !verts_xyz) return CD_ERR_ARG;
    HIPCHK(hipStreamSynchronize(c->stream));                                // (nothing of an earlier call may still read the old vertices or the old cell table)
    HIPCHK(hipMemcpy(c->d_verts, verts_xyz, sizeof(double) * 3 * (size_t)c->nv, hipMemcpyHostToDevice));
    c->stage = ST_CREATED;
    c->root_box_valid = false;
    return amb_refresh(c);
}

static int install_frame(cd_ctx *c, int mode, const double off[3], const double span[3], unsigned long long layout)
{
    if (off) for (int a = 0; a < 3; ++a) { c->frame_host[a] = off[a]; c->frame_host[3 + a] = span[a]; }
    std::memcpy(&c->frame_host[6], &layout, sizeof layout); c->frame_host[7] = 0.0;
    c->frame_mode = mode;
    HIPCHK(hipStreamSynchronize(c->stream));                                // (nothing of an earlier call may still read the old frame)
    HIPCHK(hipMemcpy(c->d_frame, c->frame_host, sizeof(double) * FRAME_WORDS, hipMemcpyHostToDevice));
    c->stage = ST_CREATED;
    return CD_OK;
}
int cd_set_morton_frame(cd_ctx *c, int mode, const double offset[3], const double span[3])
{
    if (!c) return CD_ERR_ARG;
    if (mode == CD_FRAME_REFERENCE) {
        const double ref[6] = {0.004501, -0.476622, -0.381965, 3.08, 0.76, 2.36};
        return install_frame(c, mode, ref, ref + 3, 0ull);
    }
    if (mode == CD_FRAME_CUSTOM) {
        if (!offset || !span) return CD_ERR_ARG;
        return install_frame(c, mode, offset, span, 0ull);                  // morton.h:70-89's interleave in the caller's frame
    }
    if (mode != CD_FRAME_AUTO) return CD_ERR_ARG;
    return install_frame(c, mode, nullptr, nullptr, 0ull);                  // (the step computes the frame and writes it to d_frame)
}
// A frame WITH its key layout (cd_math.h), as cd_get_morton_frame returned it -- from this context after a step in CD_FRAME_AUTO (a frame computed
// once and kept: the AUTO pass over the triangles, ~11 us at 1 M, leaves the step), or from another one (all ranks of a job in one frame).
int cd_set_morton_frame_layout(cd_ctx *c, const double offset[3], const double span[3], uint64_t layout)
{
    if (!c || !offset || !span || !layout_ok(layout)) return CD_ERR_ARG;
    for (int a = 0; a < 3; ++a) if (!(span[a] > 0.0)) return CD_ERR_ARG;
    return install_frame(c, CD_FRAME_CUSTOM, offset, span, layout);
}
// The frame the last sort used: offset, span and key layout (0: the reference's interleave).
int cd_get_morton_frame(cd_ctx *c, double offset[3], double span[3], uint64_t *layout)
{
    if (!c) return CD_ERR_ARG;
    if (c->stage < ST_SORTED) return CD_ERR_ORDER;
    double f[FRAME_WORDS];
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(f, c->d_frame, sizeof f, hipMemcpyDeviceToHost));
    for (int a = 0; a < 3; ++a) { if (offset) offset[a] = f[a]; if (span) span[a] = f[3 + a]; }
    if (layout) std::memcpy(layout, &f[6], sizeof *layout);
    return CD_OK;
}

// the onesweep look-back spins are bounded; a timeout sets one of the words d_os_ticket[8..15]
constexpr int SORT_REDO = 77;                   // internal: this form of the sort could not finish, redo with the one the flags ask for
constexpr int SORT_REDO_MAX = 4;                // redos one call can need: small windows -> large windows -> half-key -> full is three (judge_sort_flags); one to spare
namespace { int judge_sort_flags(cd_ctx *c)
{
    for (int i = 0; i < 9; ++i) if (c->sort_flags[i]) c->scratch_clean = false;     // the flag words are cleared by the memset only
    for (int i = 0; i < 8; ++i) if (c->sort_flags[i]) return CD_ERR_SORT;
    const uint32_t f = c->sort_flags[8];
    if (!f) return CD_OK;
    // The flags say WHY (cd_sort.h): the next form is the one that can finish, not the next in line (round 5 walked 0 -> 1 -> 2 -> 3: a mesh with 17 coincident centroids
    // took four redos, one more than the multi-GPU step allowed itself).
    if (c->sort_mode >= 3) return CD_ERR_SORT;                              // (the full form raises nothing)
    if (f & SORTF_FIXUP) { c->sort_mode = 3; c->left_frame = false; return SORT_REDO; }      // too many equal high halves: only the eight passes do without a fix-up hop
    if (f & SORTF_RUN) {
        // a run too long for the small windows: the large form, same passes -- unless keys also lie above the digits (then another form is due anyway)
        if (!(f & SORTF_ABOVE) && c->local_small_active && c->local_small_ok && c->sort_mode <= 1) { c->local_small_ok = false; return SORT_REDO; }
        c->sort_mode = 2; c->left_frame = false; return SORT_REDO;          // no windows: four global passes + fix-up (digits at bits 48..63 only make runs longer)
    }
    // SORTF_ABOVE alone: some key lies beyond the shifted digits (a centroid outside the Morton frame), no run was too long.  A mesh that leaves the frame may come
    // back: a context that went 0 -> 1 for THAT reason tries the first form again every SORT_RETRY_STEPS steps (sort_retry_tick; a try that fails costs one redone step in 64).
    if (c->sort_mode == 0) { c->sort_mode = 1; c->left_frame = true; c->steps_in_mode1 = 0; return SORT_REDO; }
    c->sort_mode = 2; c->left_frame = false; return SORT_REDO;
} }
// Once per CALL of a stepping entry point (not per redo), BEFORE a graph step computes its key: a context whose mesh had left the Morton frame (sort form 1) tries
// the first form again every SORT_RETRY_STEPS steps.  (Round 5 counted inside enqueue_morton_sort, which a graph replay does not run: ADVICE r05.)
static void sort_retry_tick(cd_ctx *c)
{
    if (c->sort_mode == 1 && c->left_frame && ++c->steps_in_mode1 >= SORT_RETRY_STEPS) { c->sort_mode = 0; c->steps_in_mode1 = 0; c->local_small_ok = true; graph_drop(c); }   // (the second form's runs are 16 x longer: the small windows get their chance again too)
}
static int check_sort_flags(cd_ctx *c)
{
    HIPCHK(hipMemcpyAsync(c->sort_flags, c->d_os_ticket + 8, sizeof c->sort_flags, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return judge_sort_flags(c);
}

int cd_morton_sort(cd_ctx *c)
{
    if (!c) return CD_ERR_ARG;
    sort_retry_tick(c);
    int rc = enqueue_morton_sort(c);
    if (rc) return rc;
    rc = check_sort_flags(c);
    for (int redo = 0; rc == SORT_REDO && redo < SORT_REDO_MAX; ++redo) { if ((rc = enqueue_morton_sort(c))) return rc; rc = check_sort_flags(c); }
    if (rc == SORT_REDO) rc = CD_ERR_SORT;
    if (rc) return rc;
    c->stats.ms_morton = elapsed(c, EV_MORTON0, EV_MORTON1);
    c->stats.ms_sort = elapsed(c, EV_MORTON1, EV_SORT1);
    c->stage = ST_SORTED;
    return CD_OK;
}

int cd_build_hierarchy(cd_ctx *c, uint32_t *parent_wrong_num)
{
    if (!c) return CD_ERR_ARG;
    if (c->stage < ST_SORTED) return CD_ERR_ORDER;
    int rc = enqueue_hierarchy(c, true);
    if (rc) return rc;
    uint32_t wrong = 0;
    HIPCHK(hipMemcpyAsync(&wrong, c->d_small, sizeof wrong, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stats.ms_hierarchy = elapsed(c, EV_HIER0, EV_HIER1);
    if (parent_wrong_num) *parent_wrong_num = wrong;
    c->stage = ST_BUILT;
    return CD_OK;
}

int cd_refit_boxes(cd_ctx *c)
{
    if (!c) return CD_ERR_ARG;
    c->root_box_valid = false;
    if (c->stage < ST_BUILT) return CD_ERR_ORDER;
    int rc = enqueue_refit(c, true);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stats.ms_refit = elapsed(c, EV_REFIT0, EV_REFIT1);
    c->stage = ST_REFIT;
    return CD_OK;
}

// Fused calls do not write the FP64 boxes of the internal nodes (nothing on their path reads them); whoever asks for
// them afterwards -- the exported tree, checkInternalNodes' uninitialised-box counter -- gets them from a full refit.
static int materialise_internal_boxes(cd_ctx *c)
{
    if (c->stage < ST_REFIT || (c->internal_boxes_valid && c->hierarchy_valid)) return CD_OK;
    int rc = CD_OK;
    if (!c->hierarchy_valid) rc = enqueue_hierarchy(c, false);             // k_fill_leaves (parent links reset) + k_hierarchy on the current keys
    if (!rc) rc = enqueue_refit(c, true, false);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return CD_OK;
}

static int run_check(cd_ctx *c, int which, uint32_t maxv, uint32_t *out, int nout)
{
    if (!c || !out) return CD_ERR_ARG;
    if (c->stage < ST_BUILT) return CD_ERR_ORDER;
    if (which != 2) { const int rm = materialise_internal_boxes(c); if (rm) return rm; }    // meta / parent / boxes of the reference-shaped tree
    const int n = (int)c->nt;
    hipStream_t s = c->stream;
    HIPCHK(hipEventRecord(c->ev[EV_CHK0], s));
    HIPCHK(hipMemsetAsync(c->d_small + 8, 0, 8 * sizeof(uint32_t), s));           // main.cu:113,121,129
    if (which == 0) { if (n > 1) k_check_internal<<<cdiv(n - 1, 256), 256, 0, s>>>(n, c->d_meta, c->d_parent, c->d_bounded, c->d_boxes, c->d_small + 8); }
    else if (which == 1) k_check_leaves<<<cdiv(n, 256), 256, 0, s>>>(n, c->d_parent, c->d_leaf, maxv, c->d_boxes, c->d_small + 8);
    else k_check_triangle_idx<<<cdiv(n, 256), 256, 0, s>>>(n, c->d_leaf, maxv, c->d_small + 8);
    HIPCHK(hipEventRecord(c->ev[EV_CHK1], s));
    HIPCHK(hipMemcpyAsync(out, c->d_small + 8, nout * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    c->stats.ms_check = elapsed(c, EV_CHK0, EV_CHK1);
    return CD_OK;
}
int cd_check_internal(cd_ctx *c, uint32_t out[5]) { return run_check(c, 0, 0, out, 5); }
int cd_check_leaves(cd_ctx *c, uint32_t out[4]) { return run_check(c, 1, c ? c->nv : 0, out, 4); }
int cd_check_triangle_idx(cd_ctx *c, uint32_t maxv, uint32_t *out) { return run_check(c, 2, maxv, out, 1); }

int cd_find_collisions(cd_ctx *c, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs)
{
    if (!c || (cap_pairs && !pairs)) return CD_ERR_ARG;
    if (c->stage < ST_REFIT) return CD_ERR_ORDER;
    return run_traversal(c, c->tb[0], nullptr, 0, pairs, cap_pairs, n_pairs);
}

static int build_tree_impl(cd_ctx *c, int redo);
int cd_build_tree(cd_ctx *c)
{
    if (!c) return CD_ERR_ARG;
    sort_retry_tick(c);
    return build_tree_impl(c, 0);
}
static int build_tree_impl(cd_ctx *c, int redo)
{
    int rc;
    Prezeroed fused(c);                                                    // one memset for every counter of the pipeline
    rc = enqueue_morton_sort(c, !fused_build_next(c));
    if (!rc) rc = enqueue_tree(c);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->sort_flags, c->d_os_ticket + 8, sizeof c->sort_flags, hipMemcpyDeviceToHost, c->stream));   // words 8..16
    HIPCHK(hipMemcpyAsync(c->root_box_host, c->d_boxes, sizeof(double) * 6, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = judge_sort_flags(c);
    if (rc == SORT_REDO) return redo < SORT_REDO_MAX ? build_tree_impl(c, redo + 1) : CD_ERR_SORT;     // (the form has been changed: judge_sort_flags)
    if (rc) return rc;
    if (c->stage_events) {
        c->stats.ms_morton = elapsed(c, EV_MORTON0, EV_MORTON1);
        c->stats.ms_sort = elapsed(c, EV_MORTON1, EV_SORT1);
        c->stats.ms_hierarchy = c->last_tree_fused ? 0.f : elapsed(c, EV_HIER0, EV_HIER1);
        c->stats.ms_refit = elapsed(c, EV_REFIT0, EV_REFIT1);
    }
    c->root_box_valid = true;
    c->stage = ST_REFIT;
    return CD_OK;
}

static int self_collide_impl(cd_ctx *c, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs, int redo);
int cd_self_collide(cd_ctx *c, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs)
{
    if (!c || (cap_pairs && !pairs)) return CD_ERR_ARG;
    sort_retry_tick(c);                                                    // (before a graph step computes its key)
    return self_collide_impl(c, pairs, cap_pairs, n_pairs, 0);
}
static int self_collide_impl(cd_ctx *c, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs, int redo)
{
    int rc;
    if (graph_eligible(c)) {                                               // CD_OPT_GRAPH: the steady-state step as one graph launch
        bool handled = false;
        rc = graph_step(c, pairs, cap_pairs, n_pairs, handled);
        if (rc < 0 || handled) return rc;
    }
    Prezeroed fused(c);                                                    // one memset for every counter of the pipeline
    rc = enqueue_morton_sort(c, !fused_build_next(c));
    if (!rc) rc = enqueue_tree(c);
    if (!rc) rc = run_traversal(c, c->tb[0], nullptr, 0, pairs, cap_pairs, n_pairs);     // synchronises
    fused.done();
    if (rc < 0) return rc;
    { const int rs = judge_sort_flags(c);                                   // flags came back with the traversal counters
      if (rs == SORT_REDO) return redo < SORT_REDO_MAX ? self_collide_impl(c, pairs, cap_pairs, n_pairs, redo + 1) : CD_ERR_SORT;   // (the form has been changed: judge_sort_flags)
      if (rs) return rs; }
    if (c->stage_events) {
        c->stats.ms_morton = elapsed(c, EV_MORTON0, EV_MORTON1);
        c->stats.ms_sort = elapsed(c, EV_MORTON1, EV_SORT1);
        c->stats.ms_hierarchy = c->last_tree_fused ? 0.f : elapsed(c, EV_HIER0, EV_HIER1);
        c->stats.ms_refit = elapsed(c, EV_REFIT0, EV_REFIT1);
    } else c->stats.ms_morton = c->stats.ms_sort = c->stats.ms_hierarchy = c->stats.ms_refit = 0.f;
    const bool all_stamps = c->stage_events || (c->stamp_mask & 14u) == 14u;
    c->stats.ms_pipeline = all_stamps ? elapsed(c, EV_MORTON0, EV_TRAV1) + (c->stats.ms_traverse - elapsed(c, EV_TRAV0, EV_TRAV1)) : 0.f;   // + deep pass, if any
    c->stats.ms_build_block = (c->last_tree_fused && (c->stamp_mask & 1u)) ? elapsed(c, EV_BLK0, EV_BLK1) : 0.f;
    c->stage = ST_REFIT;
    c->root_box_valid = true;
    return rc;
}

int cd_brute_force(cd_ctx *c, int box_filter, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs)
{
    if (!c || (cap_pairs && !pairs)) return CD_ERR_ARG;
    int rc = ensure_pairs(c, c->tb[0], cap_pairs > 0 ? cap_pairs : 1);
    if (rc) return rc;
    hipStream_t s = c->stream;
    HIPCHK(hipMemsetAsync(c->tb[0].d_state, 0, sizeof(TravState), s));
    k_brute_force<<<cdiv(c->nt, 256), 256, 0, s>>>(c->d_verts, c->d_vidx, c->d_ids, c->nt, box_filter, c->tb[0].d_pairs, cap_pairs, c->tb[0].d_state);
    HostCounters h;
    { int r = read_state(c, c->tb[0], h); if (r) return r; }
    HIPCHK(hipGetLastError());
    const uint64_t ncopy = h.n_pairs < cap_pairs ? h.n_pairs : cap_pairs;
    if (pairs && ncopy) HIPCHK(hipMemcpy(pairs, c->tb[0].d_pairs, sizeof(uint32_t) * 2 * ncopy, hipMemcpyDeviceToHost));
    // the device list now holds THIS call's pairs: cd_sorted_pairs / cd_collision_triangles refer to it
    c->stats.n_pairs = h.n_pairs; c->stats.pairs_tested = h.pairs_tested;
    c->last_pairs_on_device = ncopy;
    if (n_pairs) *n_pairs = h.n_pairs;
    return h.n_pairs > cap_pairs ? CD_OVERFLOW : CD_OK;
}

int cd_test_pairs(cd_ctx *c, const uint32_t *pairs, uint64_t np, uint8_t *out)
{
    if (!c || !pairs || !out) return CD_ERR_ARG;
    if (np == 0) return CD_OK;
    for (uint64_t k = 0; k < 2 * np; ++k) if (pairs[k] >= c->nt) return CD_ERR_INDEX;
    uint32_t *d_p = nullptr; uint8_t *d_o = nullptr;
    HIPCHK(hipMalloc(&d_p, sizeof(uint32_t) * 2 * np));
    if (hipMalloc(&d_o, np) != hipSuccess) { hipFree(d_p); return -(int)hipErrorOutOfMemory; }
    int rc = CD_OK;
    if (hipMemcpy(d_p, pairs, sizeof(uint32_t) * 2 * np, hipMemcpyHostToDevice) != hipSuccess) rc = -(int)hipErrorInvalidValue;
    if (!rc) {
        k_test_pairs<<<cdiv(np, 256), 256, 0, c->stream>>>(c->d_verts, c->d_vidx, c->d_ids, d_p, np, d_o);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, d_o, np, hipMemcpyDeviceToHost) != hipSuccess) rc = -(int)hipGetLastError();
    }
    hipFree(d_p); hipFree(d_o);
    return rc;
}

int cd_sorted_pairs(cd_ctx *c, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs)
{
    if (!c || !n_pairs || (cap_pairs && !pairs)) return CD_ERR_ARG;
    if (c->stage < ST_REFIT) return CD_ERR_ORDER;
    if (c->stats.n_pairs > c->last_pairs_on_device) return CD_OVERFLOW;     // the last traversal's list was truncated: rerun it with a larger cap
    const uint32_t m = (uint32_t)c->last_pairs_on_device;
    *n_pairs = m;
    if (m == 0) return CD_OK;
    int rc = pp_reserve(c, m);
    if (rc) return rc;
    hipStream_t s = c->stream;
    k_pairs_to_keys<<<cdiv(m, 256), 256, 0, s>>>(c->tb[0].d_pairs, m, c->pp_keys[0]);
    if ((rc = pp_sort(c, m))) return rc;
    k_keys_to_pairs<<<cdiv(m, 256), 256, 0, s>>>(c->pp_keys[0], m, reinterpret_cast<uint32_t *>(c->pp_keys[1]));
    const uint64_t ncopy = m < cap_pairs ? m : cap_pairs;
    if (ncopy) HIPCHK(hipMemcpyAsync(pairs, c->pp_keys[1], sizeof(uint32_t) * 2 * ncopy, hipMemcpyDeviceToHost, s));
    uint32_t f[8];
    HIPCHK(hipMemcpyAsync(f, reinterpret_cast<uint32_t *>(c->pp_os) + 8 * RADIX + 8, sizeof f, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    for (int i = 0; i < 8; ++i) if (f[i]) return CD_ERR_SORT;
    return m > cap_pairs ? CD_OVERFLOW : CD_OK;
}

int cd_collision_triangles(cd_ctx *c, uint32_t *ids, uint64_t cap, uint64_t *n)
{
    if (!c || !n || (cap && !ids)) return CD_ERR_ARG;
    if (c->stage < ST_REFIT) return CD_ERR_ORDER;
    if (c->stats.n_pairs > c->last_pairs_on_device) return CD_OVERFLOW;
    const uint64_t m2_64 = 2 * c->last_pairs_on_device;
    *n = 0;
    if (m2_64 == 0) return CD_OK;
    if (m2_64 > 0xfffffff0ull) return CD_ERR_ARG;
    const uint32_t m2 = (uint32_t)m2_64;
    int rc = pp_reserve(c, m2);
    if (rc) return rc;
    hipStream_t s = c->stream;
    k_ids_to_keys<<<cdiv(m2, 256), 256, 0, s>>>(c->tb[0].d_pairs, m2, c->pp_keys[0]);
    if ((rc = pp_sort(c, m2))) return rc;
    k_unique_flags<<<cdiv(m2, 256), 256, 0, s>>>(c->pp_keys[0], m2, c->pp_flags);
    k_scan_exclusive<<<1, 1024, 0, s>>>(c->pp_flags, m2);
    uint32_t *out = c->pp_vals[0], *d_count = reinterpret_cast<uint32_t *>(c->pp_os) + 8 * RADIX;   // ticket word 0 is free after the sort
    k_unique_scatter<<<cdiv(m2, 256), 256, 0, s>>>(c->pp_keys[0], c->pp_flags, m2, out, c->pp_cap, d_count);
    uint32_t cnt = 0, f[8];
    HIPCHK(hipMemcpyAsync(&cnt, d_count, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(f, reinterpret_cast<uint32_t *>(c->pp_os) + 8 * RADIX + 8, sizeof f, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    for (int i = 0; i < 8; ++i) if (f[i]) return CD_ERR_SORT;
    *n = cnt;
    const uint64_t ncopy = cnt < cap ? cnt : cap;
    if (ncopy) HIPCHK(hipMemcpy(ids, out, sizeof(uint32_t) * ncopy, hipMemcpyDeviceToHost));
    return cnt > cap ? CD_OVERFLOW : CD_OK;
}

int cd_export_keys(cd_ctx *c, uint64_t *keys, uint32_t *perm)
{
    if (!c) return CD_ERR_ARG;
    if (c->stage < ST_SORTED) return CD_ERR_ORDER;
    if (keys) HIPCHK(hipMemcpy(keys, c->d_keys[0], sizeof(uint64_t) * c->nt, hipMemcpyDeviceToHost));
    if (perm) HIPCHK(hipMemcpy(perm, c->d_perm[0], sizeof(uint32_t) * c->nt, hipMemcpyDeviceToHost));
    return CD_OK;
}

int cd_export_tree(cd_ctx *c, int32_t *parent, int32_t *left, int32_t *right, double *boxes, uint32_t *bounded)
{
    if (!c) return CD_ERR_ARG;
    if (c->stage < ST_BUILT) return CD_ERR_ORDER;
    const size_t n = c->nt;
    { const int rm = materialise_internal_boxes(c); if (rm) return rm; }      // after a fused call: the reference-shaped tree is built on request
    if (parent) HIPCHK(hipMemcpy(parent, c->d_parent, sizeof(int32_t) * (2 * n - 1), hipMemcpyDeviceToHost));
    if ((left || right) && n > 1) {
        std::vector<int4> m(n - 1);
        HIPCHK(hipMemcpy(m.data(), c->d_meta, sizeof(int4) * (n - 1), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n - 1; ++i) { if (left) left[i] = m[i].x; if (right) right[i] = m[i].y; }
    }
    if (boxes || bounded) { if (c->stage < ST_REFIT) return CD_ERR_ORDER; }

    if (boxes) HIPCHK(hipMemcpy(boxes, c->d_boxes, sizeof(double) * 6 * (2 * n - 1), hipMemcpyDeviceToHost));
    if (bounded && n > 1) HIPCHK(hipMemcpy(bounded, c->d_bounded, sizeof(uint32_t) * (n - 1), hipMemcpyDeviceToHost));
    return CD_OK;
}

/* Debug only (not in the public header): sums of the 12 spare words of the 64 counter shards of the last traversal. */
int cd_debug_counters(cd_ctx *c, unsigned long long out[12])
{
    if (!c || !out) return CD_ERR_ARG;
    std::vector<TravState> h(1);
    HIPCHK(hipMemcpy(h.data(), c->tb[0].d_state, sizeof(TravState), hipMemcpyDeviceToHost));
    for (int k = 0; k < 12; ++k) { out[k] = 0; for (int i = 0; i < NSHARD; ++i) out[k] += h[0].shard[i].pad[k]; }
    return CD_OK;
}

/* Debug only: the order hint's arrays -- per group of 64 leaves (ceil(nt / 64) words each) the score of the last fused build and the order made from the scores; per triangle (nt bytes) the time class its wave left in the last half traversal. */
int cd_debug_hint(cd_ctx *c, uint32_t *cost, uint32_t *order, uint8_t *tri)
{
    if (!c) return CD_ERR_ARG;
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t groups = ((size_t)c->nt + 63) / 64;
    if (tri) HIPCHK(hipMemcpy(tri, c->d_tri_cost, c->nt, hipMemcpyDeviceToHost));
    if (cost) HIPCHK(hipMemcpy(cost, c->d_cost, sizeof(uint32_t) * groups, hipMemcpyDeviceToHost));
    if (order) { if (!c->order_ready) return CD_ERR_ORDER; HIPCHK(hipMemcpy(order, c->d_order, sizeof(uint32_t) * groups, hipMemcpyDeviceToHost)); }
    return CD_OK;
}

/* Debug only: install an order for the next half traversal (must be a permutation of the groups; checked). */
int cd_debug_hint_set(cd_ctx *c, const uint32_t *order)
{
    if (!c || !order) return CD_ERR_ARG;
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t groups = (c->nt + 63u) / 64u;
    std::vector<uint8_t> seen(groups, 0);
    for (uint32_t b = 0; b < groups; ++b) { if (order[b] >= groups || seen[order[b]]) return CD_ERR_ARG; seen[order[b]] = 1; }
    HIPCHK(hipMemcpy(c->d_order, order, sizeof(uint32_t) * groups, hipMemcpyHostToDevice));
    c->order_ready = true;
    return CD_OK;
}

int cd_debug_records(cd_ctx *c, void *recs, void *qboxes, int32_t *root)
{
    if (!c) return CD_ERR_ARG;
    if (c->stage < ST_REFIT) return CD_ERR_ORDER;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (recs) HIPCHK(hipMemcpy(recs, c->d_recs32, sizeof(NodeRec32) * (size_t)c->nt, hipMemcpyDeviceToHost));
    if (qboxes) { const int rq = ensure_qbox(c); if (rq) return rq; HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipMemcpy(qboxes, c->d_qbox, sizeof(LeafBox32) * (size_t)c->nt, hipMemcpyDeviceToHost)); }
    if (root) HIPCHK(hipMemcpy(root, c->d_root, sizeof(int32_t), hipMemcpyDeviceToHost));
    return CD_OK;
}

// morton3D / expand64Bits themselves, on caller-supplied inputs; no context (one-shot device buffers on the null stream)
static int morton_batch(const void *in, size_t in_bytes, uint64_t n, const double frame[FRAME_WORDS], uint64_t *out)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return CD_ERR_NO_DEVICE;
    void *d_in = nullptr; uint64_t *d_out = nullptr; double *d_frame = nullptr;
    int rc = CD_OK;
    hipError_t e = hipMalloc(&d_in, in_bytes);
    if (e == hipSuccess) e = hipMalloc(&d_out, sizeof(uint64_t) * n);
    if (e == hipSuccess && frame) e = hipMalloc(&d_frame, sizeof(double) * FRAME_WORDS);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, in_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && frame) e = hipMemcpy(d_frame, frame, sizeof(double) * FRAME_WORDS, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const uint32_t blocks = cdiv(n, 256) < 4096u ? cdiv(n, 256) : 4096u;
        if (frame) k_morton_points<<<blocks, 256>>>(static_cast<const double *>(d_in), n, d_frame, d_out);
        else k_expand_values<<<blocks, 256>>>(static_cast<const uint64_t *>(d_in), n, d_out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(out, d_out, sizeof(uint64_t) * n, hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) rc = -(int)e;
    hipFree(d_in); hipFree(d_out); hipFree(d_frame);
    return rc;
}
int cd_morton3d_points(const double *xyz, uint64_t n, const double offset[3], const double span[3], uint64_t *keys)
{
    if (!xyz || !keys || (!offset) != (!span)) return CD_ERR_ARG;
    if (n == 0) return CD_OK;
    double frame[FRAME_WORDS] = {0.004501, -0.476622, -0.381965, 3.08, 0.76, 2.36, 0.0, 0.0};           // morton.h:45,51,57
    if (offset) for (int a = 0; a < 3; ++a) { frame[a] = offset[a]; frame[3 + a] = span[a]; }
    return morton_batch(xyz, sizeof(double) * 3 * n, n, frame, keys);
}
// the same in a frame with a key layout (cd_math.h morton3d_layout: what k_morton computes in such a frame -- xyz is then the SUM of a triangle's three vertices
// per axis, not the centroid: cd_math.h, KeyLayout); layout 0 = cd_morton3d_points
int cd_morton3d_points_layout(const double *xyz, uint64_t n, const double offset[3], const double span[3], uint64_t layout, uint64_t *keys)
{
    if (!xyz || !keys || !offset || !span || !layout_ok(layout)) return CD_ERR_ARG;
    if (n == 0) return CD_OK;
    double frame[FRAME_WORDS] = {offset[0], offset[1], offset[2], span[0], span[1], span[2], 0.0, 0.0};
    std::memcpy(&frame[6], &layout, sizeof layout);
    return morton_batch(xyz, sizeof(double) * 3 * n, n, frame, keys);
}
int cd_expand64_values(const uint64_t *v, uint64_t n, uint64_t *out)
{
    if (!v || !out) return CD_ERR_ARG;
    if (n == 0) return CD_OK;
    return morton_batch(v, sizeof(uint64_t) * n, n, nullptr, out);
}

// checkBoxOverlap / Box::merge / checkTriangleContact themselves on caller-supplied operands; no context (one-shot buffers, null stream)
int cd_box_pairs(const double *a, const double *b, uint64_t n, uint8_t *overlap, double *merged)
{
    if (!a || !b || (!overlap && !merged)) return CD_ERR_ARG;
    if (n == 0) return CD_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return CD_ERR_NO_DEVICE;
    double *d_a = nullptr, *d_b = nullptr, *d_m = nullptr; uint8_t *d_o = nullptr;
    const size_t bytes = sizeof(double) * 6 * n;
    hipError_t e = hipMalloc(&d_a, bytes);
    if (e == hipSuccess) e = hipMalloc(&d_b, bytes);
    if (e == hipSuccess && overlap) e = hipMalloc(&d_o, n);
    if (e == hipSuccess && merged) e = hipMalloc(&d_m, bytes);
    if (e == hipSuccess) e = hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_box_pairs<<<cdiv(n, 256) < 4096u ? cdiv(n, 256) : 4096u, 256>>>(d_a, d_b, n, d_o, d_m);
        e = hipGetLastError();
        if (e == hipSuccess && overlap) e = hipMemcpy(overlap, d_o, n, hipMemcpyDeviceToHost);
        if (e == hipSuccess && merged) e = hipMemcpy(merged, d_m, bytes, hipMemcpyDeviceToHost);
    }
    hipFree(d_a); hipFree(d_b); hipFree(d_o); hipFree(d_m);
    return e == hipSuccess ? CD_OK : -(int)e;
}
int cd_tri_contact_points(const double *tri, uint64_t n, uint8_t *out)
{
    if (!tri || !out) return CD_ERR_ARG;
    if (n == 0) return CD_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return CD_ERR_NO_DEVICE;
    double *d_t = nullptr; uint8_t *d_o = nullptr;
    hipError_t e = hipMalloc(&d_t, sizeof(double) * 18 * n);
    if (e == hipSuccess) e = hipMalloc(&d_o, n);
    if (e == hipSuccess) e = hipMemcpy(d_t, tri, sizeof(double) * 18 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_tri_contact_points<<<cdiv(n, 256) < 4096u ? cdiv(n, 256) : 4096u, 256>>>(d_t, n, d_o);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(out, d_o, n, hipMemcpyDeviceToHost);
    }
    hipFree(d_t); hipFree(d_o);
    return e == hipSuccess ? CD_OK : -(int)e;
}

int cd_alloc_host_pairs(uint64_t cap_pairs, uint32_t **pairs)
{
    if (!pairs || cap_pairs == 0) return CD_ERR_ARG;
    *pairs = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return CD_ERR_NO_DEVICE;
    void *p = nullptr;
    // (coherent = fine-grained: the memory the HSA memory model defines system-scope release / acquire between device and host on)
    HIPCHK(hipHostMalloc(&p, sizeof(uint32_t) * 2 * cap_pairs + 16, hipHostMallocCoherent));     // +16: the report kernel moves pairs as 16-byte quads
    PinnedPairs &pp = pinned_pairs();
    { std::lock_guard<std::mutex> lock(pp.mu); pp.v.push_back(PinnedPairs::Buf{static_cast<const uint32_t *>(p), cap_pairs, pp.next_id++}); }
    *pairs = static_cast<uint32_t *>(p);
    return CD_OK;
}
void cd_free_host_pairs(uint32_t *pairs)
{
    if (!pairs) return;
    PinnedPairs &pp = pinned_pairs();
    { std::lock_guard<std::mutex> lock(pp.mu);
      for (size_t i = 0; i < pp.v.size(); ++i) if (pp.v[i].p == pairs) { pp.v.erase(pp.v.begin() + (long)i); break; } }
    hipHostFree(pairs);
}


int cd_get_stats(cd_ctx *c, cd_stats *out) { if (!c || !out) return CD_ERR_ARG; *out = c->stats; return CD_OK; }
int cd_num_triangles(cd_ctx *c, uint32_t *nt) { if (!c || !nt) return CD_ERR_ARG; *nt = c->nt; return CD_OK; }

int cd_set_option(cd_ctx *c, int key, int64_t value)
{
    if (!c) return CD_ERR_ARG;
    if (key == CD_OPT_TRAVERSAL) { if (value != 0 && value != 1 && value != 3) return CD_ERR_ARG; c->trav_variant = (int)value; return CD_OK; }
    if (key == CD_OPT_QUERIES_PER_WAVE) { if (value < 64 || value > (1 << 20) || value % 64) return CD_ERR_ARG; c->queries_per_wave = (uint32_t)value; return CD_OK; }
    if (key == CD_OPT_SORT_FULL) { if (value < 0 || value > 2) return CD_ERR_ARG; c->sort_mode = value == 0 ? 0 : (value == 1 ? 3 : 2); c->left_frame = false; return CD_OK; }
    if (key == CD_OPT_STAGE_TIMING) { c->stage_events = value != 0; return CD_OK; }
    if (key == CD_OPT_GRAPH) { c->graph_opt = value != 0; if (!c->graph_opt) graph_drop(c); return CD_OK; }
    if (key == CD_OPT_POLL) { c->poll_opt = value != 0; return CD_OK; }
    if (key == CD_OPT_CELL_TABLE) {
        if (c->cell_table_opt == (value != 0)) return CD_OK;
        c->cell_table_opt = value != 0;
        HIPCHK(hipStreamSynchronize(c->stream));
        c->stage = ST_CREATED; c->root_box_valid = false;                  // the records of the tree that is there were encoded the other way
        return amb_refresh(c);
    }
    if (key == CD_OPT_ORDER_HINT) { c->order_hint = value != 0; c->order_hint_large = value == 2; c->order_ready = false; c->order_pending = false; graph_drop(c); return CD_OK; }       // (a captured step has its launches baked in)
    if (key == CD_OPT_KERNEL_STAMPS) { if (value < 0 || value > 15) return CD_ERR_ARG; c->stamp_mask = (uint32_t)value; return CD_OK; }
    return CD_ERR_ARG;
}

// Measurement hooks and test switches: their own entry point and their own key space, so that no hook can shadow an option
// (round 3: the polling scan's keys were once numbered like the keys the record tests use to ask for the stage-wise build).
// Setters return CD_OK; getters write *out (required) and ignore `value`.
int cd_debug_option(cd_ctx *c, int key, int64_t value, int64_t *out)
{
    if (!c) return CD_ERR_ARG;
    switch (key) {
    case CD_DBG_LDS_PAD:         c->dbg_lds_pad = (uint32_t)value; return CD_OK;
    case CD_DBG_EXACT_BLOCKS:    if (value < 1 || value > 65535) return CD_ERR_ARG; c->exact_blocks = (int)((value + NSHARD - 1) / NSHARD * NSHARD); return CD_OK;   // (a workgroup owns a shard: whole multiples of 64)
    case CD_DBG_NO_SHARED_PATH:  c->dbg_no_shared_path = value != 0; return CD_OK;
    case CD_DBG_DIAG:            c->dbg_diag = value != 0; return CD_OK;
    case CD_DBG_STAGEWISE_BUILD: c->dbg_no_fused_build = value != 0; return CD_OK;
    case CD_DBG_SPLIT_CROSS:     c->dbg_split_cross = value != 0; return CD_OK;
    case CD_DBG_SORT_WINDOWS:    if (value < 0 || value > 2) return CD_ERR_ARG; c->dbg_sort_windows = (uint32_t)value; c->local_small_ok = true; graph_drop(c); return CD_OK;
    case CD_DBG_REPORT_COPIES:   c->dbg_report_copies = value != 0; graph_drop(c); return CD_OK;
    case CD_DBG_STORE_QBOX:      c->dbg_store_qbox = value != 0; graph_drop(c); return CD_OK;
    case CD_DBG_BIG_OFFSETS:     c->dbg_big_offsets = value != 0; graph_drop(c); return CD_OK;
    case CD_DBG_POLL_SCAN:       c->dbg_poll_check = value != 0; return CD_OK;
    case CD_DBG_GET_POLL_STALE:     if (!out) return CD_ERR_ARG; *out = c->poll_stale; return CD_OK;
    case CD_DBG_GET_POLL_FALLBACKS: if (!out) return CD_ERR_ARG; *out = c->poll_fallbacks; return CD_OK;
    case CD_DBG_GET_POLLED_STEPS:   if (!out) return CD_ERR_ARG; *out = c->polled_steps; return CD_OK;
    case CD_DBG_GET_POLL_FB_WHY:    if (!out) return CD_ERR_ARG; *out = (int64_t)c->poll_fb_busy | ((int64_t)c->poll_fb_late_word << 16) | ((int64_t)c->poll_fb_lost << 32); return CD_OK;
    case CD_DBG_GET_POLL_MAX_WAIT_US: if (!out) return CD_ERR_ARG; *out = c->poll_max_wait_us; return CD_OK;
    case CD_DBG_GET_TREE_WAS_FUSED: if (!out) return CD_ERR_ARG; *out = c->last_tree_fused ? 1 : 0; return CD_OK;
    case CD_DBG_GET_SORT_FORM:   if (out) *out = c->sort_mode; return CD_OK;
    case CD_DBG_GET_ORDER_STATE: {          // the order hint as it stands: 0 none built, 1 a permutation of the groups that differs from the plain order, 2 the plain order itself, -1 NOT a permutation (a bug)
        if (!out) return CD_ERR_ARG;
        *out = 0;
        if (!c->order_ready) return CD_OK;
        HIPCHK(hipStreamSynchronize(c->stream));
        const uint32_t groups = (c->nt + 63u) / 64u;
        std::vector<uint32_t> o(groups), plain(groups);
        HIPCHK(hipMemcpy(o.data(), c->d_order, sizeof(uint32_t) * groups, hipMemcpyDeviceToHost));
        std::vector<uint8_t> seen(groups, 0);
        bool perm = true, same = true;
        for (uint32_t b = 0; b < groups; ++b) {
            if (o[b] >= groups || seen[o[b]]) { perm = false; break; }
            seen[o[b]] = 1;
            same = same && o[b] == half_vblock_host(b, groups);
        }
        *out = !perm ? -1 : (same ? 2 : 1);
        return CD_OK;
    }
    default: return CD_ERR_ARG;
    }
}

int cd_set_vertex_id_base(cd_ctx *c, uint32_t base) { if (!c) return CD_ERR_ARG; c->vbase = base; return CD_OK; }

int cd_root_box(cd_ctx *c, double box[6])
{
    if (!c || !box) return CD_ERR_ARG;
    if (c->stage < ST_REFIT) return CD_ERR_ORDER;
    if (c->root_box_valid) { memcpy(box, c->root_box_host, sizeof(double) * 6); return CD_OK; }   // came back with an earlier read-back
    // n == 1: the only node is leaf 0 = node 0 as well
    HIPCHK(hipMemcpy(box, c->d_boxes, sizeof(double) * 6, hipMemcpyDeviceToHost));
    return CD_OK;
}

int cd_pack_queries(cd_ctx *c, const double box[6], void *d_out, uint64_t cap, uint64_t *n)
{
    if (!c || !box || !n || (cap && !d_out)) return CD_ERR_ARG;
    if (c->stage < ST_REFIT) return CD_ERR_ORDER;
    hipStream_t s = c->stream;
    // the box and the counter live in the context's small scratch words (d_small[104..117]), not in the traversal state:
    // packing does not disturb the statistics of the traversal before it
    double *d_box = reinterpret_cast<double *>(c->d_small + 104);
    unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(c->d_small + 116);
    { const int rq = ensure_qbox(c); if (rq) return rq; }
    HIPCHK(hipMemcpyAsync(d_box, box, sizeof(double) * 6, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), s));
    k_pack_queries<<<cdiv(c->nt, PACK_THREADS), PACK_THREADS, 0, s>>>(c->d_verts, c->d_leaf, c->d_boxes, c->d_qbox, (int)c->nt, d_box, 1, -1, nullptr,
                                                                       reinterpret_cast<ExtQuery *>(d_out), cap, d_cnt, c->vbase);
    unsigned long long h = 0;
    HIPCHK(hipMemcpyAsync(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    *n = h;
    return h > cap ? CD_OVERFLOW : CD_OK;
}

int cd_find_collisions_queries(cd_ctx *c, const void *d_queries, uint64_t nq, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs)
{
    if (!c || (cap_pairs && !pairs) || (nq && !d_queries) || nq > 0xffffffffull) return CD_ERR_ARG;
    if (c->stage < ST_REFIT) return CD_ERR_ORDER;
    if (nq == 0) { if (n_pairs) *n_pairs = 0; return CD_OK; }
    return run_traversal(c, c->tb[0], d_queries, nq, pairs, cap_pairs, n_pairs);
}

}  // extern "C"

#include "cd_multi.h"
