/*
 * mi355cd.h -- C ABI of libmi355cd.so: MI355X-native (HIP, gfx950) triangle-mesh collision detection.
 *
 * Drop-in boundary for the CollisionDetection hot path of Asichurter/GPU-Computing-Course.  The
 * reference has no FFI; its boundary is the sequence of kernel call sites in CollisionDetection/main.cu.
 * Each export below names the call site (reference file:line) it replaces.  Host pointers in, the
 * library owns all device memory behind an opaque context; plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns int: 0 = ok; <0 = -(hipError_t) or one of CD_ERR_*; >0 = CD_OVERFLOW
 *     (the reference prints and exits via HANDLE_ERROR, common/book.h:21-30; this library never exits).
 *   - one context per device; a context is not thread-safe; different contexts are independent.
 *   - all calls block until the stage has finished (the reference synchronises after every kernel,
 *     main.cu:93,100,109,...).  Per-stage device times are measured with HIP events on the
 *     context's own stream and read back through cd_get_stats().
 *   - node ids: internal node i -> i (root = 0, main.cu:142 passes &internal_nodes[0]);
 *     leaf j (Morton-sorted position) -> (n-1)+j.  -1 = NULL.
 *   - boxes are {x1,x2,y1,y2,z1,z2} doubles, the field order of box.cuh:9.
 */
#ifndef MI355CD_H
#define MI355CD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cd_ctx cd_ctx;

enum {
    CD_OK            = 0,
    CD_OVERFLOW      = 1,      /* more pairs than cap_pairs; *n_pairs holds the true count          */
    CD_ERR_ARG       = -1001,  /* null / zero-sized / inconsistent argument                         */
    CD_ERR_ORDER     = -1002,  /* stage called before the stage it depends on                       */
    CD_ERR_NO_DEVICE = -1003,  /* no HIP device: the library has no CPU fallback                    */
    CD_ERR_INDEX     = -1004,  /* a vertex index >= nv (checked at cd_create)                       */
    CD_ERR_SORT      = -1005,  /* a bounded device-side wait in the sort timed out (results invalid) */
    CD_ERR_IO        = -1006,  /* file cannot be opened / read                                      */
    CD_ERR_FORMAT    = -1007,  /* a `v` / `f` line is not in the reference's dialect, or no geometry */
    CD_ERR_RCCL      = -1008,  /* librccl could not be loaded, or an RCCL call failed (cd_multi_*)   */
    CD_ERR_PEER      = -1009,  /* cd_multi_*: ANOTHER rank failed in this step; every rank returned in the same step (that rank its own error) */
    CD_ERR_INJECTED  = -1010   /* cd_multi_step: the failure asked for with CD_MULTI_INJECT_FAILURE (tests)  */
};

/* Morton normalisation frame (morton.h:43-58 hard-codes one data set's bounds). */
enum {
    CD_FRAME_REFERENCE = 0,    /* the constants of morton.h:45,51,57 -> keys bit-identical to morton3D */
    CD_FRAME_AUTO      = 1,    /* from the mesh, computed on the device in every step: AABB of the centroids AND the
                                  key layout that suits it (below: "the adaptive frame")                      */
    CD_FRAME_CUSTOM    = 2     /* caller-supplied offset[3], span[3]                                   */
};

typedef struct cd_stats {
    /* device time of the last run of each stage, milliseconds (HIP events on the context stream) */
    float ms_morton;           /* centroid + Morton keys                    (load_obj.h:89-101)      */
    float ms_sort;             /* radix sort by key                         (load_obj.h:107)         */
    float ms_hierarchy;        /* leaf fill + Karras hierarchy              (main.cu:92,99)          */
    float ms_refit;            /* bottom-up AABB refit                      (main.cu:107)            */
    float ms_traverse;         /* traversal + exact test                    (main.cu:142)            */
    float ms_check;            /* verifier kernels                          (main.cu:115,123,131)    */
    uint32_t traverse_launches;/* kernel launches inside the last traversal (0: the step was a graph replay, CD_OPT_GRAPH) */
    uint32_t stack_overflows;  /* queries that needed the deep-stack fallback in the last traversal */
    uint64_t n_pairs;          /* contacts found by the last traversal      (main.cu:145 test_val)   */
    uint64_t pairs_tested;     /* (query, leaf) pairs with strictly overlapping AABBs                */
    uint64_t node_visits;      /* internal nodes visited                                             */
    uint64_t wave_steps;       /* descent-loop iterations summed over wavefronts (lane utilisation =  */
                               /* node_visits / (64 * wave_steps)); 0 for CD_OPT_TRAVERSAL 0           */
    uint64_t candidates;       /* (query, leaf) candidates the fp32 descent handed to the exact kernel */
    float ms_descend;          /* shallow pass: memset + descent kernel (part of ms_traverse)          */
    float ms_exact;            /* shallow pass: exact-test kernel        (part of ms_traverse)          */
    uint32_t sort_passes;      /* global digit passes of the last sort: 2 (hybrid), 4 (half-key) or 8     */
    float ms_pipeline;         /* fused calls: pipeline start -> end of the traversal kernels, one event pair */
    float ms_build_block;      /* fused calls: the kernel that builds hierarchy + boxes + records of the 512-leaf     */
                               /* blocks (k_refit_seg_local<fused>), from its own dispatch packet; part of ms_refit  */
    float ms_descend_clock;    /* the descent kernel (CD_OPT_TRAVERSAL 3) timed by ITSELF: first wave start -> last wave end on the    */
                               /* device's constant-rate wall clock (s_memrealtime, hipDeviceAttributeWallClockRate).  Taken in every */
                               /* call at no cost; slightly below ms_descend (the dispatch packet's stamps also cover launch and the  */
                               /* end-of-kernel write-back).  0 when the traversal needed a deep pass or another variant ran           */
} cd_stats;

/* main.cu:64 loadObj (load_obj.h:24-103), host side, multi-threaded: parse `v x y z` (as float, widened to double)
 * and `f a/ta b/tb c/tc` (1-based) lines in file order.  Unlike the reference it does not compute Morton codes
 * or sort (cd_morton_sort does, on the GPU) and it returns an error instead of exiting.  The arrays are
 * malloc'ed; release them with cd_free_obj.  threads <= 0: one per hardware thread.  Needs no GPU. */
int cd_load_obj(const char *path, double **verts_xyz, uint32_t *nv, uint32_t **vidx3, uint32_t *nt, int threads);
void cd_free_obj(double *verts_xyz, uint32_t *vidx3);

/* main.cu:78-88  cudaMalloc + cudaMemcpy of vec3f[V], Triangle[N], u64[N], Node[N], Node[N-1].
 * verts_xyz: nv x 3 doubles (vec3f.cuh:14-23).  vidx3: nt x 3 vertex indices (triangle.cuh:9).
 * ids: nt triangle IDs (triangle.cuh:6; load_obj.h:94 uses the face ordinal) or NULL for 0..nt-1. */
int cd_create(cd_ctx **out, const double *verts_xyz, uint32_t nv,
              const uint32_t *vidx3, const uint32_t *ids, uint32_t nt);
void cd_destroy(cd_ctx *ctx);                                               /* main.cu:156-163 */

/* Replace vertex positions (same nv, same topology): the per-frame re-run of a cloth simulation. */
int cd_update_vertices(cd_ctx *ctx, const double *verts_xyz);

/* morton.h:43-58: choose the normalisation frame (default CD_FRAME_REFERENCE). offset/span are
 * read only for CD_FRAME_CUSTOM.  REFERENCE and CUSTOM interleave 20 bits an axis x, y, z exactly as morton.h:70-89.
 *
 * The adaptive frame (CD_FRAME_AUTO; not reference behaviour -- morton.h has its constants and nothing else).  For a mesh
 * those constants do not fit, the library takes offset / span from the bounds of the centroids and DEALS the 60 key bits to
 * the axes so that the cells of every tree level are near cubes in units of the triangles' own mean extent per axis (a thin,
 * long mesh normalised per axis with the fixed interleave gets cells of 400 : 1 and a poor tree).  The pair set does not
 * depend on the keys (any correct BVH gives the reference's set); the tree's cost does.  A layout is one 64-bit word:
 * bit 63 set | A | B << 2 | C << 4 | nA << 8 | nAB << 16 | nABC << 24 -- axes A, B, C (0 = x, 1 = y, 2 = z, by decreasing
 * weight); the key is, from its top bit down, nA bits of A's cell index, nAB pairs (A, B), nABC triples (A, B, C); a cell index along
 * an axis with b bits is floor(((p1 + p2 + p3) - 3 offset) * (2^b / (3 span))) clamped to [0, 2^b - 1];
 * nA + 2 nAB + 3 nABC <= 60.  0 = the reference's interleave.  Derivation and numbers: csrc/cd_math.h, DESIGN.md. */
int cd_set_morton_frame(cd_ctx *ctx, int mode, const double offset[3], const double span[3]);
/* The frame the last sort used -- offset, span, key layout (any may be NULL) -- and a frame WITH a layout installed as
 * CD_FRAME_CUSTOM: a frame CD_FRAME_AUTO computed once and the caller keeps (the AUTO pass over the triangles, ~11 us at
 * 1 M, leaves the step; a centroid that later leaves the frame takes the last cell of its axis), or one frame for all
 * the ranks of a job.  CD_ERR_ORDER before the first sort; CD_ERR_ARG for a word that is not a layout. */
int cd_get_morton_frame(cd_ctx *ctx, double offset[3], double span[3], uint64_t *layout);
int cd_set_morton_frame_layout(cd_ctx *ctx, const double offset[3], const double span[3], uint64_t layout);

/* morton.h:70-89 morton3D(x, y, z) and morton.h:7-29 expand64Bits(v) themselves, on n caller-supplied inputs (host
 * pointers; no context): the device functions cd_morton_sort uses, exposed so that the reference's own functions can be
 * compared value by value (tests/golden/morton_ref.npz holds outputs of the reference's morton.h compiled unmodified).
 * offset / span: both NULL = the constants of morton.h:45,51,57, else a custom frame.  Defined where the reference is
 * undefined: a negative or NaN normalised coordinate maps to cell 0 (morton.h:78's assert is compiled out in Release). */
int cd_morton3d_points(const double *xyz, uint64_t n, const double offset[3], const double span[3], uint64_t *keys);
/* the same in a frame with a key layout: what cd_morton_sort computes per triangle in such a frame, where xyz is the SUM p1 + p2 + p3 of the triangle's vertices per
 * axis (a cell there is floor((sum - 3 offset) * (2^bits / (3 span))), no division per key: csrc/cd_math.h); layout 0 = the call above, xyz the centroid */
int cd_morton3d_points_layout(const double *xyz, uint64_t n, const double offset[3], const double span[3], uint64_t layout, uint64_t *keys);
int cd_expand64_values(const uint64_t *v, uint64_t n, uint64_t *out);

/* box.cuh:40-43 checkBoxOverlap(a, b), box.cuh:24-32 Box::merge(a, b) and tri_contact.cuh:19-78 checkTriangleContact themselves, on n
 * caller-supplied operands (host pointers; no context), for the same purpose: tests/golden/contact_ref.npz holds outputs of the
 * reference's box.cuh / tri_contact.cuh compiled unmodified.  Boxes are {x1,x2,y1,y2,z1,z2} (box.cuh:9); overlap[k] = 0/1 and
 * merged (n x 6) may each be NULL (not both).  tri: n x 18 doubles = P1 P2 P3 Q1 Q2 Q3 positions; out[k] = 0/1 -- no ID rule and
 * no neighbour gate (those are cd_test_pairs). */
int cd_box_pairs(const double *a, const double *b, uint64_t n, uint8_t *overlap, double *merged);
int cd_tri_contact_points(const double *tri, uint64_t n, uint8_t *out);

/* load_obj.h:89-107: centroid + morton3D per face, then sort_by_key(mortons, triangles) -- on the GPU. */
int cd_morton_sort(cd_ctx *ctx);

/* main.cu:92 fillLeafNodes + main.cu:99 generateHierarchyParallel.  *parent_wrong_num is the counter
 * printed at main.cu:103 (children that already had a parent; 0 for a correct tree). May be NULL. */
int cd_build_hierarchy(cd_ctx *ctx, uint32_t *parent_wrong_num);

/* main.cu:107 calBoundingBox: leaf boxes (box.cuh:13-22) and bottom-up merge (box.cuh:24-32). */
int cd_refit_boxes(cd_ctx *ctx);

/* main.cu:115 checkInternalNodes: out = {nullParentNum, wrongBoundNum, nullChildNum, notInternalCount,
 * uninitBoxCount} in the order printed at main.cu:119. */
int cd_check_internal(cd_ctx *ctx, uint32_t out[5]);
/* main.cu:123 checkLeafNodes: out = {nullParentNum, nullTriangleNum, notLeafCount, illegalBoxCount}
 * (main.cu:127).  Triangle::selfCheck's hard-coded 632674 (triangle.cuh:13) is the context's nv. */
int cd_check_leaves(cd_ctx *ctx, uint32_t out[4]);
/* main.cu:131 checkTriangleIdx(leaves, vs, n, maxv, count). */
int cd_check_triangle_idx(cd_ctx *ctx, uint32_t maxv, uint32_t *out);

/* main.cu:142-146 findCollisions + D2H of count and pair list.  pairs: cap_pairs x 2 uint32,
 * interleaved (smaller ID, larger ID), unordered (atomicAdd append, collision.cuh:40-42).
 * pairs may be NULL with cap_pairs 0 (count only).  Returns CD_OVERFLOW when *n_pairs > cap_pairs
 * (the reference writes past its 500-pair buffer instead, main.cu:81). */
int cd_find_collisions(cd_ctx *ctx, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs);

/* main.cu:74,81 (the reference's colTris host array + cudaMalloc'd copy): an output buffer for cap_pairs pairs in PINNED host memory.
 * Optional: every entry point that returns pairs takes any host pointer; handed THIS buffer (the pointer as returned, cap_pairs up to
 * its capacity), cd_find_collisions / cd_self_collide let the GPU write the pairs straight into it -- no staging copy on the host
 * (~4 us of a 0.24 ms step at 20 k pairs).  Release with cd_free_host_pairs, after the last call that uses it. */
int cd_alloc_host_pairs(uint64_t cap_pairs, uint32_t **pairs);
void cd_free_host_pairs(uint32_t *pairs);

/* cd_morton_sort -> cd_build_hierarchy -> cd_refit_boxes queued back to back, one host synchronisation: the tree
 * without the traversal (the multi-GPU step exchanges query leaves while the local traversal runs). */
int cd_build_tree(cd_ctx *ctx);

/* Fused convenience call: cd_morton_sort -> cd_build_hierarchy -> cd_refit_boxes -> cd_find_collisions
 * queued back to back on the context stream with a single host synchronisation at the end. */
int cd_self_collide(cd_ctx *ctx, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs);

/* main.cu:149-154: the pair list of the LAST traversal, sorted ascending by (smaller ID, larger ID) on the device --
 * a deterministic order for diffing (the reference prints in atomicAdd arrival order).  Needs the last
 * cd_find_collisions / cd_self_collide to have had cap_pairs >= its n_pairs (else CD_OVERFLOW). */
int cd_sorted_pairs(cd_ctx *ctx, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs);
/* main.cu:33-45 makeAndPrintSet: the sorted set of distinct triangle IDs that occur in the pair list,
 * built on the device (sort + unique).  *n = number of distinct IDs (may exceed cap -> CD_OVERFLOW). */
int cd_collision_triangles(cd_ctx *ctx, uint32_t *ids, uint64_t cap, uint64_t *n);

/* check.cuh:117-141 checkDirectComp: O(N^2) all-pairs on the device, no tree.  box_filter != 0 also
 * requires the strict leaf-AABB overlap the BVH path applies (collision.cuh:31-36). Same output format
 * as cd_find_collisions. */
int cd_brute_force(cd_ctx *ctx, int box_filter, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs);

/* tri_contact.cuh:80-87 checkTriangleContactHelper over an explicit list of triangle-index pairs
 * (original triangle order), including the neighborCount<1 gate of collision.cuh:38.  out[k] = 0/1. */
int cd_test_pairs(cd_ctx *ctx, const uint32_t *pairs, uint64_t n_pairs, uint8_t *out);

/* Read-back for verification (the reference's tree is device pointers and cannot be exported). Any
 * pointer may be NULL.  keys/perm: nt entries, sorted order (perm[j] = original triangle of leaf j).
 * parent: 2nt-1; left/right: nt-1; boxes: (2nt-1) x 6; bounded: nt-1 (Node::bounded, bvh.cuh:28). */
int cd_export_keys(cd_ctx *ctx, uint64_t *keys, uint32_t *perm);
int cd_export_tree(cd_ctx *ctx, int32_t *parent, int32_t *left, int32_t *right, double *boxes,
                   uint32_t *bounded);

/* Tuning knobs; none of them changes results (pair sets, pairs_tested are identical for every setting). */
enum {
    CD_OPT_TRAVERSAL        = 0,   /* 0: lane-private FP64 descent, exact test inline (the reference's shape,        */
                                   /*    collision.cuh:19-71); 1: fp32 conservative descent of every query from the   */
                                   /*    root with a wavefront-shared LDS candidate queue + a second kernel for the   */
                                   /*    exact tests; 3 (default): half                                                */
                                   /*    traversal -- a query meets only the leaves to its right in Morton order and   */
                                   /*    every hit counts as the two ordered pairs the reference tests   */
    CD_OPT_SORT_FULL        = 2,   /* 0 (default): hybrid -- 2 global passes on 16 key bits (44..59, or 48..63 when a key reaches 2^60), the rest of the high half */
                                   /*    sorted inside LDS windows, stable fix-up of equal-high-half runs; falls back to 2, then  */
                                   /*    to 1, by itself when a run is too long.  2: half-key -- 4 global passes + the fix-up.   */
                                   /*    1: all 8 digit passes.  All give the identical stable order by the full 64-bit key.     */
    CD_OPT_KERNEL_STAMPS    = 4,   /* with CD_OPT_STAGE_TIMING 0: which time stamps a fused call still takes, a bit mask -- 1: the block-build   */
                                   /*    kernel (cd_stats.ms_build_block), 2: the descent kernel (ms_descend), 4: the exact kernel (with 2:      */
                                   /*    ms_exact, ms_traverse), 8: pipeline start (with 2 and 4: ms_pipeline).  Default 15.  A stamp rides on   */
                                   /*    its kernel's dispatch packet and still costs ~5 us of idle GPU; a field whose stamps are off reads 0    */
    CD_OPT_STAGE_TIMING     = 3,   /* 1 (default): HIP events around every stage (cd_stats.ms_morton ... ms_refit); 0: only the  */
                                   /*    events of the pipeline as a whole and of the descent kernel (ms_pipeline, ms_traverse,   */
                                   /*    ms_descend, ms_exact) -- each stage boundary costs a few idle microseconds                */
    CD_OPT_GRAPH            = 5,   /* 1: cd_self_collide replays its steady-state step -- the nine kernel launches -- as ONE hipGraph launch, captured on   */
                                   /*    the first eligible call (CD_OPT_STAGE_TIMING 0, CD_OPT_KERNEL_STAMPS 0, default sort / build / traversal, a previous */
                                   /*    step done); anything a replay cannot answer (a sort flag, an overflow, a deep pass) falls back to the stream path.  */
                                   /*    Default 0: measured equal to the stream path within noise -- the host is ahead of the GPU either way */
    CD_OPT_POLL             = 6,   /* 1 (default): with no time stamp pending (CD_OPT_STAGE_TIMING 0 and CD_OPT_KERNEL_STAMPS 0) a step's end is read off a   */
                                   /*    sequence word the report kernel stores last into pinned host memory (the host spins on its own memory; after 20 ms,   */
                                   /*    and every 64th step anyway, it synchronises the stream; so do the first two steps into a report area or a pinned    */
                                   /*    pair buffer the device has not written before); 0: always hipStreamSynchronize                                      */
    CD_OPT_CELL_TABLE       = 7,   /* 1 (default): when the vertices are uploaded (cd_create, cd_update_vertices) and some coordinate is not an fp32 value, a table of the   */
                                   /*    vertices' fp32 cells is built (which cells hold two distinct doubles), so that the fp32 boxes of the traversal keep equal bounds  */
                                   /*    equal: touching boxes do not overlap after rounding (a full-double mesh then steps as fast as its float-rounded copy; the table  */
                                   /*    costs the upload ~0.4 ms per million vertices).  0: no table, every coordinate that is not an fp32 value is rounded outward    */
                                   /*    (a full-double structured mesh: ~2 x the step time, nothing added to the upload).  Vertices that are all fp32 values -- what   */
                                   /*    the reference's loader produces, load_obj.h:38 -- never have a table                                                            */
    CD_OPT_ORDER_HINT       = 8,   /* 1 (default): the half traversal (CD_OPT_TRAVERSAL 3) of a fused call takes its groups of 64 leaves longest-first (per XCD; the kernel ends with its    */
                                   /*    unluckiest wave slot), by how long the PREVIOUS traversal's waves took -- remembered per triangle, so that the hint survives a mesh that moves and  */
                                   /*    sorts differently: 1 M cloth at rest -7 us per step, sheets moving a quad per frame -4 us.  Scheduling only: every group is traversed in every      */
                                   /*    step, results do not depend on it.  Trees of more than 2048 blocks (1 M triangles) and the stage-wise API run in the plain order.  0: always the   */
                                   /*    plain order.  2: the hint for larger trees too (an XCD's list sorted chunk by chunk of 2048 groups; measured slower there: 8 M 454 -> 471 us)    */
    CD_OPT_QUERIES_PER_WAVE = 1    /* variant 1: queries one wave works through with dynamic lane refill (x64)         */
};
int cd_set_option(cd_ctx *ctx, int key, int64_t value);

/* Measurement hooks of tools/ and switches the tests use to ask for one code path or another -- NOT part of the interface that mirrors the
 * reference, free to change, and in a key space of their own so that none of them can be mistaken for an option above.  Setters take
 * `value` and return CD_OK; getters (CD_DBG_GET_*) write *out.  None of the setters changes results. */
enum {
    CD_DBG_LDS_PAD            = 0,   /* extra dynamic LDS bytes per traversal workgroup (occupancy experiments)                              */
    CD_DBG_EXACT_BLOCKS       = 1,   /* workgroups of the exact kernel (default 1024)                                                         */
    CD_DBG_NO_SHARED_PATH     = 2,   /* variant 1 without the shared root path                                                                */
    CD_DBG_DIAG               = 3,   /* run the descent kernel's DIAG instance: per-phase step counts and s_memtime ticks -> cd_debug_counters */
    CD_DBG_STAGEWISE_BUILD    = 4,   /* fused entry points build the tree stage by stage (k_hierarchy + refit) instead of in one pass         */
    CD_DBG_SPLIT_CROSS        = 5,   /* the fused build's cross nodes by k_cross_meta + k_cross_records instead of k_cross_fused               */
    CD_DBG_REPORT_COPIES      = 6,   /* 1: the report kernel copies the first pairs into the host buffer (32 workgroups), as before round 4, instead of the exact kernel posting them (A/B) */
    CD_DBG_SORT_WINDOWS       = 7,   /* the in-LDS window sort behind the two global passes: 0 (default) windows of 2048 keys (512 threads, two workgroups a CU) until a run is too long   */
                                     /* for them, then 4096; 1 always windows of 4096 keys (one 1024-thread workgroup a CU); 2 always 2048.  Same keys and permutation either way (A/B, tests) */
    CD_DBG_GET_SORT_FORM      = 8,   /* the form the next sort takes: 0 two global passes on key bits 44..59 + window sorts (default), 1 the same on bits 48..63 (a key beyond 2^60: a  */
                                     /* centroid outside the Morton frame; retried as 0 every 64 sorts), 2 four passes + fix-up, 3 all eight passes (runs too long for the forms before)   */
    CD_DBG_STORE_QBOX         = 9,   /* 1: the fused build always stores the per-leaf query boxes (qbox[]); default 0: only when a reader is known -- the half traversal and the cross    */
                                     /* nodes take leaf boxes out of the records (round 6: 32 of the block build's 106 bytes a leaf).  Same records, same results (A/B, tests)              */
    CD_DBG_POLL_SCAN          = 10,  /* polled completion: poison the pair area before a step, scan it the moment the sequence word is seen   */
    CD_DBG_GET_POLL_STALE     = 11,  /* ... steps in which the scan found a pair missing (must stay 0)                                        */
    CD_DBG_GET_POLL_FALLBACKS = 12,  /* ... polled waits that ran into the 20 ms budget and ended in a stream synchronise                     */
    CD_DBG_GET_POLLED_STEPS   = 13,  /* ... reports whose end was read off the sequence word                                                  */
    CD_DBG_GET_POLL_FB_WHY    = 16,  /* ... why those waits ran out: bits 0..15 the stream was still busy at the time-out (the step was late), 16..31 the stream had drained and    */
                                     /* the word came with the synchronise (late in flight), 32..47 the word was not there after the drain (LOST: must stay 0)                          */
    CD_DBG_GET_POLL_MAX_WAIT_US = 17, /* ... the longest polled wait that ended in the word (sampled every 16th step), microseconds                                                   */
    CD_DBG_BIG_OFFSETS        = 18,  /* 1: the half traversal runs the instance that forms 64-bit record addresses (what trees of more than 2^27 leaves get) on a tree of any size (tests)  */
    CD_DBG_GET_TREE_WAS_FUSED = 14,  /* 1: the tree that is there was made by the one-pass build                                              */
    CD_DBG_GET_ORDER_STATE    = 15   /* the order hint (CD_OPT_ORDER_HINT) as it stands: 0 none built yet; 1 a permutation of the groups of 64 leaves that differs from the plain order;    */
                                     /* 2 the plain order itself; -1 not a permutation (must never be)                                                                                        */
};
int cd_debug_option(cd_ctx *ctx, int key, int64_t value, int64_t *out);

int cd_get_stats(cd_ctx *ctx, cd_stats *out);
/* Diagnostics of the last traversal, filled only after cd_debug_option(ctx, CD_DBG_DIAG, 1, NULL): sums over the descent's waves of
 * {chain steps, chain hops inside / outside the query's 256-leaf block, descent visits inside / outside it, longest
 * chain of each wave, 6 spare}.  Not part of any result. */
int cd_debug_counters(cd_ctx *ctx, unsigned long long out[12]);
/* The order hint's arrays (CD_OPT_ORDER_HINT), any may be NULL: per group of 64 leaves (ceil(nt / 64) words each) the score the last fused build gave it and the order
 * made from the scores (CD_ERR_ORDER while the tree that is there has none); per triangle, by its index in the face list (nt bytes), the time class (1.28 us each) its
 * wave left in the last half traversal.  Not part of any result. */
int cd_debug_hint(cd_ctx *ctx, uint32_t *cost, uint32_t *order, uint8_t *tri);
/* ... and the other way: install `order` (a permutation of the groups, checked: CD_ERR_ARG otherwise) as the hint the next half traversal of the tree that is there takes
 * (experiments with predictors: build the tree with cd_build_tree, install, cd_find_collisions). */
int cd_debug_hint_set(cd_ctx *ctx, const uint32_t *order);
/* Diagnostics: the fp32 traversal records of the current tree as the descent reads them -- recs: n x 64 bytes (n x 32 bytes
 * of right halves {lo[3], hi[3], link, last | flags}, then n x 32 bytes of left halves {lo[3], hi[3], link, first}, both
 * indexed by split), qboxes: n x 32 bytes {lo[3], hi[3], flags, 0}, root: the root record's split.  Either may be NULL. */
int cd_debug_records(cd_ctx *ctx, void *recs, void *qboxes, int32_t *root);
int cd_num_triangles(cd_ctx *ctx, uint32_t *nt);

/* ---- multi-GPU cross-rank pass (new work defined by the north star; no reference call site) ----
 * Query record, 88 bytes, device resident: the three vertices, the triangle ID and its three GLOBAL
 * vertex indices (neighborCount, triangle.cuh:18-30, compares indices). */
typedef struct cd_query {
    double   v[9];
    uint32_t id;
    uint32_t vidx[3];
} cd_query;

/* Global id of local vertex 0 (default 0).  Local triangles index the context's own vertex array;
 * across ranks neighborCount must compare GLOBAL vertex ids, so cd_pack_queries adds this base to the
 * indices it emits and cd_find_collisions_queries adds it to the local leaves' indices before comparing. */
int cd_set_vertex_id_base(cd_ctx *ctx, uint32_t base);
/* AABB of the whole local tree (box of internal node 0). */
int cd_root_box(cd_ctx *ctx, double box[6]);
/* Compact the local leaves whose AABB strictly overlaps `box` (box.cuh:40-43) into d_out, a DEVICE
 * buffer of cap records owned by the caller (e.g. a torch tensor handed to RCCL). *n = number found
 * (may exceed cap -> CD_OVERFLOW, nothing beyond cap is written). */
int cd_pack_queries(cd_ctx *ctx, const double box[6], void *d_out, uint64_t cap, uint64_t *n);
/* Traverse nq external queries (DEVICE buffer of cd_query) against the local tree; a pair
 * (q.id, leaf.id) is reported when q.id < leaf.id (tri_contact.cuh:81), no shared vertex index and
 * the exact test passes.  Output as cd_find_collisions. */
int cd_find_collisions_queries(cd_ctx *ctx, const void *d_queries, uint64_t nq,
                               uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs);

/* ---- the multi-GPU step in C/C++ behind the ABI (the harness it slots into is main.cu:47-174) ----
 * One process per GPU, one cd_ctx per process holding that rank's object(s) with GLOBAL triangle IDs and a vertex-id
 * base (cd_set_vertex_id_base).  A step = box of the rank's triangles -> ncclAllGather of the boxes -> one pack launch for
 * all overlapping peers (straight from the triangles) -> ncclAllGather of the per-peer counts -> grouped ncclSend / ncclRecv
 * of the records, WHILE the rank sorts, builds and traverses its own tree -> the received queries against that tree.
 * The one decision that makes the ranks repeat part of the step together (slab capacity) is taken from data all ranks
 * hold, so no rank is left waiting in a collective.  RCCL is loaded at run time (dlopen "librccl.so"; the environment
 * variable MI355CD_RCCL_LIBRARY names another library that provides the same ten calls). */
typedef struct cd_multi cd_multi;
enum {
    CD_MULTI_SELF_PEER = 1,    /* test mode: a rank also exchanges with ITSELF (ncclSend / ncclRecv to its own rank), so a    */
                               /* 1-rank communicator exercises every phase; the cross pass then reports the local pairs again */
    CD_MULTI_TIMING    = 2,    /* record HIP events at the phase boundaries (cd_multi_info.ms_*); costs a few idle us each     */
    CD_MULTI_CROSS_SERIAL = 8, /* A/B switch: the pass over the received queries runs behind the local traversal on its stream  */
                               /* instead of beside it on the second stream                                                     */
    CD_MULTI_SELF_SLICE = 4,   /* with CD_MULTI_SELF_PEER: the rank exchanges only the tenth of its triangles at its upper x end */
                               /* with itself -- a one-GPU rehearsal at the scale of a 10 % neighbour overlap                  */
    CD_MULTI_PRIORITY_STREAM = 32, /* at creation only: the second stream (all-gathers, pack, send / receive, the pass over the received queries) gets the */
                               /* device's highest stream priority.  Measured on one GPU (self-peer rehearsal, 1 M triangles): the two streams' kernels */
                               /* then slow each other down -- 0.65 against 0.32 ms per step -- so it is off by default                                     */
    CD_MULTI_INJECT_FAILURE = 16, /* test hook: this rank's NEXT step fails locally (CD_ERR_INJECTED) before its pipeline starts; the flag */
                               /* clears itself.  Every other rank must return CD_ERR_PEER from the same step, none may block      */
    CD_MULTI_INJECT_ALLOC_FAILURE = 64 /* test hook: this rank's NEXT allocation of its send / receive slabs fails (as out of memory); the flag   */
                               /* clears itself.  The step in which that happens fails on every rank; the next one allocates again   */
};
typedef struct cd_multi_info {
    uint32_t world, rank;          /* as the communicator reports them                                        */
    uint32_t n_peers;              /* ranks this rank sent to or received from                                */
    uint32_t host_syncs;           /* host synchronisations of the step (2 unless a pass had to be redone)    */
    uint32_t attempts;             /* 1 + collective repeats (slabs grown)                                    */
    uint32_t failed_rank_plus1;    /* a step that returned CD_ERR_PEER / a local error: 1 + the lowest rank that published a failure, else 0 */
    uint64_t sent_queries, recv_queries, local_pairs, cross_pairs, pairs_tested, query_cap;
    float ms_tree, ms_allgather, ms_pack, ms_counts, ms_exchange, ms_local, ms_cross;   /* CD_MULTI_TIMING; -1 = not measured.  NOT additive: */
                                   /* first stream: tree (Morton keys .. fused build), local (own traversal); second stream, beside them: allgather  */
                                   /* (from the step's start: box of the triangles + its all-gather), pack, counts (all-gather + copy to the host),  */
                                   /* exchange (send / receive), cross (end of the exchange -> end of the pass over the received queries)            */
    float pad1;
} cd_multi_info;
/* ncclGetUniqueId: 128 bytes, produced on one rank and handed to all (by whatever the launcher has: MPI, a file, ...). */
int cd_multi_unique_id(void *id128);
/* ncclCommInitRank(world, id, rank) on the current HIP device + the step's buffers.  COLLECTIVE: every rank of the communicator
 * calls it.  query_cap_per_peer: records per peer slab (0 = nt / 8 + 1024); the ranks agree on the LARGEST request here (one
 * 8-byte all-gather), so shards of unequal size start from a common capacity; it grows collectively when a step needs more.
 * One cd_multi per context (CD_ERR_ORDER otherwise).  Lifetime: cd_multi_destroy before cd_destroy; a context destroyed first
 * detaches the cd_multi, whose further steps return CD_ERR_ORDER.
 * Failure semantics of cd_multi_step: a rank whose own work fails still joins the step's collectives and publishes its error in
 * the count matrix; every rank then returns from the SAME step -- the failing rank its error, the others CD_ERR_PEER -- before
 * any send / receive is posted.  An error met after that point is returned to its caller and published by that rank's next step.
 * NOT covered by "every rank returns from the same step": (i) a rank whose RCCL all-gather cannot even be ENQUEUED (ncclAllGather
 * returns an error on that rank alone) -- its peers are inside the collective, the communicator is dead, and only destroying it
 * ends their wait; (ii) CD_ERR_ORDER / CD_ERR_RCCL / CD_ERR_ARG returned at the entry of cd_multi_step (context destroyed, no RCCL
 * library, null handle) and a failure of cd_multi_create before its small agreement buffers exist: those ranks never reach a collective,
 * so their peers must not call into one either -- these are programming or installation errors every rank of a job shares.
 * A failed allocation of the send / receive slabs (creation or growth) IS covered: the rank keeps its old buffers, publishes the
 * error, and allocates again at the start of its next step. */
int cd_multi_create(cd_multi **out, cd_ctx *ctx, const void *id128, int rank, int world, uint64_t query_cap_per_peer, int flags);
/* The same over a communicator the caller owns (an opaque ncclComm_t); it is not destroyed by cd_multi_destroy. */
int cd_multi_create_from_comm(cd_multi **out, cd_ctx *ctx, void *nccl_comm, uint64_t query_cap_per_peer, int flags);
void cd_multi_destroy(cd_multi *m);
/* Change CD_MULTI_* flags between steps (e.g. untimed steps first, then a few with CD_MULTI_TIMING). */
int cd_multi_set_flags(cd_multi *m, int flags);
/* One step.  pairs: local pairs first, then the cross pairs this rank owns (output format of cd_find_collisions);
 * returns CD_OVERFLOW when they do not fit cap_pairs (*n_pairs holds the true count).  info may be NULL. */
int cd_multi_step(cd_multi *m, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs, cd_multi_info *info);

/* Library / build identification: "mi355cd <version> gfx950". */
const char *cd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355CD_H */
