// cd_multi.h -- the multi-GPU step behind the C ABI (include/mi355cd.h, cd_multi_*): one process per GPU, triangles
// sharded by object, RCCL over xGMI for the one exchange the path has.  Included at the end of mi355cd.hip (it uses
// the stage enqueuers and the traversal passes of that file).
//
// The reference is single-GPU (main.cu:47-174 is the harness this slots into); SURVEY.md 8(e) defines what a step has to
// deliver (local pairs + each cross pair {a, b}, a.ID < b.ID, exactly once, from the owner of b: tri_contact.cuh:81
// applied to external queries too).  The order of work is chosen so that everything the ranks exchange is on its way
// before a rank starts on its own tree:
//   1. the box of all the rank's triangles, a reduction over their vertices (= the value node 0 of its tree will hold);
//      from here on two streams:
//   B. all-gather of the boxes (ncclAllGather, 48 B per rank); ONE launch compacts, for every peer whose box strictly
//      overlaps this rank's (box.cuh:40-43), the triangles that overlap that peer's box into cd_query records --
//      straight from the triangles in their original order: a query is a triangle, whatever tree it will sit in; the
//      per-peer counts are all-gathered as a world x world matrix, so every rank knows what it sends, what it receives
//      and whether ANY rank ran out of room -- the one decision taken from that matrix (grow the slabs) is the same on
//      every rank: no rank leaves a collective the others are still in
//   A. beside it, without waiting for anything: the rank's OWN pipeline (Morton keys, sort, fused build, half traversal
//      of its own tree, report)
//   B. once the host has read the matrix: the records travel (grouped ncclSend / ncclRecv, each peer's slice on its own
//      xGMI link, no ring), then -- as soon as the tree exists -- the received queries are traversed against it, beside
//      the local traversal; both reports are read with one wait.
// Host synchronisations per step: two (the count matrix -- with the GPU busy on A; both traversal passes).  A sort that
// must be redone in another form is local: the rank repeats its pipeline and the cross pass alone, after the collectives.
//
// Failing TOGETHER.  A rank that fails locally must not leave its peers waiting in a collective it never joins.  Every row of the
// count matrix therefore carries a STATUS word (slot W; 0 = fine): a rank whose own work has failed up to the point where the
// rows are all-gathered -- allocation, launch, its own pipeline's enqueue, an error kept from the previous step -- still joins
// both all-gathers, packs nothing and publishes its error; after the one read of the matrix EVERY rank sees it, nobody posts a
// send or a receive, and all return in the same step: the failing rank its own error, the others CD_ERR_PEER.  Nothing is
// allocated between the matrix and the exchange (the receive buffer is sized with the send slabs, the per-peer capacity is
// COMMON to all ranks from creation on and grows by a rule of the whole matrix), so the decision taken from the matrix is the
// last one that can differ; an error after it (an RCCL call, the final wait) closes the group it has opened, is returned, and is
// KEPT: the rank's next cd_multi_step publishes it, so the peers of a caller that goes on learn of it one step later.
//
// RCCL is loaded at run time (dlopen) by the first cd_multi_* call, so single-GPU users of the library do not pay for
// it; there is no fallback transport: without librccl the calls return CD_ERR_RCCL.
#pragma once
#include <dlfcn.h>
#include <cstdlib>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

RcclApi load_rccl()
{
    RcclApi api;
    // MI355CD_RCCL_LIBRARY: load THAT library and nothing else (a site's own RCCL build; the tests' in-process
    // loopback, tests/loopback_rccl).  Otherwise the process' RCCL: the copy already loaded (PyTorch's) wins by soname.
    const char *over = std::getenv("MI355CD_RCCL_LIBRARY");
    void *h = nullptr;
    if (over && *over) h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    else {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
    if (h) {
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
        api.CommCount = (decltype(api.CommCount))dlsym(h, "ncclCommCount");
        api.CommUserRank = (decltype(api.CommUserRank))dlsym(h, "ncclCommUserRank");
        api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
        api.Send = (decltype(api.Send))dlsym(h, "ncclSend");
        api.Recv = (decltype(api.Recv))dlsym(h, "ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
        if (api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommCount && api.CommUserRank && api.AllGather && api.Send && api.Recv &&
            api.GroupStart && api.GroupEnd)
            api.handle = h;
    }
    return api;
}

RcclApi *rccl()
{
    static RcclApi api = load_rccl();                   // (initialised once, thread-safe: ranks may be threads of one process)
    return api.handle ? &api : nullptr;
}

#define NCCLCHK(expr)                                                  \
    do {                                                               \
        ncclResult_t r_ = (expr);                                      \
        if (r_ != ncclSuccess) return CD_ERR_RCCL;                     \
    } while (0)

// rehearsal only (CD_MULTI_SELF_SLICE): the box the rank sees of ITSELF as a peer keeps the upper tenth of its x extent --
// about the share of its triangles a config-4 neighbour's box covers
__global__ void k_slice_box(double *box) { box[0] = box[1] - 0.1 * (box[1] - box[0]); }
__global__ void k_set_word(unsigned long long *w, unsigned long long v) { *w = v; }

// The box of all the rank's triangles when every vertex belongs to one (checked at cd_create): a streaming min / max over the VERTICES -- 24 coalesced
// bytes each, no index gathers -- in ONE launch: a workgroup leaves its partial box and arrives on a counter, the workgroup that arrives last folds the
// partials (device-scope loads) and writes the box; the counter resets itself.  The kernel also zeroes what the second stream needs zeroed before its
// next jobs (the row of per-peer counts, the counters of the pass over the received queries): the stream's prologue was five launches -- triangle-wise
// bounds 23 us, a one-workgroup fold 9 us, two fills and the all-gather -- in front of the pack; this is one of ~5 us.
constexpr int VBOX_BLOCKS = 64, VBOX_THREADS = 1024;      // (at most 64 workgroups: the fold is one wave)
__global__ __launch_bounds__(VBOX_THREADS) void k_vertex_box(const double *__restrict__ verts, uint32_t nv, double *partial /* [6][VBOX_BLOCKS] */, uint32_t *arrive,
                                                             double *__restrict__ box /* x1 x2 y1 y2 z1 z2 */, uint32_t *__restrict__ zero0, uint32_t nzero0,
                                                             uint32_t *__restrict__ zero1, uint32_t nzero1)
{
    __shared__ double sm[VBOX_THREADS / 64][6];
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (blockIdx.x == 0) { for (uint32_t i = tid; i < nzero0; i += VBOX_THREADS) zero0[i] = 0u; }
    if (blockIdx.x == 1 || gridDim.x == 1) { for (uint32_t i = tid; i < nzero1; i += VBOX_THREADS) zero1[i] = 0u; }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t v = blockIdx.x * VBOX_THREADS + tid; v < nv; v += gridDim.x * VBOX_THREADS) {
        const d3 p = load_vertex(verts, v);
        lo[0] = p.x < lo[0] ? p.x : lo[0]; hi[0] = p.x > hi[0] ? p.x : hi[0];
        lo[1] = p.y < lo[1] ? p.y : lo[1]; hi[1] = p.y > hi[1] ? p.y : hi[1];
        lo[2] = p.z < lo[2] ? p.z : lo[2]; hi[2] = p.z > hi[2] ? p.z : hi[2];
    }
    for (int a = 0; a < 3; ++a) { const double l = wave_min(lo[a]), h = wave_max(hi[a]); if (lane == 0) { sm[w][2 * a] = l; sm[w][2 * a + 1] = h; } }
    __syncthreads();
    if (tid < 6) {
        const bool is_lo = (tid & 1u) == 0u;
        double v = sm[0][tid];
        for (int ww = 1; ww < VBOX_THREADS / 64; ++ww) { const double t = sm[ww][tid]; v = is_lo ? (t < v ? t : v) : (t > v ? t : v); }
        __hip_atomic_store(&partial[tid * gridDim.x + blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);                                         // the six stores above came from lanes of THIS wave: complete before the arrival
        s_last = (atomicAdd(arrive, 1u) + 1u == gridDim.x) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && tid < 64) {                                                  // (one wave: lane b takes workgroup b's partials, all six loads in flight together)
        double v[6];
        for (int a = 0; a < 6; ++a) v[a] = tid < gridDim.x ? __hip_atomic_load(&partial[a * gridDim.x + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((a & 1) ? -1e300 : 1e300);
        for (int a = 0; a < 6; ++a) { const double r = (a & 1) ? wave_max(v[a]) : wave_min(v[a]); if (tid == 0) box[a] = r; }
        if (tid == 0) *arrive = 0u;                                            // (for the next step)
    }
}

enum MEv { ME_START, ME_LOC0, ME_TREE, ME_GATHER, ME_PACK, ME_COUNTS, ME_XCH0, ME_XCH1, ME_LOCAL, ME_CROSS, ME_COUNT };

}  // namespace

struct cd_multi {
    cd_ctx *c = nullptr;
    ncclComm_t comm = nullptr; bool own_comm = false;
    int rank = 0, world = 1, flags = 0;
    uint64_t qcap = 0;                           // records per peer slab -- the SAME on every rank (max of the ranks' requests at creation, grown from the shared count matrix)
    int sticky_err = 0;                          // an error met AFTER the last collective decision of a step: published by the next step's status word
    hipStream_t xstream = nullptr;               // payload exchange, beside the context's stream.  Default priority: with the device's HIGHEST priority
                                                 // (CD_MULTI_PRIORITY_STREAM) the two streams' kernels slow each other down -- 0.65 against 0.32 ms per step in
                                                 // the one-GPU rehearsal at config 4's scale (profiles/r03_experiments/multi_priority_stream.log)
    hipEvent_t ev_payload = nullptr, ev_counts = nullptr, ev_tree = nullptr, ev_cross = nullptr, ev_box = nullptr, ev[ME_COUNT] = {};
    double *d_myroot = nullptr;                  // 6: the box of all this rank's triangles, from their vertices
    double *d_partial = nullptr;                 // per-block bounds of that reduction (the context's own are the first stream's, for its Morton frame)
    double *d_roots = nullptr;                   // world x 6
    uint32_t *d_arrive = nullptr;                // k_vertex_box's arrival counter (self-resetting)
    unsigned long long *d_row = nullptr;         // world + 1: records packed for each peer | this rank's status word
    unsigned long long *d_matrix = nullptr;      // world x (world + 1), all-gathered rows
    unsigned long long *h_matrix = nullptr;      // pinned copy
    double *h_roots = nullptr;                   // pinned, world x 6
    uint64_t slab_cap = 0;                       // records per peer slab the two buffers below were ALLOCATED with (< qcap only after a growth whose allocation failed)
    ExtQuery *d_send = nullptr;                  // world slabs of slab_cap records
    ExtQuery *d_recv = nullptr;                  // world x qcap as well: whatever the peers send fits (allocated WITH the slabs, never between matrix and exchange)
    std::vector<uint32_t> scratch_pairs;
};

namespace {

void multi_free(cd_multi *m)
{
    if (!m) return;
    if (m->c) { hipStreamSynchronize(m->c->stream); if (m->c->attached_multi == m) m->c->attached_multi = nullptr; }
    if (m->xstream) { hipStreamSynchronize(m->xstream); hipStreamDestroy(m->xstream); }
    if (m->ev_payload) hipEventDestroy(m->ev_payload);
    if (m->ev_counts) hipEventDestroy(m->ev_counts);
    if (m->ev_tree) hipEventDestroy(m->ev_tree);
    if (m->ev_box) hipEventDestroy(m->ev_box);
    if (m->ev_cross) hipEventDestroy(m->ev_cross);
    for (int i = 0; i < ME_COUNT; ++i) if (m->ev[i]) hipEventDestroy(m->ev[i]);
    hipFree(m->d_arrive); hipFree(m->d_myroot); hipFree(m->d_partial); hipFree(m->d_roots); hipFree(m->d_row); hipFree(m->d_matrix); hipFree(m->d_send); hipFree(m->d_recv);
    if (m->h_matrix) hipHostFree(m->h_matrix);
    if (m->h_roots) hipHostFree(m->h_roots);
    if (m->own_comm && m->comm && rccl()) rccl()->CommDestroy(m->comm);
    delete m;
}

// What a rank needs to AGREE with the others (creation's all-gather of the capacities, a step's status word): the second stream, a row, the matrix and its pinned
// copy -- a few hundred bytes, allocated FIRST, so that a rank whose larger allocations fail afterwards can still tell the others (round-3 advisor: the agreement used
// to be skipped when any of the first allocations failed, and the peers waited in ncclAllGather).  A rank that cannot even have these cannot tell anybody anything:
// that is a rank that died, and the job's launcher deals with it.
int multi_alloc_agreement(cd_multi *m)
{
    const size_t W = (size_t)m->world;
    if (m->flags & CD_MULTI_PRIORITY_STREAM) {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&m->xstream, hipStreamNonBlocking, greatest));
    } else HIPCHK(hipStreamCreateWithFlags(&m->xstream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&m->d_row, sizeof(unsigned long long) * (W + 1)));
    HIPCHK(hipMalloc(&m->d_matrix, sizeof(unsigned long long) * W * (W + 1)));
    HIPCHK(hipHostMalloc(&m->h_matrix, sizeof(unsigned long long) * W * (W + 1), hipHostMallocDefault));
    HIPCHK(hipMalloc(&m->d_myroot, sizeof(double) * 6));
    HIPCHK(hipMalloc(&m->d_roots, sizeof(double) * 6 * W));
    return CD_OK;
}
int multi_alloc(cd_multi *m)
{
    const size_t W = (size_t)m->world;
    if (m->flags & CD_MULTI_INJECT_ALLOC_FAILURE) { m->flags &= ~CD_MULTI_INJECT_ALLOC_FAILURE; return -(int)hipErrorOutOfMemory; }   // test hook (at creation: this rank's first allocation fails)
    HIPCHK(hipEventCreateWithFlags(&m->ev_payload, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_counts, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&m->ev_tree));                                                    // (a kernel's stop event: with time stamps)
    HIPCHK(hipEventCreateWithFlags(&m->ev_box, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_cross, hipEventDisableTiming));
    for (int i = 0; i < ME_COUNT; ++i) HIPCHK(hipEventCreate(&m->ev[i]));
    HIPCHK(hipMalloc(&m->d_arrive, sizeof(uint32_t) * 16));
    HIPCHK(hipMemset(m->d_arrive, 0, sizeof(uint32_t) * 16));
    HIPCHK(hipMalloc(&m->d_partial, sizeof(double) * BOUNDS_STRIDE * BOUNDS_BLOCKS));
    HIPCHK(hipHostMalloc(&m->h_roots, sizeof(double) * 6 * W, hipHostMallocDefault));
    // the external pass has its own counters, candidates, pairs and deferred list (TravBuf [1])
    TravBuf &tb = m->c->tb[1];
    if (!tb.d_state) { HIPCHK(hipMalloc(&tb.d_state, sizeof(TravState))); tb.state_owned = true; }
    if (!tb.d_cand) { tb.cand_cap = (uint64_t)(1u << 20); HIPCHK(hipMalloc(&tb.d_cand, sizeof(Candidates) * tb.cand_cap)); }
    if (!tb.d_defer) { tb.defer_cap = 1u << 14; HIPCHK(hipMalloc(&tb.d_defer, sizeof(uint2) * tb.defer_cap)); }
    return CD_OK;
}

// the send slabs and the receive buffer, both world x qcap records: (re)allocated together, only where a failure can still be
// published through the status word of the NEXT all-gather
// The new buffers are allocated BEFORE the old ones are released: when an allocation fails the rank keeps a consistent pair (of the
// old capacity), reports the error through its status word, and the next step -- which sees slab_cap < qcap, the capacity the other
// ranks have by now -- tries again before anything is packed or received; nothing is ever packed into or received by a null or
// undersized buffer.
int multi_slabs(cd_multi *m)
{
    if (m->d_send && m->d_recv && m->slab_cap >= m->qcap) return CD_OK;
    ExtQuery *ns = nullptr, *nr = nullptr;
    hipError_t e = (m->flags & CD_MULTI_INJECT_ALLOC_FAILURE) ? hipErrorOutOfMemory : hipSuccess;      // test hook: this rank's next slab allocation fails
    m->flags &= ~CD_MULTI_INJECT_ALLOC_FAILURE;
    if (e == hipSuccess) e = hipMalloc(&ns, sizeof(ExtQuery) * (size_t)m->world * m->qcap);
    if (e == hipSuccess) e = hipMalloc(&nr, sizeof(ExtQuery) * (size_t)m->world * m->qcap);
    if (e != hipSuccess) { hipFree(ns); hipFree(nr); (void)hipGetLastError(); return -(int)e; }
    hipFree(m->d_send); hipFree(m->d_recv);
    m->d_send = ns; m->d_recv = nr; m->slab_cap = m->qcap;
    return CD_OK;
}

}  // namespace

// cd_destroy with a cd_multi still attached: the multi step's state is released first (its streams wait for the context's), the
// cd_multi object itself stays valid for cd_multi_destroy and refuses further steps (CD_ERR_ORDER)
void multi_detach_from(cd_ctx *c)
{
    cd_multi *m = c->attached_multi;
    if (!m) return;
    hipStreamSynchronize(c->stream);
    if (m->xstream) hipStreamSynchronize(m->xstream);
    m->c = nullptr;
    c->attached_multi = nullptr;
}

extern "C" {

int cd_multi_unique_id(void *id128)
{
    if (!id128) return CD_ERR_ARG;
    RcclApi *r = rccl();
    if (!r) return CD_ERR_RCCL;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    NCCLCHK(r->GetUniqueId(reinterpret_cast<ncclUniqueId *>(id128)));
    return CD_OK;
}

static int multi_create_common(cd_multi **out, cd_ctx *ctx, ncclComm_t comm, bool own, int rank, int world, uint64_t query_cap_per_peer, int flags)
{
    cd_multi *m = new (std::nothrow) cd_multi();
    if (!m) return CD_ERR_ARG;
    m->c = ctx; m->comm = comm; m->own_comm = own; m->rank = rank; m->world = world; m->flags = flags;
    m->qcap = query_cap_per_peer ? query_cap_per_peer : (uint64_t)(ctx->nt / 8 + 1024);
    RcclApi *r = rccl();
    int rc = r ? multi_alloc_agreement(m) : CD_ERR_RCCL;
    if (rc) { if (!own) m->comm = nullptr; multi_free(m); return rc; }     // (cannot agree on anything: see multi_alloc_agreement)
    rc = multi_alloc(m);
    // The per-peer capacity must be the SAME on every rank (the decision to grow it is taken by all from the shared matrix):
    // shards of unequal size -- or callers with different arguments -- would start with different values, so the ranks agree
    // on the largest request here, once (creation is a collective already: ncclCommInitRank).  A rank whose allocations failed
    // still joins (with 0), so nobody waits for it; it then returns its error.
    {
        const unsigned long long mine = rc ? 0ull : (unsigned long long)m->qcap;
        k_set_word<<<1, 1, 0, m->xstream>>>(m->d_row, mine);
        const ncclResult_t ar = r->AllGather(m->d_row, m->d_matrix, 1, ncclUint64, m->comm, m->xstream);
        hipError_t e = hipMemcpyAsync(m->h_matrix, m->d_matrix, sizeof(unsigned long long) * (size_t)world, hipMemcpyDeviceToHost, m->xstream);
        if (e == hipSuccess) e = hipStreamSynchronize(m->xstream);
        if (!rc && ar != ncclSuccess) rc = CD_ERR_RCCL;
        if (!rc && e != hipSuccess) rc = -(int)e;
        if (!rc) {
            bool peer_failed = false;
            for (int p = 0; p < world; ++p) { if (m->h_matrix[p] == 0) peer_failed = true; m->qcap = std::max<uint64_t>(m->qcap, m->h_matrix[p]); }
            if (peer_failed) rc = CD_ERR_PEER;
        }
    }
    if (!rc) rc = multi_slabs(m);
    if (rc) { if (!own) m->comm = nullptr; multi_free(m); return rc; }     // (an owned communicator is destroyed by multi_free, ONCE)
    ctx->attached_multi = m;
    *out = m;
    return CD_OK;
}

int cd_multi_create(cd_multi **out, cd_ctx *ctx, const void *id128, int rank, int world, uint64_t query_cap_per_peer, int flags)
{
    if (!out || !ctx || !id128 || world < 1 || rank < 0 || rank >= world) return CD_ERR_ARG;
    *out = nullptr;
    if (ctx->attached_multi) return CD_ERR_ORDER;                          // one cd_multi per context
    RcclApi *r = rccl();
    if (!r) return CD_ERR_RCCL;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    NCCLCHK(r->CommInitRank(&comm, world, id, rank));
    return multi_create_common(out, ctx, comm, true, rank, world, query_cap_per_peer, flags);     // (on failure multi_free has destroyed the communicator)
}

int cd_multi_create_from_comm(cd_multi **out, cd_ctx *ctx, void *nccl_comm, uint64_t query_cap_per_peer, int flags)
{
    if (!out || !ctx || !nccl_comm) return CD_ERR_ARG;
    *out = nullptr;
    if (ctx->attached_multi) return CD_ERR_ORDER;
    RcclApi *r = rccl();
    if (!r) return CD_ERR_RCCL;
    int world = 0, rank = 0;
    NCCLCHK(r->CommCount(static_cast<ncclComm_t>(nccl_comm), &world));
    NCCLCHK(r->CommUserRank(static_cast<ncclComm_t>(nccl_comm), &rank));
    return multi_create_common(out, ctx, static_cast<ncclComm_t>(nccl_comm), false, rank, world, query_cap_per_peer, flags);
}

void cd_multi_destroy(cd_multi *m) { multi_free(m); }

int cd_multi_set_flags(cd_multi *m, int flags) { if (!m) return CD_ERR_ARG; m->flags = flags; return CD_OK; }

int cd_multi_step(cd_multi *m, uint32_t *pairs, uint64_t cap_pairs, uint64_t *n_pairs, cd_multi_info *info)
{
    if (!m) return CD_ERR_ARG;
    RcclApi *r = rccl();
    if (!r) return CD_ERR_RCCL;                                               // (cannot happen: a cd_multi exists only where librccl was loaded)
    if (!m->c) {
        // The context was destroyed under this cd_multi (cd_destroy detaches it): this rank has nothing to step -- but its peers are in the step's collectives.  It
        // joins them with an empty box, no queries and CD_ERR_ORDER in its status word; every rank reads it from the matrix and leaves the step before anything
        // is sent (round-3 advisor: the bare return left the peers waiting in ncclAllGather).
        const size_t RWo = (size_t)m->world + 1;
        hipStream_t xs = m->xstream;
        (void)hipMemsetAsync(m->d_myroot, 0, sizeof(double) * 6, xs);
        (void)r->AllGather(m->d_myroot, m->d_roots, 6, ncclDouble, m->comm, xs);
        (void)hipMemsetAsync(m->d_row, 0, sizeof(unsigned long long) * RWo, xs);
        k_set_word<<<1, 1, 0, xs>>>(m->d_row + m->world, (unsigned long long)(uint32_t)(-CD_ERR_ORDER));
        (void)r->AllGather(m->d_row, m->d_matrix, RWo, ncclUint64, m->comm, xs);
        (void)hipStreamSynchronize(xs);
        (void)hipGetLastError();
        if (info) { std::memset(info, 0, sizeof *info); info->world = (uint32_t)m->world; info->failed_rank_plus1 = (uint32_t)m->rank + 1u; }
        return CD_ERR_ORDER;
    }
    cd_ctx *c = m->c;
    sort_retry_tick(c);
    hipStream_t s = c->stream;
    const int W = m->world, me = m->rank;
    const size_t RW = (size_t)W + 1;                                          // a row of the matrix: W counts | the rank's status word
    const bool self_peer = (m->flags & CD_MULTI_SELF_PEER) != 0, timing = (m->flags & CD_MULTI_TIMING) != 0;
    uint32_t syncs = 0, attempts = 0;
    auto mark = [&](int e, hipStream_t st) { if (timing) hipEventRecord(m->ev[e], st); };
    if (info) std::memset(info, 0, sizeof *info);

    // What has gone wrong on THIS rank so far.  From here to the read of the count matrix nothing returns: a rank with an error
    // still joins both all-gathers (it packs nothing and publishes the error in its row), so that every rank leaves the step
    // together -- see the header of this file.
    int local_err = m->sticky_err;                                            // kept from the previous step: an error met after its last collective decision
    m->sticky_err = 0;
    if (!local_err && cap_pairs && !pairs) local_err = CD_ERR_ARG;
    if (!local_err && (m->flags & CD_MULTI_INJECT_FAILURE)) local_err = CD_ERR_INJECTED;   // test hook: this rank's next step fails locally
    m->flags &= ~CD_MULTI_INJECT_FAILURE;
    if (!local_err && m->slab_cap < m->qcap) { const int rc = multi_slabs(m); if (rc) local_err = rc; }   // a growth of an earlier step whose allocation failed here: again, before anything is packed
    auto late = [&](int rc) { m->sticky_err = rc; c->scratch_clean = false; hipStreamSynchronize(s); hipStreamSynchronize(m->xstream); return rc; };   // an error AFTER the decision: returned, and published by the next step
#define LATE_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return late(-(int)e_); } while (0)
#define SOFT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess && !local_err) local_err = -(int)e_; } while (0)

    TravBuf &t0 = c->tb[0], &t1 = c->tb[1];
    const bool fast_path = c->trav_variant != 0;                              // (variant 0 has no candidate stage: take the general path)
    const uint64_t cap = cap_pairs;
    uint64_t spec0 = !pairs ? 0 : cap < SPEC_PAIRS ? cap : SPEC_PAIRS, spec1 = spec0;     // (enqueue_report clamps them the same way)
    if (fast_path && !local_err) {
        int rc = ensure_pairs(c, t0, cap > 0 ? cap : 1);
        if (!rc) rc = ensure_pairs(c, t1, cap > 0 ? cap : 1);
        if (rc) local_err = rc;
    }
    const bool se = c->stage_events;
    struct RestoreStageEvents { cd_ctx *c; bool v; uint32_t mask; ~RestoreStageEvents() { c->stage_events = v; c->stamp_mask = mask; c->prezeroed = false; } } restore{c, se, c->stamp_mask};
    c->stamp_mask = 0;                            // no per-kernel time stamps inside a multi step (~5 us of idle GPU each; nobody reads them here)

    // The rank's OWN pipeline -- Morton keys, sort, fused build, the half traversal of its own tree, the report -- needs
    // nothing from the other ranks: it runs on the first stream while the boxes, the counts and the records travel on the
    // second.  A sort that has to be redone in another form (cd_sort.h) is this rank's own business: it repeats this part
    // alone, after the step's collectives.
    // (in two parts: the sort goes to the device BEFORE the host spends its tens of microseconds on the second stream's RCCL
    //  calls, the rest after them)
    auto enqueue_sort = [&](bool again) -> int {                                           // (again: the bounds of an auto frame are still there)
        mark(ME_LOC0, s);
        c->prezeroed = true;                                                               // (one memset for every counter of the pipeline, as in cd_self_collide)
        const int rc = enqueue_morton_sort(c, !fused_build_next(c), /*frame_ready=*/again);
        if (rc) c->prezeroed = false;
        return rc;
    };
    auto enqueue_rest = [&]() -> int {
        c->tree_done_event = m->ev_tree;                                                   // (rides on the tree's last kernel when that launch can carry it)
        int rc = enqueue_tree(c);
        const bool carried = c->tree_done_event == nullptr;
        c->tree_done_event = nullptr;
        c->prezeroed = false;
        if (rc) return rc;
        // (the sort's flags come back with the local pass's report; a copy of their own into pageable memory stalls the stream for ~20 us)
        if (!fast_path) HIPCHK(hipMemcpyAsync(c->sort_flags, c->d_os_ticket + 8, sizeof c->sort_flags, hipMemcpyDeviceToHost, s));
        if (!carried) HIPCHK(hipEventRecord(m->ev_tree, s));                               // the tree exists: the cross pass may start (second stream)
        mark(ME_TREE, s);
        if (fast_path) {
            // (stage events off: the kernels' time stamps ride on their dispatch packets, no barrier packets between the passes)
            c->stage_events = false;
            QuerySrc src{c->d_leaf, c->d_boxes, c->d_qbox, c->d_root, nullptr, nullptr, c->d_os_ticket + 8};
            // (the fused build has just zeroed the traversal counters itself -- ZeroPlan, cd_build.h -- unless the sort took a form that
            //  goes without; a memset here sits between the tree and the traversal: 12 us of the step)
            if (!c->scratch_clean) HIPCHK(hipMemsetAsync(t0.d_state, 0, sizeof(TravState), s));
            launch_pass<false, false>(c, t0, src, c->nt, cap);
            rc = enqueue_report(c, t0, pairs != nullptr, spec0);
            if (rc) return rc;
        }
        mark(ME_LOCAL, s);
        return CD_OK;
    };

    // ---- 1: what the OTHER ranks need from this one comes first: the box of all its triangles (a reduction over their
    // vertices: the value node 0 of the tree will hold, known before there is a tree).  Then the step forks.  SECOND stream:
    // all-gather of the boxes, pack of the triangles that overlap each peer's box (from the triangles in their ORIGINAL
    // order), all-gather of the count matrix.  Order of issue on the host: the sort's launches (first stream: ~85 us of
    // work to chew on), then the second stream's box / all-gather / pack (RCCL's calls cost the host tens of microseconds),
    // then tree and traversal, then -- the rank's status being final only now -- the all-gather of the rows.
    // The pack streams the vertices while the sort's latency-bound passes leave the memory system idle.
    mark(ME_START, s);
    hipStream_t xs = m->xstream;
    // The box of the triangles is the SECOND stream's first job (its own pass over the vertices, its own partials): the first
    // stream starts the rank's pipeline at once and no event ties the two together before the tree is there (an event record
    // between two kernels of a stream is a barrier packet: ~6 us of idle GPU).
    if (!local_err) { const int rc = enqueue_sort(false); if (rc) local_err = rc; }
    // (the counters of the pass over the received queries are zeroed here too, where the second stream has time -- behind the exchange a
    //  memset would sit between the records' arrival and the pass, 8 us + a launch gap on the step's longest chain)
    const bool zero_cross = !(m->flags & CD_MULTI_CROSS_SERIAL);
    bool row_zeroed = false;
    if (c->all_verts_referenced) {
        k_vertex_box<<<VBOX_BLOCKS, VBOX_THREADS, 0, xs>>>(c->d_verts, c->nv, m->d_partial, m->d_arrive, m->d_myroot,
                                                           reinterpret_cast<uint32_t *>(m->d_row), (uint32_t)(2 * RW),
                                                           reinterpret_cast<uint32_t *>(t1.d_state), zero_cross ? (uint32_t)(sizeof(TravState) / sizeof(uint32_t)) : 0u);
        row_zeroed = true;
    } else {
        k_centroid_bounds<true><<<BOUNDS_BLOCKS, 256, 0, xs>>>(c->d_verts, c->d_vidx, c->nt, m->d_partial);
        k_frame_from_bounds<<<1, 256, 0, xs>>>(m->d_partial, BOUNDS_BLOCKS, nullptr, m->d_myroot);
        if (zero_cross) SOFT_HIP(hipMemsetAsync(t1.d_state, 0, sizeof(TravState), xs));
    }
    int failed_rank = -1;
    for (;; ++attempts) {
        if (r->AllGather(m->d_myroot, m->d_roots, 6, ncclDouble, m->comm, xs) != ncclSuccess && !local_err) local_err = CD_ERR_RCCL;
        if (self_peer && (m->flags & CD_MULTI_SELF_SLICE)) k_slice_box<<<1, 1, 0, xs>>>(m->d_roots + 6 * (size_t)me);   // rehearsal at config 4's scale
        mark(ME_GATHER, xs);
        if (!(row_zeroed && attempts == 0)) SOFT_HIP(hipMemsetAsync(m->d_row, 0, sizeof(unsigned long long) * RW, xs));   // (the first round's: by k_vertex_box)
        if (!local_err && m->d_send && m->slab_cap >= m->qcap)
            k_pack_triangles<<<cdiv(c->nt, PACK_THREADS), PACK_THREADS, 0, xs>>>(c->d_verts, c->d_vidx, c->d_ids, (int)c->nt, m->d_roots, W,
                                                                                  self_peer ? -1 : me, m->d_roots + 6 * (size_t)me,
                                                                                  m->d_send, (unsigned long long)m->qcap, m->d_row, c->vbase);
        mark(ME_PACK, xs);
        // the rank's own tree and traversal go to the first stream BEFORE its row is published: whatever their enqueue reports is in the status word
        if (attempts == 0 && !local_err) { const int rc = enqueue_rest(); if (rc) local_err = rc; }   // (a repeat only grows the slabs and packs again: the pipeline in flight stays valid)
        SOFT_HIP(hipGetLastError());
        if (local_err) k_set_word<<<1, 1, 0, xs>>>(m->d_row + W, (unsigned long long)(uint32_t)(-local_err));
        if (r->AllGather(m->d_row, m->d_matrix, RW, ncclUint64, m->comm, xs) != ncclSuccess && !local_err) local_err = CD_ERR_RCCL;
        SOFT_HIP(hipMemcpyAsync(m->h_matrix, m->d_matrix, sizeof(unsigned long long) * W * RW, hipMemcpyDeviceToHost, xs));
        SOFT_HIP(hipMemcpyAsync(m->h_roots, m->d_roots, sizeof(double) * 6 * W, hipMemcpyDeviceToHost, xs));
        mark(ME_COUNTS, xs);
        SOFT_HIP(hipEventRecord(m->ev_counts, xs));
        SOFT_HIP(hipEventSynchronize(m->ev_counts)); ++syncs;                              // host synchronisation 1 of 2: the counts (the first stream keeps working)
        // every decision taken from the matrix is a function of the WHOLE matrix: identical on every rank
        for (int a = 0; a < W && failed_rank < 0; ++a) if (m->h_matrix[(size_t)a * RW + W]) failed_rank = a;
        if (failed_rank >= 0 || local_err) break;                                          // (local_err without a published status: the all-gather itself failed here)
        unsigned long long mx = 0;
        for (int a = 0; a < W; ++a) for (int p = 0; p < W; ++p) mx = std::max(mx, m->h_matrix[(size_t)a * RW + p]);
        if (mx <= m->qcap) break;
        if (attempts >= 5) { failed_rank = W; break; }                                     // (the same count on every rank: they all give up here)
        m->qcap = mx + mx / 4 + 1024;                                                      // same rule, same input -> same capacity on every rank
        { const int rc = multi_slabs(m); if (rc) local_err = rc; }                         // (a failure here is published by the next round's status word)
    }
    if (failed_rank >= 0 || local_err) {
        // Some rank has failed (perhaps this one): NOBODY posts a send or a receive.  Whatever this rank has in flight drains; the next
        // step starts from scratch.
        hipStreamSynchronize(s); hipStreamSynchronize(xs);
        (void)hipGetLastError();
        c->scratch_clean = false; c->prezeroed = false;
        if (info) { info->world = (uint32_t)W; info->rank = (uint32_t)me; info->host_syncs = syncs; info->attempts = attempts + 1; info->query_cap = m->qcap;
                    info->failed_rank_plus1 = failed_rank >= 0 && failed_rank < W ? (uint32_t)failed_rank + 1u : 0u; }
        if (n_pairs) *n_pairs = 0;
        if (local_err) return local_err;
        return failed_rank == W ? CD_ERR_ARG : CD_ERR_PEER;
    }

    // ---- 2: payload exchange on the second stream (everything it reads was complete at the synchronisation above)
    uint64_t sent = 0, recvd = 0; uint32_t n_peers = 0;
    std::vector<uint64_t> roff((size_t)W + 1, 0);
    for (int p = 0; p < W; ++p) {
        const uint64_t from_p = m->h_matrix[(size_t)p * RW + me];
        roff[p + 1] = roff[p] + from_p;
        const uint64_t to_p = m->h_matrix[(size_t)me * RW + p];
        sent += to_p; recvd += from_p;
        if (to_p || from_p) ++n_peers;
    }
    // (recvd <= W x qcap: every count passed the capacity test above, and the receive buffer was allocated with the slabs)
    mark(ME_XCH0, m->xstream);
    if (sent || recvd) {
        bool bad = r->GroupStart() != ncclSuccess;
        for (int p = 0; p < W; ++p) {
            const uint64_t to_p = m->h_matrix[(size_t)me * RW + p], from_p = m->h_matrix[(size_t)p * RW + me];
            if (to_p) bad |= r->Send(m->d_send + (size_t)p * m->qcap, to_p * sizeof(ExtQuery), ncclChar, p, m->comm, m->xstream) != ncclSuccess;
            if (from_p) bad |= r->Recv(m->d_recv + roff[p], from_p * sizeof(ExtQuery), ncclChar, p, m->comm, m->xstream) != ncclSuccess;
        }
        bad |= r->GroupEnd() != ncclSuccess;                                               // (a group that was opened is always closed)
        if (bad) return late(CD_ERR_RCCL);
    }
    mark(ME_XCH1, m->xstream);
    LATE_HIP(hipEventRecord(m->ev_payload, m->xstream));

    // ---- 3: the pass over the received queries, on the SECOND stream behind the exchange: it needs the records and the
    // tree, nothing of the local traversal -- the two descents share the chip and the latency-bound ends of the two passes
    // (exact tests, reports) overlap.  ONE wait for both.
    uint64_t n_local = 0, n_cross = 0, tested = 0;
    int rc_l = CD_OK, rc_x = CD_OK;
    bool need_general_l = !fast_path, need_general_x = !fast_path;
    for (int redo = 0;; ++redo) {
        if (redo) { int rc = enqueue_sort(true); if (!rc) rc = enqueue_rest(); if (rc) return late(rc); }   // this rank's sort in its next form, then everything that follows it
        const bool serial = (m->flags & CD_MULTI_CROSS_SERIAL) != 0;                        // A/B: the cross pass behind the local one, on its stream
        hipStream_t cs = serial ? s : m->xstream;
        if (fast_path && recvd) {
            if (redo || serial) LATE_HIP(hipMemsetAsync(t1.d_state, 0, sizeof(TravState), cs));   // (first time round: zeroed at the start of the step, see there)
            LATE_HIP(hipStreamWaitEvent(cs, serial ? m->ev_payload : m->ev_tree, 0));
            QuerySrc srcx{c->d_leaf, c->d_boxes, c->d_qbox, c->d_root, m->d_recv, nullptr, c->d_os_ticket + 8};
            m->scratch_pairs.resize(2 * (size_t)spec1 + 2);
            int rc;
            {
                struct OnStream { cd_ctx *c; hipStream_t keep; OnStream(cd_ctx *c_, hipStream_t st) : c(c_), keep(c_->stream) { c->stream = st; c->quiet_pass = true; }
                                  ~OnStream() { c->stream = keep; c->quiet_pass = false; } } on(c, cs);
                launch_pass<true, false>(c, t1, srcx, (uint32_t)recvd, cap);
                rc = enqueue_report(c, t1, pairs != nullptr, spec1);
            }
            if (rc) return late(rc);
        }
        mark(ME_CROSS, cs);
        LATE_HIP(hipEventRecord(m->ev_cross, m->xstream));
        LATE_HIP(hipStreamWaitEvent(s, m->ev_cross, 0));
        LATE_HIP(hipStreamSynchronize(s)); ++syncs;                                          // host synchronisation 2 of 2 (both streams)
        LATE_HIP(hipGetLastError());
        if (fast_path) std::memcpy(c->sort_flags, reinterpret_cast<const Report *>(t0.h_report)->sort_flags, sizeof c->sort_flags);
        const int js = judge_sort_flags(c);                                                // (escalates c->sort_mode when a run was too long for this form)
        if (js == SORT_REDO && redo < SORT_REDO_MAX) continue;
        if (js != CD_OK) return late(js == SORT_REDO ? CD_ERR_SORT : js);
        break;
    }
    c->stage_events = se;
    c->stage = ST_REFIT;                                                      // (only now: the sort's flags have been judged)
    std::memcpy(c->root_box_host, m->h_roots + 6 * (size_t)me, sizeof(double) * 6); c->root_box_valid = true;
    if (fast_path) {
        HostCounters h0 = {}, h1 = {};
        parse_report(c, t0, h0, pairs, spec0);
        need_general_l = shards_overflowed(t0, h0) || h0.n_deferred > 0;
        if (!need_general_l) {
            n_local = h0.n_pairs; tested += h0.pairs_tested;
            const uint64_t ncopy = std::min<uint64_t>(n_local, cap);
            if (pairs && ncopy > spec0) { LATE_HIP(hipMemcpy(pairs + 2 * spec0, t0.d_pairs + 2 * spec0, sizeof(uint32_t) * 2 * (ncopy - spec0), hipMemcpyDeviceToHost)); ++syncs; }
            rc_l = n_local > cap ? CD_OVERFLOW : CD_OK;
        }
        if (recvd) {
            parse_report(c, t1, h1, m->scratch_pairs.data(), spec1);
            need_general_x = shards_overflowed(t1, h1) || h1.n_deferred > 0;
        } else need_general_x = false;
        if (recvd && !need_general_x && !need_general_l) {
            n_cross = h1.n_pairs; tested += h1.pairs_tested;
            const uint64_t room = cap > std::min<uint64_t>(n_local, cap) ? cap - std::min<uint64_t>(n_local, cap) : 0;
            const uint64_t ncopy = std::min<uint64_t>(n_cross, room);
            const uint64_t from_spec = std::min<uint64_t>(ncopy, spec1);
            if (pairs && from_spec) std::memcpy(pairs + 2 * n_local, m->scratch_pairs.data(), sizeof(uint32_t) * 2 * from_spec);
            if (pairs && ncopy > from_spec) { LATE_HIP(hipMemcpy(pairs + 2 * (n_local + from_spec), t1.d_pairs + 2 * from_spec, sizeof(uint32_t) * 2 * (ncopy - from_spec), hipMemcpyDeviceToHost)); ++syncs; }
            rc_x = n_cross > room ? CD_OVERFLOW : CD_OK;
        } else if (recvd && !need_general_x) need_general_x = true;     // the local pass is redone below: append the cross pairs after its final count
    }
    // general path (rare: a candidate shard overflowed, or a stack overflowed into the deep pass): the pass that needs it
    // runs again on its own, with the retries and the deep pass of run_traversal
    if (need_general_l) {
        uint64_t nl = 0;
        rc_l = run_traversal(c, t0, nullptr, 0, pairs, cap, &nl);
        if (rc_l < 0) return late(rc_l);
        n_local = nl; tested += c->stats.pairs_tested; syncs += 1;
    }
    if (need_general_x && recvd) {
        LATE_HIP(hipStreamWaitEvent(s, m->ev_payload, 0));
        const uint64_t used = std::min<uint64_t>(n_local, cap);
        uint64_t nx = 0;
        rc_x = run_traversal(c, t1, m->d_recv, recvd, pairs ? pairs + 2 * used : nullptr, cap - used, &nx);
        if (rc_x < 0) return late(rc_x);
        n_cross = nx; tested += c->stats.pairs_tested; syncs += 1;
    }
    // (the payload exchange has drained: the second stream was waited for above)
    c->stats.n_pairs = n_local + n_cross; c->stats.pairs_tested = tested;
    c->last_pairs_on_device = 0;                                              // two lists: cd_sorted_pairs does not apply to a multi step
    if (n_pairs) *n_pairs = n_local + n_cross;
    if (info) {
        info->world = (uint32_t)W; info->rank = (uint32_t)me; info->n_peers = n_peers; info->host_syncs = syncs; info->attempts = attempts + 1;
        info->sent_queries = sent; info->recv_queries = recvd; info->local_pairs = n_local; info->cross_pairs = n_cross; info->pairs_tested = tested;
        info->query_cap = m->qcap;
        if (timing) {
            auto el = [&](int a, int b) { float ms = 0.f; return hipEventElapsedTime(&ms, m->ev[a], m->ev[b]) == hipSuccess ? ms : -1.f; };
            info->ms_allgather = el(ME_START, ME_GATHER); info->ms_pack = el(ME_GATHER, ME_PACK); info->ms_counts = el(ME_PACK, ME_COUNTS);
            info->ms_tree = el(ME_LOC0, ME_TREE);
            hipEventSynchronize(m->ev[ME_XCH1]);
            info->ms_exchange = el(ME_XCH0, ME_XCH1);
            if (fast_path && !need_general_l && !need_general_x) { info->ms_local = el(ME_TREE, ME_LOCAL); info->ms_cross = el(ME_XCH1, ME_CROSS); }
            else { info->ms_local = -1.f; info->ms_cross = -1.f; }
        }
    }
    return (rc_l == CD_OVERFLOW || rc_x == CD_OVERFLOW) ? CD_OVERFLOW : CD_OK;
#undef LATE_HIP
#undef SOFT_HIP
}

}  // extern "C"
