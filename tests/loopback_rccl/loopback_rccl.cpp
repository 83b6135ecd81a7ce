// TEST INFRASTRUCTURE (not part of the product): an in-process stand-in for librccl.so, so that cd_multi_step's
// world > 1 protocol -- slab offsets, count-matrix indexing, the collective redo decisions, send/recv matching -- can be
// exercised on the ONE GPU a test box has.  RCCL itself refuses two ranks on one device; here every "rank" is a host
// THREAD of one process with its own cd_ctx, and a collective is a thread rendezvous + device-to-device copies enqueued
// on the callers' streams with the stream semantics of the real calls (data is read after the producer's stream reached
// the call, and a buffer is free for reuse once the caller's stream has passed the call).
// libmi355cd.so loads it instead of RCCL when MI355CD_RCCL_LIBRARY names it (cd_multi.h).  Only the ten entry points
// cd_multi.h resolves are provided.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct Msg {
    const void *ptr; size_t bytes; hipEvent_t ready; hipEvent_t done = nullptr; bool completed = false, mismatch = false;
};

struct Group {
    int world = 0, joined = 0, left = 0;
    std::mutex mu; std::condition_variable cv;
    // barrier
    int arrived = 0; unsigned long long generation = 0;
    // all-gather slots
    std::vector<const void *> send; std::vector<size_t> bytes; std::vector<hipEvent_t> ready, done;
    // point-to-point mailboxes [src * world + dst]
    std::vector<std::deque<std::shared_ptr<Msg>>> box;

    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long g = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};

struct Comm { std::shared_ptr<Group> g; int rank; };

std::mutex g_mu;
std::map<unsigned long long, std::shared_ptr<Group>> g_groups;
unsigned long long g_next_id = 1;

struct Pending { bool is_send; void *buf; size_t bytes; int peer; Comm *comm; hipStream_t st; };
thread_local int t_depth = 0;
thread_local std::vector<Pending> t_ops;

size_t dtype_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

#define H(expr) do { if ((expr) != hipSuccess) return ncclUnhandledCudaError; } while (0)

ncclResult_t run_p2p(std::vector<Pending> &ops)
{
    std::vector<std::shared_ptr<Msg>> mine(ops.size());
    // 1. post every send (never blocks)
    for (size_t i = 0; i < ops.size(); ++i) {
        Pending &o = ops[i];
        if (!o.is_send) continue;
        auto msg = std::make_shared<Msg>();
        msg->ptr = o.buf; msg->bytes = o.bytes;
        H(hipEventCreateWithFlags(&msg->ready, hipEventDisableTiming));
        H(hipEventRecord(msg->ready, o.st));
        Group &g = *o.comm->g;
        { std::lock_guard<std::mutex> lk(g.mu); g.box[(size_t)o.comm->rank * g.world + o.peer].push_back(msg); }
        g.cv.notify_all();
        mine[i] = msg;
    }
    // 2. every receive: wait for its message, copy on the receiver's stream
    ncclResult_t res = ncclSuccess;
    for (Pending &o : ops) {
        if (o.is_send) continue;
        Group &g = *o.comm->g;
        std::shared_ptr<Msg> msg;
        {
            std::unique_lock<std::mutex> lk(g.mu);
            auto &q = g.box[(size_t)o.peer * g.world + o.comm->rank];
            g.cv.wait(lk, [&] { return !q.empty(); });
            msg = q.front(); q.pop_front();
        }
        hipEvent_t done = nullptr;
        if (msg->bytes != o.bytes) { msg->mismatch = true; res = ncclInvalidArgument; }      // a real RCCL would hang or corrupt: make it loud
        else {
            H(hipStreamWaitEvent(o.st, msg->ready, 0));
            H(hipMemcpyAsync(o.buf, msg->ptr, o.bytes, hipMemcpyDeviceToDevice, o.st));
        }
        H(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        H(hipEventRecord(done, o.st));
        { std::lock_guard<std::mutex> lk(g.mu); msg->done = done; msg->completed = true; }
        g.cv.notify_all();
    }
    // 3. every send returns once its receiver has enqueued the copy; the sender's stream then waits for that copy
    for (size_t i = 0; i < ops.size(); ++i) {
        Pending &o = ops[i];
        if (!o.is_send) continue;
        Group &g = *o.comm->g;
        { std::unique_lock<std::mutex> lk(g.mu); g.cv.wait(lk, [&] { return mine[i]->completed; }); }
        if (mine[i]->mismatch) res = ncclInvalidArgument;
        H(hipStreamWaitEvent(o.st, mine[i]->done, 0));
        // (events are leaked on purpose: destroying an event another stream may still be waiting on is the kind of bug this
        //  stand-in must not add; a test run creates a few thousand)
    }
    return res;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof *id);
    std::lock_guard<std::mutex> lk(g_mu);
    const unsigned long long v = g_next_id++;
    std::memcpy(id->internal, &v, sizeof v);
    std::memcpy(id->internal + 8, "loopback", 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    unsigned long long v; std::memcpy(&v, id.internal, sizeof v);
    if (std::memcmp(id.internal + 8, "loopback", 8) != 0 || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<Group> g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto &slot = g_groups[v];
        if (!slot) {
            slot = std::make_shared<Group>();
            slot->world = nranks;
            slot->send.assign(nranks, nullptr); slot->bytes.assign(nranks, 0);
            slot->ready.assign(nranks, nullptr); slot->done.assign(nranks, nullptr);
            slot->box.resize((size_t)nranks * nranks);
        }
        g = slot;
    }
    if (g->world != nranks) return ncclInvalidArgument;
    H(hipEventCreateWithFlags(&g->ready[rank], hipEventDisableTiming));
    H(hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming));
    Comm *c = new Comm{g, rank};
    {   // like the real call: returns once every rank of the communicator has joined
        std::unique_lock<std::mutex> lk(g->mu);
        ++g->joined; g->cv.notify_all();
        g->cv.wait(lk, [&] { return g->joined >= g->world; });
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<Comm *>(comm); return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int *n) { *n = reinterpret_cast<Comm *>(comm)->g->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *r) { *r = reinterpret_cast<Comm *>(comm)->rank; return ncclSuccess; }

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    Group &g = *c->g;
    const size_t bytes = count * dtype_size(dt);
    const int me = c->rank;
    g.send[me] = sendbuff; g.bytes[me] = bytes;
    H(hipEventRecord(g.ready[me], st));
    g.barrier();                                            // every rank has published its buffer and its "data ready" event
    ncclResult_t res = ncclSuccess;
    for (int p = 0; p < g.world; ++p) {
        if (g.bytes[p] != bytes) { res = ncclInvalidArgument; continue; }
        char *dst = static_cast<char *>(recvbuff) + (size_t)p * bytes;
        if (p == me && dst == sendbuff) continue;           // in place
        H(hipStreamWaitEvent(st, g.ready[p], 0));
        H(hipMemcpyAsync(dst, g.send[p], bytes, hipMemcpyDeviceToDevice, st));
    }
    H(hipEventRecord(g.done[me], st));
    g.barrier();                                            // every rank has enqueued its copies
    for (int p = 0; p < g.world; ++p) H(hipStreamWaitEvent(st, g.done[p], 0));   // my send buffer is free once every reader is through
    g.barrier();                                            // slots and events may be reused
    return res;
}

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Pending> ops; ops.swap(t_ops);
    return run_p2p(ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (peer < 0 || peer >= c->g->world) return ncclInvalidArgument;
    t_ops.push_back(Pending{true, const_cast<void *>(buf), count * dtype_size(dt), peer, c, st});
    if (t_depth == 0) { std::vector<Pending> ops; ops.swap(t_ops); return run_p2p(ops); }
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (peer < 0 || peer >= c->g->world) return ncclInvalidArgument;
    t_ops.push_back(Pending{false, buf, count * dtype_size(dt), peer, c, st});
    if (t_depth == 0) { std::vector<Pending> ops; ops.swap(t_ops); return run_p2p(ops); }
    return ncclSuccess;
}

}  // extern "C"
