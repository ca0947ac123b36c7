// TEST SCAFFOLDING: a stand-in for librccl.so whose ranks are THREADS of one process sharing one GPU.
//
// clsim_amd/csrc/comm.cpp loads RCCL by name at run time and honours CLSIMHIP_RCCL_LIBRARY; tests/test_comm_fake_rccl.py
// points that variable at this library (in a child process) so that the multi-rank branch of clsimhip_gather_hits -- the
// count all-gather, the grouped ncclSend / ncclRecv pairs, the overflow decision -- executes on the one GPU a test box
// has.  It implements the nine entry points comm.cpp binds, with RCCL's calling rules checked rather than assumed:
//   * ncclCommInitRank is collective (returns when all ranks of the id have joined),
//   * ncclAllGather is collective, ordered after the work queued on the caller's stream,
//   * ncclSend / ncclRecv must pair up (peer, byte count) -- a send nobody receives, a receive nobody sends, or a size
//     mismatch is reported as an error after a timeout instead of hanging the GPU as the real library would,
//   * ncclGroupStart / ncclGroupEnd must balance; transfers posted inside a group start at GroupEnd.
// Data moves with hipMemcpyAsync (device to device) on the receiver's stream.  Nothing here is fast or clever.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr auto kTimeout = std::chrono::seconds(20);

struct World {
    int n = 0;
    int joined = 0;
    // all-gather rendezvous
    std::vector<const void *> ag_src;
    int ag_arrived = 0, ag_done = 0;
    uint64_t ag_round = 0;
    // posted sends: (src, dst) -> {ptr, bytes, consumed}
    struct Posted { const void *ptr; size_t bytes; bool consumed; bool bad; };
    std::map<std::pair<int, int>, Posted> sends;
};

struct FakeComm {
    std::shared_ptr<World> world;
    int rank = 0;
};

std::mutex g_mutex;                                   // one lock for everything: this is a test double
std::condition_variable g_cv;
std::map<std::string, std::shared_ptr<World>> g_worlds;
std::atomic<uint64_t> g_next_id{1};
std::atomic<int> g_errors{0};

struct Op { bool send; const void *src; void *dst; size_t bytes; int peer; FakeComm *comm; hipStream_t stream; };
thread_local int t_group_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

ncclResult_t fail(const char *what)
{
    std::fprintf(stderr, "fake_rccl: %s\n", what);
    ++g_errors;
    return ncclInternalError;
}

ncclResult_t run_ops(std::vector<Op> &ops)
{
    // a transfer is ordered after the work already queued on the poster's stream
    for (Op &o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return fail("hipStreamSynchronize");
    std::unique_lock<std::mutex> lock(g_mutex);
    for (Op &o : ops)
        if (o.send) {
            auto key = std::make_pair(o.comm->rank, o.peer);
            if (o.comm->world->sends.count(key)) return fail("two sends to the same peer in flight");
            o.comm->world->sends[key] = World::Posted{o.src, o.bytes, false, false};
        }
    g_cv.notify_all();
    ncclResult_t result = ncclSuccess;
    for (Op &o : ops)
        if (!o.send) {
            World &w = *o.comm->world;
            auto key = std::make_pair(o.peer, o.comm->rank);
            if (!g_cv.wait_for(lock, kTimeout, [&] { return w.sends.count(key) && !w.sends[key].consumed; })) {
                result = fail("ncclRecv: the peer never posted the matching ncclSend");
                continue;
            }
            World::Posted &p = w.sends[key];
            if (p.bytes != o.bytes) {
                p.bad = true;
                result = fail("ncclSend / ncclRecv byte counts differ");
            } else {
                lock.unlock();
                const bool ok = hipMemcpyAsync(o.dst, p.ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) == hipSuccess &&
                                hipStreamSynchronize(o.stream) == hipSuccess;
                lock.lock();
                if (!ok) result = fail("device copy");
            }
            w.sends[key].consumed = true;
            g_cv.notify_all();
        }
    for (Op &o : ops)
        if (o.send) {
            World &w = *o.comm->world;
            auto key = std::make_pair(o.comm->rank, o.peer);
            if (!g_cv.wait_for(lock, kTimeout, [&] { return w.sends[key].consumed; }))
                result = fail("ncclSend: the peer never posted the matching ncclRecv");
            else if (w.sends[key].bad)
                result = ncclInternalError;
            w.sends.erase(key);
        }
    return result;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "fake-rccl-%llu", static_cast<unsigned long long>(g_next_id++));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::unique_lock<std::mutex> lock(g_mutex);
    const std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
    std::shared_ptr<World> &w = g_worlds[key];
    if (!w) { w = std::make_shared<World>(); w->n = nranks; w->ag_src.assign(static_cast<size_t>(nranks), nullptr); }
    if (w->n != nranks) return fail("ncclCommInitRank: ranks disagree about the world size");
    std::shared_ptr<World> world = w;
    ++world->joined;
    g_cv.notify_all();
    if (!g_cv.wait_for(lock, kTimeout, [&] { return world->joined >= world->n; })) return fail("ncclCommInitRank: not every rank joined");
    FakeComm *c = new FakeComm;
    c->world = world;
    c->rank = rank;
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<FakeComm *>(comm);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    const size_t bytes = sendcount * type_size(datatype);
    if (!c || !bytes) return ncclInvalidArgument;
    if (t_group_depth) return fail("ncclAllGather inside a group is not modelled");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail("hipStreamSynchronize");
    World &w = *c->world;
    std::unique_lock<std::mutex> lock(g_mutex);
    // (nobody leaves a round before every rank has finished it, so a rank that enters finds a fresh round)
    const uint64_t round = w.ag_round;
    w.ag_src[static_cast<size_t>(c->rank)] = sendbuff;
    ++w.ag_arrived;
    g_cv.notify_all();
    if (!g_cv.wait_for(lock, kTimeout, [&] { return w.ag_arrived >= w.n; })) return fail("ncclAllGather: not every rank arrived");
    std::vector<const void *> src = w.ag_src;
    lock.unlock();
    bool ok = true;
    for (int r = 0; r < w.n; ++r)
        ok = ok && hipMemcpyAsync(static_cast<uint8_t *>(recvbuff) + static_cast<size_t>(r) * bytes, src[static_cast<size_t>(r)], bytes,
                                  hipMemcpyDeviceToDevice, stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(stream) == hipSuccess;
    lock.lock();
    if (++w.ag_done == w.n) { w.ag_done = 0; w.ag_arrived = 0; ++w.ag_round; }
    g_cv.notify_all();
    // nobody may reuse its send buffer before every rank has copied it
    if (!g_cv.wait_for(lock, kTimeout, [&] { return w.ag_round != round; })) return fail("ncclAllGather: a rank did not finish");
    return ok ? ncclSuccess : fail("device copy");
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c || peer < 0 || peer >= c->world->n || peer == c->rank) return ncclInvalidArgument;
    Op o{true, sendbuff, nullptr, count * type_size(datatype), peer, c, stream};
    if (t_group_depth) { t_ops.push_back(o); return ncclSuccess; }
    std::vector<Op> one{o};
    return run_ops(one);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (!c || peer < 0 || peer >= c->world->n || peer == c->rank) return ncclInvalidArgument;
    Op o{false, nullptr, recvbuff, count * type_size(datatype), peer, c, stream};
    if (t_group_depth) { t_ops.push_back(o); return ncclSuccess; }
    std::vector<Op> one{o};
    return run_ops(one);
}

ncclResult_t ncclGroupStart()
{
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_group_depth <= 0) return fail("ncclGroupEnd without ncclGroupStart");
    if (--t_group_depth) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return ops.empty() ? ncclSuccess : run_ops(ops);
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument (fake rccl)" : "internal error (fake rccl)"; }

// test hooks: unbalanced groups left open by the caller's thread, errors seen so far
int fake_rccl_open_groups() { return t_group_depth; }
int fake_rccl_errors() { return g_errors.load(); }

} // extern "C"
