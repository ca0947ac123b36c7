// TEST SCAFFOLDING: a stand-in for librccl.so whose ranks are THREADS of one process sharing one GPU.
//
// clsim_amd/csrc/comm.cpp loads RCCL by name at run time and honours CLSIMHIP_RCCL_LIBRARY; tests/test_comm_fake_rccl.py
// points that variable at this library (in a child process) so that the multi-rank branch of clsimhip_gather_hits -- the
// count all-gather, the grouped ncclSend / ncclRecv pairs, the overflow decision -- executes on the one GPU a test box
// has.  It implements the eleven entry points comm.cpp binds, with RCCL's calling rules checked rather than assumed:
//   * ncclCommInitRank is collective (returns when all ranks of the id have joined),
//   * ncclAllGather is collective, ordered after the work queued on the caller's stream,
//   * ncclSend / ncclRecv must pair up (peer, byte count) -- a send nobody receives, a receive nobody sends, or a size
//     mismatch is reported as an error after a timeout instead of hanging the GPU as the real library would,
//   * ncclGroupStart / ncclGroupEnd must balance; transfers posted inside a group start at GroupEnd.
// Data moves with hipMemcpyAsync (device to device) on the receiver's stream.  Nothing here is fast or clever.
//
// Second mode, FAKE_RCCL_DIR=<directory under /dev/shm>: the ranks are PROCESSES that time-share one GPU (bench.py's
// rehearsal of its --gpus N path on a one-GPU box, tests/test_bench_multi_rank_rehearsal.py).  Payloads then travel
// through files in that directory (device -> file by the sender, file -> device by the receiver); the rules checked are
// the same (everybody joins, an all-gather needs every rank, a receive needs its send with the same byte count).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

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

// ---- process mode: files in FAKE_RCCL_DIR ----
const char *shared_dir() { static const char *d = std::getenv("FAKE_RCCL_DIR"); return (d && d[0]) ? d : nullptr; }

bool write_file_atomically(const std::string &path, const void *data, size_t bytes)
{
    const std::string tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = (bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes);
    std::fclose(f);
    return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}

bool wait_for_file(const std::string &path, size_t *bytes)
{
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    struct stat st;
    while (stat(path.c_str(), &st) != 0) {
        if (std::chrono::steady_clock::now() > deadline) return false;
        usleep(200);
    }
    if (bytes) *bytes = static_cast<size_t>(st.st_size);
    return true;
}

bool read_file(const std::string &path, void *data, size_t bytes)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = (bytes == 0 || std::fread(data, 1, bytes, f) == bytes);
    std::fclose(f);
    return ok;
}

struct FileComm {
    std::string dir;
    int rank = 0, n = 0;
    uint64_t ag_round = 0;
    std::map<int, uint64_t> sent, received;            // per peer: transfers so far
};

ncclResult_t file_run_ops(std::vector<Op> &ops);

ncclResult_t run_ops(std::vector<Op> &ops)
{
    if (shared_dir()) return file_run_ops(ops);
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

ncclResult_t file_run_ops(std::vector<Op> &ops)
{
    for (Op &o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return fail("hipStreamSynchronize");
    ncclResult_t result = ncclSuccess;
    std::vector<uint8_t> host;
    for (Op &o : ops)
        if (o.send) {
            FileComm *c = reinterpret_cast<FileComm *>(o.comm);
            host.resize(o.bytes);
            if (o.bytes && hipMemcpy(host.data(), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("download for ncclSend");
            const std::string path = c->dir + "/p2p." + std::to_string(c->rank) + "." + std::to_string(o.peer) + "." + std::to_string(c->sent[o.peer]++);
            if (!write_file_atomically(path, host.data(), o.bytes)) return fail("ncclSend: cannot write the transfer file");
        }
    for (Op &o : ops)
        if (!o.send) {
            FileComm *c = reinterpret_cast<FileComm *>(o.comm);
            const std::string path = c->dir + "/p2p." + std::to_string(o.peer) + "." + std::to_string(c->rank) + "." + std::to_string(c->received[o.peer]++);
            size_t bytes = 0;
            if (!wait_for_file(path, &bytes)) { result = fail("ncclRecv: the peer never posted the matching ncclSend"); continue; }
            if (bytes != o.bytes) { result = fail("ncclSend / ncclRecv byte counts differ"); unlink(path.c_str()); continue; }
            host.resize(bytes);
            if (!read_file(path, host.data(), bytes)) { result = fail("ncclRecv: cannot read the transfer file"); continue; }
            unlink(path.c_str());
            if (bytes && (hipMemcpyAsync(o.dst, host.data(), bytes, hipMemcpyHostToDevice, o.stream) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess))
                result = fail("upload for ncclRecv");
        }
    return result;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (shared_dir()) {
        std::memset(id, 0, sizeof *id);
        std::snprintf(id->internal, sizeof id->internal, "fake-rccl-%ld-%llu", static_cast<long>(getpid()), static_cast<unsigned long long>(g_next_id++));
        return ncclSuccess;
    }
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "fake-rccl-%llu", static_cast<unsigned long long>(g_next_id++));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (shared_dir()) {
        FileComm *c = new FileComm;
        c->dir = std::string(shared_dir()) + "/" + std::string(id.internal, strnlen(id.internal, sizeof id.internal));
        c->rank = rank; c->n = nranks;
        mkdir(c->dir.c_str(), 0700);                    // (every rank tries; the first one wins)
        if (!write_file_atomically(c->dir + "/joined." + std::to_string(rank), &nranks, sizeof nranks)) return fail("ncclCommInitRank: cannot write into FAKE_RCCL_DIR");
        for (int r = 0; r < nranks; ++r) {
            size_t bytes = 0;
            int theirs = 0;
            if (!wait_for_file(c->dir + "/joined." + std::to_string(r), &bytes) || !read_file(c->dir + "/joined." + std::to_string(r), &theirs, sizeof theirs))
                return fail("ncclCommInitRank: not every rank joined");
            if (theirs != nranks) return fail("ncclCommInitRank: ranks disagree about the world size");
        }
        *comm = reinterpret_cast<ncclComm_t>(c);
        return ncclSuccess;
    }
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
    if (shared_dir()) delete reinterpret_cast<FileComm *>(comm);
    else delete reinterpret_cast<FakeComm *>(comm);
    return ncclSuccess;
}

// FAKE_RCCL_LIE_ABOUT_COUNT=1: the communicator reports one rank fewer than joined (the library must refuse it)
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    if (!comm || !count) return ncclInvalidArgument;
    *count = shared_dir() ? reinterpret_cast<const FileComm *>(comm)->n : reinterpret_cast<const FakeComm *>(comm)->world->n;
    if (const char *e = std::getenv("FAKE_RCCL_LIE_ABOUT_COUNT")) if (e[0] == '1') --*count;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank)
{
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = shared_dir() ? reinterpret_cast<const FileComm *>(comm)->rank : reinterpret_cast<const FakeComm *>(comm)->rank;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    const size_t bytes = sendcount * type_size(datatype);
    if (!comm || !bytes) return ncclInvalidArgument;
    if (t_group_depth) return fail("ncclAllGather inside a group is not modelled");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail("hipStreamSynchronize");
    if (shared_dir()) {
        FileComm *fc = reinterpret_cast<FileComm *>(comm);
        const uint64_t round = fc->ag_round++;
        std::vector<uint8_t> host(bytes);
        if (hipMemcpy(host.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("download for ncclAllGather");
        if (!write_file_atomically(fc->dir + "/ag." + std::to_string(round) + "." + std::to_string(fc->rank), host.data(), bytes)) return fail("ncclAllGather: cannot write");
        std::vector<uint8_t> all(bytes * static_cast<size_t>(fc->n));
        for (int r = 0; r < fc->n; ++r) {
            size_t got = 0;
            const std::string path = fc->dir + "/ag." + std::to_string(round) + "." + std::to_string(r);
            if (!wait_for_file(path, &got)) return fail("ncclAllGather: not every rank arrived");
            if (got != bytes) return fail("ncclAllGather: ranks disagree about the element count");
            if (!read_file(path, all.data() + bytes * static_cast<size_t>(r), bytes)) return fail("ncclAllGather: cannot read");
        }
        if (round >= 2) unlink((fc->dir + "/ag." + std::to_string(round - 2) + "." + std::to_string(fc->rank)).c_str());   // (everybody is past that round)
        if (hipMemcpyAsync(recvbuff, all.data(), all.size(), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            return fail("upload for ncclAllGather");
        return ncclSuccess;
    }
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
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
    if (shared_dir()) {
        FileComm *fc = reinterpret_cast<FileComm *>(comm);
        if (!fc || peer < 0 || peer >= fc->n || peer == fc->rank) return ncclInvalidArgument;
    } else if (!c || peer < 0 || peer >= c->world->n || peer == c->rank) return ncclInvalidArgument;
    Op o{true, sendbuff, nullptr, count * type_size(datatype), peer, c, stream};
    if (t_group_depth) { t_ops.push_back(o); return ncclSuccess; }
    std::vector<Op> one{o};
    return run_ops(one);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    if (shared_dir()) {
        FileComm *fc = reinterpret_cast<FileComm *>(comm);
        if (!fc || peer < 0 || peer >= fc->n || peer == fc->rank) return ncclInvalidArgument;
    } else if (!c || peer < 0 || peer >= c->world->n || peer == c->rank) return ncclInvalidArgument;
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
