// Multi-GPU gather of detected photons behind the C ABI (BASELINE.json: "RCCL gather of hit photons
// over xGMI").  The propagation itself shards with no exchange -- contiguous step ranges per GPU, every GPU with its
// own RNG streams and photon buffer -- so the only collective is the gather of the results on one rank:
//   1. ncclAllGather of the ranks' hit counts (4 bytes each),
//   2. one point-to-point transfer per peer inside ncclGroupStart/End: 7 peers -> 7 distinct xGMI links, no ring
//      (a ring all-gather would push every rank's photons over every link).
// The reference has no counterpart: its multi-device model is a vector of independent converters behind
// I3CLSimServer (private/clsim/I3CLSimServer.cxx:77-137), results collected by host threads.
//
// RCCL is loaded at run time (dlopen), not linked: a host process that already carries an RCCL (PyTorch ships one)
// keeps its single instance, and a host without multi-GPU use never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "converter.h"

namespace clsimhip {

namespace {
struct Rccl {
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
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

const Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    static std::string error;
    std::call_once(once, [] {
        std::vector<std::string> names;
        if (const char *e = std::getenv("CLSIMHIP_RCCL_LIBRARY")) names.push_back(e);
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        for (const std::string &n : names) {
            r.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) { error = std::string("cannot load RCCL: ") + dlerror(); return; }
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p && error.empty()) error = std::string("RCCL symbol missing: ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!error.empty()) throw Error(CLSIMHIP_ERR_DEVICE, error);
    return r;
}

void nccl_check(ncclResult_t e, const char *what)
{
    if (e != ncclSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(what) + ": " + rccl().GetErrorString(e));
}
void hip_ok(hipError_t e, const char *what)
{
    if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
} // namespace

struct Comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    // what a rank announces: {hit counter, 0, room in its gather buffer (low, high word)}; the root's room decides, on
    // every rank alike, how much travels
    static constexpr size_t kWords = 4;
    uint32_t *d_mine = nullptr;         // kWords, device
    uint32_t *d_all = nullptr;          // world x kWords, device
    uint32_t *h_mine = nullptr;         // kWords, pinned
    uint32_t *h_all = nullptr;          // world x kWords, pinned
    hipEvent_t counted = nullptr;
    // what the COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank after ncclCommInitRank), not what the
    // caller passed in: a multi-GPU record quotes these
    int rccl_ranks = 0, rccl_rank = -1;
    char pci_bus_id[32] = {0};
    // per-gather timing on the caller's stream (events resolved lazily, in comm_statistics)
    struct Timed { hipEvent_t begin, end; };
    std::vector<Timed> in_flight, spare;
    uint64_t gathers = 0, records_sent = 0, records_received = 0;
    double gather_ms = 0.0;
    std::mutex mutex;

    void resolve_timers()
    {
        for (Timed &t : in_flight) {
            float ms = 0.f;
            if (hipEventSynchronize(t.end) == hipSuccess && hipEventElapsedTime(&ms, t.begin, t.end) == hipSuccess) gather_ms += ms;
            spare.push_back(t);
        }
        in_flight.clear();
    }
    Timed take_timer()
    {
        if (in_flight.size() >= 256) resolve_timers();
        if (!spare.empty()) { Timed t = spare.back(); spare.pop_back(); return t; }
        Timed t{nullptr, nullptr};
        hip_ok(hipEventCreate(&t.begin), "hipEventCreate");
        hip_ok(hipEventCreate(&t.end), "hipEventCreate");
        return t;
    }
};

void comm_unique_id(uint8_t id[CLSIMHIP_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == CLSIMHIP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    nccl_check(rccl().GetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, &u, sizeof u);
}

Comm *comm_create(int device, int rank, int world, const uint8_t id[CLSIMHIP_UNIQUE_ID_BYTES])
{
    if (world < 1 || rank < 0 || rank >= world) throw Error(CLSIMHIP_ERR_ARGUMENT, "rank / world size out of range");
    if (!id) throw Error(CLSIMHIP_ERR_ARGUMENT, "unique id is (null)");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available");
    if (device < 0 || device >= count) throw Error(CLSIMHIP_ERR_ARGUMENT, "device ordinal out of range");
    DeviceGuard on_device(device);
    std::unique_ptr<Comm> c(new Comm);
    c->device = device; c->rank = rank; c->world = world;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    nccl_check(rccl().CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank");
    // the communicator's own view must be the caller's: a record that says "N ranks" quotes ncclCommCount
    nccl_check(rccl().CommCount(c->comm, &c->rccl_ranks), "ncclCommCount");
    nccl_check(rccl().CommUserRank(c->comm, &c->rccl_rank), "ncclCommUserRank");
    if (c->rccl_ranks != world || c->rccl_rank != rank) {
        const std::string what = "the RCCL communicator reports rank " + std::to_string(c->rccl_rank) + " of " + std::to_string(c->rccl_ranks) +
                                 ", asked for rank " + std::to_string(rank) + " of " + std::to_string(world);
        (void)rccl().CommDestroy(c->comm);
        throw Error(CLSIMHIP_ERR_DEVICE, what);
    }
    if (hipDeviceGetPCIBusId(c->pci_bus_id, static_cast<int>(sizeof c->pci_bus_id), device) != hipSuccess) c->pci_bus_id[0] = 0;
    const size_t words = Comm::kWords * static_cast<size_t>(world);
    hip_ok(hipMalloc(reinterpret_cast<void **>(&c->d_mine), sizeof(uint32_t) * Comm::kWords), "hipMalloc");
    hip_ok(hipMalloc(reinterpret_cast<void **>(&c->d_all), sizeof(uint32_t) * words), "hipMalloc");
    hip_ok(hipHostMalloc(reinterpret_cast<void **>(&c->h_mine), sizeof(uint32_t) * Comm::kWords, hipHostMallocDefault), "hipHostMalloc");
    hip_ok(hipHostMalloc(reinterpret_cast<void **>(&c->h_all), sizeof(uint32_t) * words, hipHostMallocDefault), "hipHostMalloc");
    hip_ok(hipEventCreateWithFlags(&c->counted, hipEventDisableTiming), "hipEventCreate");
    return c.release();
}

void comm_info(Comm *c, int *ranks, int *rank, int *device, char *pci_bus_id, size_t pci_bytes)
{
    if (!c) throw Error(CLSIMHIP_ERR_ARGUMENT, "communicator is (null)");
    if (ranks) *ranks = c->rccl_ranks;
    if (rank) *rank = c->rccl_rank;
    if (device) *device = c->device;
    if (pci_bus_id && pci_bytes) {
        std::strncpy(pci_bus_id, c->pci_bus_id, pci_bytes - 1);
        pci_bus_id[pci_bytes - 1] = 0;
    }
}

// Blocks until the gathers issued so far have finished on their streams.
void comm_statistics(Comm *c, uint64_t *gathers, double *gather_ms, uint64_t *records_sent, uint64_t *records_received, bool reset)
{
    if (!c) throw Error(CLSIMHIP_ERR_ARGUMENT, "communicator is (null)");
    DeviceGuard on_device(c->device);
    std::lock_guard<std::mutex> lock(c->mutex);
    c->resolve_timers();
    if (gathers) *gathers = c->gathers;
    if (gather_ms) *gather_ms = c->gather_ms;
    if (records_sent) *records_sent = c->records_sent;
    if (records_received) *records_received = c->records_received;
    if (reset) { c->gathers = c->records_sent = c->records_received = 0; c->gather_ms = 0.0; }
}

void comm_destroy(Comm *c)
{
    if (!c) return;
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    (void)hipFree(c->d_mine);
    (void)hipFree(c->d_all);
    if (c->h_mine) (void)hipHostFree(c->h_mine);
    if (c->h_all) (void)hipHostFree(c->h_all);
    if (c->counted) (void)hipEventDestroy(c->counted);
    c->resolve_timers();
    for (Comm::Timed &t : c->spare) { (void)hipEventDestroy(t.begin); (void)hipEventDestroy(t.end); }
    if (previous >= 0) (void)hipSetDevice(previous);
    delete c;
}

// Blocks the calling thread only until the hit counts are known (the kernel that produced them must have finished
// anyway); the payload transfers are left running on `stream`.
//
// Every decision about what travels is taken from all-gathered numbers that every rank holds -- the ranks' hit counters
// and the room in the ROOT's gather buffer -- so the sends and receives always pair up.  A gather buffer that is too
// small is not an error one rank may throw on its own (its peers would post sends that nobody receives, and an RCCL send
// kernel spins on the GPU until it is matched): the root receives the prefix that fits, rank by rank, and then EVERY
// rank reports CLSIMHIP_ERR_ARGUMENT.  The same goes for unequal photon buffer capacities: a rank announces what it
// stored, not its counter.
void comm_gather_hits(Comm *c, const void *d_photons, const void *d_hit_count, size_t capacity, int root, void *d_gathered,
                      size_t gathered_capacity, uint64_t *counts_out, hipStream_t stream)
{
    if (!c) throw Error(CLSIMHIP_ERR_ARGUMENT, "communicator is (null)");
    if (!d_photons || !d_hit_count) throw Error(CLSIMHIP_ERR_ARGUMENT, "device pointers are (null)");
    if (root < 0 || root >= c->world) throw Error(CLSIMHIP_ERR_ARGUMENT, "root rank out of range");
    // a root without a buffer announces room for nothing: the collective below still pairs up and every rank gets the error
    if (c->rank == root && !d_gathered) gathered_capacity = 0;
    DeviceGuard on_device(c->device);
    const Rccl &R = rccl();
    const size_t W = Comm::kWords;
    std::lock_guard<std::mutex> lock(c->mutex);
    const Comm::Timed timer = c->take_timer();
    hip_ok(hipEventRecord(timer.begin, stream), "event");
    // (whatever happens below, the pair is closed and accounted for)
    struct CloseTimer {
        Comm *c; Comm::Timed t; hipStream_t s;
        ~CloseTimer() { (void)hipEventRecord(t.end, s); c->in_flight.push_back(t); ++c->gathers; }
    } close_timer{c, timer, stream};
    c->h_mine[0] = 0;
    c->h_mine[1] = static_cast<uint32_t>(std::min<size_t>(capacity, 0xffffffffu));
    c->h_mine[2] = static_cast<uint32_t>(static_cast<uint64_t>(gathered_capacity) & 0xffffffffu);
    c->h_mine[3] = static_cast<uint32_t>(static_cast<uint64_t>(gathered_capacity) >> 32);
    hip_ok(hipMemcpyAsync(c->d_mine, c->h_mine, sizeof(uint32_t) * W, hipMemcpyHostToDevice, stream), "upload gather header");
    hip_ok(hipMemcpyAsync(c->d_mine, d_hit_count, sizeof(uint32_t), hipMemcpyDeviceToDevice, stream), "copy hit counter");
    nccl_check(R.AllGather(c->d_mine, c->d_all, W, ncclUint32, c->comm, stream), "ncclAllGather (hit counts)");
    hip_ok(hipMemcpyAsync(c->h_all, c->d_all, sizeof(uint32_t) * W * static_cast<size_t>(c->world), hipMemcpyDeviceToHost, stream), "download hit counts");
    hip_ok(hipEventRecord(c->counted, stream), "event");
    hip_ok(hipEventSynchronize(c->counted), "hit counts");
    // The kernel's counter keeps counting past the capacity of the photon buffer (propagation_kernel.c.cl:329-334);
    // a rank sends what it stored, and no more than still fits on the root.
    const uint64_t room = static_cast<uint64_t>(c->h_all[W * root + 2]) | (static_cast<uint64_t>(c->h_all[W * root + 3]) << 32);
    std::vector<size_t> travels(static_cast<size_t>(c->world));
    uint64_t total = 0, accepted = 0;
    for (int r = 0; r < c->world; ++r) {
        const uint64_t stored = std::min<uint64_t>(c->h_all[W * r], c->h_all[W * r + 1]);
        if (counts_out) counts_out[r] = c->h_all[W * r];
        travels[r] = static_cast<size_t>(std::min<uint64_t>(stored, room - accepted));
        accepted += travels[r];
        total += stored;
        if (r != root && c->rank == root) c->records_received += travels[r];
        if (r == c->rank && c->rank != root) c->records_sent += travels[r];
    }
    constexpr size_t kRecord = sizeof(clsimhip_photon);
    if (c->world > 1) {
        nccl_check(R.GroupStart(), "ncclGroupStart");
        ncclResult_t failed = ncclSuccess;
        const char *where = "";
        if (c->rank == root) {
            size_t offset = 0;
            for (int r = 0; r < c->world && failed == ncclSuccess; ++r) {
                if (r != root && travels[r] != 0) {
                    failed = R.Recv(static_cast<uint8_t *>(d_gathered) + offset * kRecord, travels[r] * kRecord, ncclUint8, r, c->comm, stream);
                    where = "ncclRecv";
                }
                offset += travels[r];
            }
        } else if (travels[c->rank] != 0) {
            failed = R.Send(d_photons, travels[c->rank] * kRecord, ncclUint8, root, c->comm, stream);
            where = "ncclSend";
        }
        const ncclResult_t closed = R.GroupEnd();          // a group that was opened is always closed
        nccl_check(failed, where);
        nccl_check(closed, "ncclGroupEnd");
    }
    if (c->rank == root && travels[root] != 0) {
        size_t offset = 0;
        for (int r = 0; r < root; ++r) offset += travels[r];
        hip_ok(hipMemcpyAsync(static_cast<uint8_t *>(d_gathered) + offset * kRecord, d_photons, travels[root] * kRecord, hipMemcpyDeviceToDevice, stream),
               "copy the root's own photons");
    }
    if (total > room)
        throw Error(CLSIMHIP_ERR_ARGUMENT, "gather buffer too small for the detected photons of all ranks: " + std::to_string(total) +
                                               " stored, room for " + std::to_string(room) + " on the root (it received the first " +
                                               std::to_string(accepted) + ", rank by rank)");
}

} // namespace clsimhip
