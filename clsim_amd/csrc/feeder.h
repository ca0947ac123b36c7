// Feeder: the worker thread of I3CLSimLightSourceToStepConverterAsync (private/clsim/I3CLSimLightSourceToStepConverterAsync.cxx
// WorkerThread_impl :178-392, EnqueueLightSource / EnqueueBarrier / GetConversionResultWithBarrierInfo :470-600).
#pragma once
#include <atomic>
#include <memory>
#include <thread>
#include <vector>

#include "converter.h"
#include "lightsource.h"
#include "step_store.h"

namespace clsimhip {

// The feeder's GPU step producer: one HIP stream of its own (non-blocking, highest priority: a propagator on the same GPU
// runs persistent grids, and a launch that has to share the chip with one gets slivers of it), device and page-locked host
// buffers that only grow.  (The stateless clsimhip_generate_steps allocates and frees device memory per call; hipFree
// waits for the whole device, i.e. for the propagation kernel that happens to be running.)
class StepProducer {
public:
    explicit StepProducer(int device) : device_(device) {}
    ~StepProducer();
    StepProducer(const StepProducer &) = delete;
    StepProducer &operator=(const StepProducer &) = delete;
    // the steps of `requests` (padded to `granularity`); the pointer stays valid until the next call
    const clsimhip_step *generate(const std::vector<clsimhip_step_request> &requests, uint64_t seed, size_t granularity, size_t &real, size_t &padded);
    // stream and buffers for `steps` steps and `requests` requests now, not inside the first light source's conversion (page-locking
    // 150 MB takes tens of milliseconds: the reference's benchmark flow waited for them in front of its first bunch).  Never throws:
    // without a device, or without the memory, the first generate() reports it.
    void reserve(size_t steps, size_t requests) noexcept;

private:
    void ensure(size_t steps, size_t requests);
    int device_;
    hipStream_t stream_ = nullptr;
    void *d_steps_ = nullptr, *d_req_ = nullptr, *d_first_ = nullptr;
    clsimhip_step *h_steps_ = nullptr;
    void *h_req_ = nullptr, *h_first_ = nullptr;
    size_t cap_steps_ = 0, cap_req_ = 0;
};

class Feeder {
public:
    struct Result {                         // the tuple the reference puts on queueFromGeant4_ (:222, :268)
        std::unique_ptr<std::vector<clsimhip_step>> steps;
        std::vector<uint32_t> finished;     // light sources whose steps have all left the store
        bool last_before_barrier = false;
    };
    // ppc may be null: the feeder then only accepts light sources that come with their steps (enqueue_steps)
    Feeder(const PPCConverter *ppc, int device, uint64_t seed, size_t max_bunch_size, size_t granularity, size_t queue_depth);
    ~Feeder();
    void enqueue_light_source(const clsimhip_particle &particle);
    void enqueue_steps(uint32_t identifier, const clsimhip_step *steps, size_t n);      // a source whose steps the caller made (a propagator's output)
    void enqueue_barrier();
    bool barrier_active() const { return barrier_enqueued_.load(); }
    bool more_steps_available() const { return !out_->empty(); }
    // false on timeout (timeout_ms < 0: wait for ever)
    bool get_result(double timeout_ms, Result &out);
    std::string worker_error() const;
    void check_worker() const;

private:
    struct Item {
        bool barrier = false, has_particle = false;
        clsimhip_particle particle{};
        uint32_t identifier = 0;
        std::vector<clsimhip_step> steps;
    };
    void worker();
    void flush(bool reset_barrier);
    void insert_and_flush(const clsimhip_step *steps, size_t n);

    const PPCConverter *ppc_;
    int device_;
    uint64_t seed_;
    size_t max_bunch_, granularity_;
    std::unique_ptr<BoundedQueue<Item>> in_;
    std::unique_ptr<BoundedQueue<Result>> out_;
    StepStore store_{0};
    std::deque<uint32_t> markers_;
    OccurrenceCounter occurrences_;     // worker thread only
    std::unique_ptr<StepProducer> producer_;    // worker thread only
    std::atomic<bool> barrier_enqueued_{false};
    std::thread thread_;
    mutable std::mutex error_mutex_;
    std::string error_;
};

} // namespace clsimhip
