// The feeder thread of the reference's asynchronous light-source converter, restated on this library's pieces.
// private/clsim/I3CLSimLightSourceToStepConverterAsync.cxx:
//   main loop :340-392      take a light source (or the barrier) from the input queue, flush, convert, push the marker
//   flushStepStore :209-273 bunches of maxBunchSize steps leave the store in ascending photon count, each with the
//                           identifiers of the light sources that have completely left it; at the barrier the rest goes
//                           out padded with no-op steps to ((size / granularity) + 1) * granularity
//   getStepsFromParameterization :282-315   the parameterisation (here: the PPC front end + the GPU step producer) hands
//                           its steps over in bunches of at most maxBunchSize; every bunch is inserted, then flushed
// One difference by construction: the reference converts a light source on the host, step by step; here the steps of a
// light source are born on the GPU in one launch (steps_kernel.hip) and enter the store in the same order and chunks.
#include "feeder.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" int clsimhip_generate_steps(int device, const clsimhip_step_request *requests, size_t n, uint64_t seed, size_t granularity,
                                       clsimhip_step *steps_out, size_t capacity, size_t *padded_out);
extern "C" int clsimhip_count_generated_steps(const clsimhip_step_request *requests, size_t n, size_t granularity, size_t *steps_out, size_t *padded_out);
extern "C" const char *clsimhip_last_error(const clsimhip_converter *c);

namespace clsimhip {

namespace {
void hip_must(hipError_t e, const char *what)
{
    if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
} // namespace

StepProducer::~StepProducer()
{
    if (!stream_ && !d_steps_ && !h_steps_) return;
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    (void)hipSetDevice(device_);
    if (stream_) { (void)hipStreamSynchronize(stream_); (void)hipStreamDestroy(stream_); }
    (void)hipFree(d_steps_); (void)hipFree(d_req_); (void)hipFree(d_first_);
    if (h_steps_) (void)hipHostFree(h_steps_);
    if (h_req_) (void)hipHostFree(h_req_);
    if (h_first_) (void)hipHostFree(h_first_);
    if (previous >= 0) (void)hipSetDevice(previous);
}

// stream and buffers (the caller has selected the device)
void StepProducer::ensure(size_t steps, size_t requests)
{
    if (!stream_) {
        int least = 0, greatest = 0;
        hip_must(hipDeviceGetStreamPriorityRange(&least, &greatest), "stream priorities");
        hip_must(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, greatest), "step producer stream");
    }
    if (steps > cap_steps_) {
        const size_t cap = steps + steps / 4;
        (void)hipFree(d_steps_); d_steps_ = nullptr;
        if (h_steps_) { (void)hipHostFree(h_steps_); h_steps_ = nullptr; }
        cap_steps_ = 0;
        hip_must(hipMalloc(&d_steps_, cap * sizeof(clsimhip_step)), "step buffer");
        hip_must(hipHostMalloc(reinterpret_cast<void **>(&h_steps_), cap * sizeof(clsimhip_step), hipHostMallocDefault), "pinned step buffer");
        cap_steps_ = cap;
    }
    if (requests > cap_req_) {
        const size_t cap = 2 * requests;
        (void)hipFree(d_req_); (void)hipFree(d_first_); d_req_ = d_first_ = nullptr;
        if (h_req_) { (void)hipHostFree(h_req_); h_req_ = nullptr; }
        if (h_first_) { (void)hipHostFree(h_first_); h_first_ = nullptr; }
        cap_req_ = 0;
        hip_must(hipMalloc(&d_req_, cap * sizeof(clsimhip_step_request)), "request buffer");
        hip_must(hipMalloc(&d_first_, cap * sizeof(uint64_t)), "offset buffer");
        hip_must(hipHostMalloc(&h_req_, cap * sizeof(clsimhip_step_request), hipHostMallocDefault), "pinned request buffer");
        hip_must(hipHostMalloc(&h_first_, cap * sizeof(uint64_t), hipHostMallocDefault), "pinned offset buffer");
        cap_req_ = cap;
    }
}

void StepProducer::reserve(size_t steps, size_t requests) noexcept
{
    try {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device_ < 0 || device_ >= count) { (void)hipGetLastError(); return; }
        DeviceGuard on_device(device_);
        ensure(steps, requests);
    } catch (...) {
        (void)hipGetLastError();
    }
}

const clsimhip_step *StepProducer::generate(const std::vector<clsimhip_step_request> &requests, uint64_t seed, size_t granularity, size_t &real, size_t &padded)
{
    // the plan of clsimhip_generate_steps (c_api.cpp: plan_steps), with its checks
    const size_t n = requests.size();
    if (granularity == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "granularity must not be 0");
    std::vector<uint64_t> first(n + 1, 0);
    for (size_t i = 0; i < n; ++i) {
        const clsimhip_step_request &q = requests[i];
        if (q.kind > CLSIMHIP_STEPS_MUON) throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown step request kind");
        if (q.photons_per_step == 0 && q.num_steps > 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "photonsPerStep may not be <= 0!");
        if (q.kind == CLSIMHIP_STEPS_CASCADE && !(q.pa > 0.f)) throw Error(CLSIMHIP_ERR_ARGUMENT, "cascade shape parameter must be positive");
        first[i + 1] = first[i] + q.num_steps + (q.num_photons_in_last_step > 0 ? 1 : 0);
    }
    real = static_cast<size_t>(first[n]);
    padded = ((real + granularity - 1) / granularity) * granularity;
    if (padded == 0) return nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available (the step producer has no CPU fallback)");
    if (device_ < 0 || device_ >= count) throw Error(CLSIMHIP_ERR_DEVICE, "device ordinal out of range");
    DeviceGuard on_device(device_);
    ensure(padded, n + 1);
    if (n) std::memcpy(h_req_, requests.data(), n * sizeof(clsimhip_step_request));
    std::memcpy(h_first_, first.data(), (n + 1) * sizeof(uint64_t));
    if (n) hip_must(hipMemcpyAsync(d_req_, h_req_, n * sizeof(clsimhip_step_request), hipMemcpyHostToDevice, stream_), "upload requests");
    hip_must(hipMemcpyAsync(d_first_, h_first_, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream_), "upload offsets");
    hip_must(launch_generate_steps(static_cast<const clsimhip_step_request *>(d_req_), static_cast<const uint64_t *>(d_first_),
                                   static_cast<uint32_t>(n ? n : 1), real, padded, seed, d_steps_, stream_), "step generation kernel launch");
    hip_must(hipMemcpyAsync(h_steps_, d_steps_, padded * sizeof(clsimhip_step), hipMemcpyDeviceToHost, stream_), "download steps");
    hip_must(hipStreamSynchronize(stream_), "step generation");
    return h_steps_;
}

Feeder::Feeder(const PPCConverter *ppc, int device, uint64_t seed, size_t max_bunch_size, size_t granularity, size_t queue_depth)
    : ppc_(ppc), device_(device), seed_(seed), max_bunch_(max_bunch_size), granularity_(granularity)
{
    if (max_bunch_ == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "MaxBunchSize of 0 is invalid!");                   // :418-419
    if (granularity_ == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "BunchSizeGranularity of 0 is invalid!");         // :407-408
    if (max_bunch_ % granularity_ != 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "MaxBunchSize is not a multiple of BunchSizeGranularity!");   // :83-84
    in_.reset(new BoundedQueue<Item>(queue_depth ? queue_depth : 10));       // queueToGeant4_, queueFromGeant4_ (:64-65: depth 10 by default)
    out_.reset(new BoundedQueue<Result>(queue_depth ? queue_depth : 10));
    if (ppc_) {
        // the step producer's stream and buffers now (Initialize() of the Python / C++ classes), sized for a light source of three
        // bunches (a 40 TeV cascade is 2.4 bunches of a million steps); a larger one grows them when it comes
        producer_.reset(new StepProducer(device_));
        // (capped: at the 6.1M-stream limit three bunches would page-lock and device-allocate 1.1 GB each for a feeder that may never be
        // fed -- ADVICE r5; 2.5M steps cover a 40 TeV cascade at the usual bunch of a million)
        producer_->reserve(std::min<size_t>(3 * max_bunch_, 2500000), 4096);
    }
    thread_ = std::thread([this] { worker(); });
}

Feeder::~Feeder()
{
    in_->close();
    out_->close();
    if (thread_.joinable()) thread_.join();
}

std::string Feeder::worker_error() const
{
    std::lock_guard<std::mutex> lk(error_mutex_);
    return error_;
}

// A feeder whose worker thread has died (a device error in the step producer, say) accepts nothing more: the caller gets
// the worker's message instead of an item that would be dropped, or of a put() that waits on a queue nobody drains.
void Feeder::check_worker() const
{
    const std::string e = worker_error();
    if (!e.empty()) throw Error(CLSIMHIP_ERR_DEVICE, "feeder thread: " + e);
}

void Feeder::enqueue_light_source(const clsimhip_particle &particle)
{
    check_worker();
    if (!ppc_) throw Error(CLSIMHIP_ERR_STATE, "this feeder was created without a particle parameterisation");
    if (barrier_enqueued_) throw Error(CLSIMHIP_ERR_STATE, "A barrier is enqueued! You must receive all steps before enqueuing a new particle.");   // :476-477
    Item it;
    it.has_particle = true;
    it.particle = particle;
    it.identifier = particle.identifier;
    in_->put(std::move(it));
    check_worker();                                         // (the worker may have died while this call waited for room)
}

void Feeder::enqueue_steps(uint32_t identifier, const clsimhip_step *steps, size_t n)
{
    check_worker();
    if (barrier_enqueued_) throw Error(CLSIMHIP_ERR_STATE, "A barrier is enqueued! You must receive all steps before enqueuing a new particle.");
    Item it;
    it.identifier = identifier;
    it.steps.assign(steps, steps + n);
    in_->put(std::move(it));
    check_worker();
}

void Feeder::enqueue_barrier()
{
    check_worker();
    bool expected = false;
    if (!barrier_enqueued_.compare_exchange_strong(expected, true)) throw Error(CLSIMHIP_ERR_STATE, "A barrier is already enqueued!");   // :497-498
    Item it;
    it.barrier = true;
    in_->put(std::move(it));
    if (!worker_error().empty()) { barrier_enqueued_ = false; check_worker(); }
}

bool Feeder::get_result(double timeout_ms, Result &out)
{
    {
        const std::string e = worker_error();
        if (!e.empty()) throw Error(CLSIMHIP_ERR_DEVICE, "feeder thread: " + e);
    }
    bool got;
    if (timeout_ms < 0. || std::isnan(timeout_ms)) got = out_->get(out);
    else got = out_->get_for(out, static_cast<long>(timeout_ms * 1000.));
    if (!got) {
        const std::string e = worker_error();
        if (!e.empty()) throw Error(CLSIMHIP_ERR_DEVICE, "feeder thread: " + e);
        return false;
    }
    if (out.last_before_barrier) barrier_enqueued_ = false;       // :560-566: the barrier is reset by the reply that carries it
    return true;
}

// flushStepStore (:209-273)
void Feeder::flush(bool reset_barrier)
{
    while (store_.size() >= max_bunch_) {
        Result r;
        r.steps.reset(new std::vector<clsimhip_step>(max_bunch_));
        const size_t n = store_.pop_bunch(max_bunch_, r.steps->data());
        r.steps->resize(n);
        while (!markers_.empty() && store_.count(markers_.front()) == 0) {      // :217-221
            r.finished.push_back(markers_.front());
            markers_.pop_front();
        }
        out_->put(std::move(r));
    }
    if (!reset_barrier) return;
    clsimhip_step no_op{};                                  // NoOpStepTemplate (:246-254): direction (0, 0, -1), beta 1, nothing else
    no_op.theta = 3.14159265358979323846f;                  // I3CLSimStep::SetDir(I3Direction(0, 0, -1)): theta = pi, phi = 0
    no_op.beta = 1.f;
    Result r;
    const size_t padded = store_.size_with_dummy_fill(granularity_);
    r.steps.reset(new std::vector<clsimhip_step>(padded));
    store_.pop_bunch_filled(padded, r.steps->data(), no_op);
    if (!store_.empty()) throw Error(CLSIMHIP_ERR_STATE, "Internal logic error. step store should be empty.");
    r.finished.assign(markers_.begin(), markers_.end());   // :263-265
    markers_.clear();
    r.last_before_barrier = true;
    out_->put(std::move(r));
}

// :298-312: the parameterisation's bunches (at most maxBunchSize steps each) are inserted whole, then the store is flushed
void Feeder::insert_and_flush(const clsimhip_step *steps, size_t n)
{
    for (size_t lo = 0; lo < n; lo += max_bunch_) {
        const size_t hi = std::min(n, lo + max_bunch_);
        store_.insert_many(steps + lo, hi - lo);
        flush(false);
    }
}

void Feeder::worker()
{
    try {
        for (;;) {
            Item it;
            if (!in_->get(it)) break;                       // closed: shut down
            flush(it.barrier);                              // :355-361
            if (it.barrier) continue;
            if (it.has_particle) {
#ifdef CLSIMHIP_DEVELOPER
                static const bool trace = std::getenv("CLSIMHIP_FEEDER_TRACE") != nullptr;      // analysis (developer build): where a light source's time goes
#else
                constexpr bool trace = false;
#endif
                const auto t_a = std::chrono::steady_clock::now();
                std::vector<clsimhip_step_request> requests;
                ppc_->enqueue(it.particle, requests);
                const auto t_b = std::chrono::steady_clock::now();
                // one random stream set per light source: results do not depend on what else is in the queue
                // (and an identifier that comes back gets streams of its own: OccurrenceCounter, lightsource.h)
                const uint64_t seed = seed_ ^ (0x9E3779B97F4A7C15ull * (static_cast<uint64_t>(it.identifier) + 1ull)) ^ occurrences_.mix(it.identifier);
                if (!producer_) producer_.reset(new StepProducer(device_));
                size_t real = 0, padded = 0;
                const clsimhip_step *steps = producer_->generate(requests, seed, 1, real, padded);
                const auto t_c = std::chrono::steady_clock::now();
                if (real) insert_and_flush(steps, real);
                if (trace) {
                    const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                    std::fprintf(stderr, "feeder: light source %u: %zu requests in %.1f ms, %zu steps born and downloaded in %.1f ms, store + bunches %.1f ms\n",
                                 it.identifier, requests.size(), ms(t_a, t_b), real, ms(t_b, t_c), ms(t_c, std::chrono::steady_clock::now()));
                }
            } else {
                insert_and_flush(it.steps.data(), it.steps.size());
            }
            markers_.push_back(it.identifier);              // :388: eligible for finalisation after the next bunch
        }
    } catch (const std::exception &e) {
        {
            std::lock_guard<std::mutex> lk(error_mutex_);
            error_ = e.what();
            if (error_.empty()) error_ = "unknown error";
        }
        // both ends: a consumer waiting for steps and a producer waiting for room wake up and find the error
        out_->close();
        in_->close();
    }
}

} // namespace clsimhip
