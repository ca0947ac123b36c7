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

#include <cmath>
#include <cstring>

extern "C" int clsimhip_generate_steps(int device, const clsimhip_step_request *requests, size_t n, uint64_t seed, size_t granularity,
                                       clsimhip_step *steps_out, size_t capacity, size_t *padded_out);
extern "C" int clsimhip_count_generated_steps(const clsimhip_step_request *requests, size_t n, size_t granularity, size_t *steps_out, size_t *padded_out);
extern "C" const char *clsimhip_last_error(const clsimhip_converter *c);

namespace clsimhip {

Feeder::Feeder(const PPCConverter *ppc, int device, uint64_t seed, size_t max_bunch_size, size_t granularity, size_t queue_depth)
    : ppc_(ppc), device_(device), seed_(seed), max_bunch_(max_bunch_size), granularity_(granularity)
{
    if (max_bunch_ == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "MaxBunchSize of 0 is invalid!");                   // :418-419
    if (granularity_ == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "BunchSizeGranularity of 0 is invalid!");         // :407-408
    if (max_bunch_ % granularity_ != 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "MaxBunchSize is not a multiple of BunchSizeGranularity!");   // :83-84
    in_.reset(new BoundedQueue<Item>(queue_depth ? queue_depth : 10));       // queueToGeant4_, queueFromGeant4_ (:64-65: depth 10 by default)
    out_.reset(new BoundedQueue<Result>(queue_depth ? queue_depth : 10));
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
        for (size_t i = lo; i < hi; ++i) store_.insert(steps[i]);
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
                std::vector<clsimhip_step_request> requests;
                ppc_->enqueue(it.particle, requests);
                size_t real = 0, padded = 0;
                if (clsimhip_count_generated_steps(requests.data(), requests.size(), 1, &real, &padded) != CLSIMHIP_OK)
                    throw Error(CLSIMHIP_ERR_STATE, clsimhip_last_error(nullptr));
                std::vector<clsimhip_step> steps(padded);
                if (padded) {
                    // one random stream set per light source: results do not depend on what else is in the queue
                    // (and an identifier that comes back gets streams of its own: OccurrenceCounter, lightsource.h)
                    const uint64_t seed = seed_ ^ (0x9E3779B97F4A7C15ull * (static_cast<uint64_t>(it.identifier) + 1ull)) ^ occurrences_.mix(it.identifier);
                    if (clsimhip_generate_steps(device_, requests.data(), requests.size(), seed, 1, steps.data(), steps.size(), &padded) != CLSIMHIP_OK)
                        throw Error(CLSIMHIP_ERR_DEVICE, clsimhip_last_error(nullptr));
                }
                insert_and_flush(steps.data(), real);
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
