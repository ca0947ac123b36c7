// Host runtime of the MI355X step->photon converter: configuration, table
// compilation, RNG set-up, the worker thread that owns the device, statistics.
// MI355X-native counterpart of I3CLSimStepToPhotonConverterOpenCL
// (private/opencl/I3CLSimStepToPhotonConverterOpenCL.cxx).
#pragma once
#include <chrono>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "host_model.h"
#include "kparams.h"

namespace clsimhip {

hipError_t launch_prop_kernel(const KParams &P, const KVariant &v, hipStream_t stream);
// pooled scheduling (prop_pool_kernel.hip): same results, propagation without photon histories only
hipError_t launch_pool_kernel(const KParams &P, const KVariant &v, hipStream_t stream);
hipError_t launch_pool_keep_kernel(const KParams &P, const KVariant &v, hipStream_t stream);     // prop_pool_keep_kernel.hip: without STOP_PHOTONS_ON_DETECTION
bool pool_kernel_fits(uint32_t table_words, uint32_t keep_strings, int num_layers);
size_t pool_kernel_max_steps();     // bunches beyond this many steps do not fit the pooled kernel's pending entries (23-bit step index)
hipError_t launch_eval_math(int what, const float *xs, const float *ys, uint32_t n, float *out, hipStream_t stream);
hipError_t launch_check_math(int what, int exp_lo, int exp_hi, uint32_t *result, uint32_t result_cap, hipStream_t stream);
size_t prop_kernel_lds_bytes(uint32_t table_words);
int prop_kernel_block_size();
hipError_t launch_generate_flasher_steps(const clsimhip_flasher_config &cfg, const clsimhip_flasher_request *d_requests, const void *d_plan,
                                         uint32_t n_requests, uint64_t total, uint64_t seed, const float *d_profiles, void *d_out,
                                         hipStream_t stream);
hipError_t launch_generate_steps(const clsimhip_step_request *d_requests, const uint64_t *d_first_step, uint32_t n_requests,
                                 uint64_t total_real, uint64_t total_padded, uint64_t seed, void *d_out, hipStream_t stream);
hipError_t launch_tab_kernel(const KParams &P, const KVariant &v, hipStream_t stream);
hipError_t launch_eval_function(const KParams &P, int lengths_kind, bool has_tilt, bool fast, int what, int layer, const float4 *in, uint32_t n, float4 *out,
                                hipStream_t stream);
hipError_t launch_eval_random(const KParams &P, bool fast, int what, int generator, uint64_t *x, const uint32_t *a, uint32_t n_streams, uint32_t draws,
                              float *out, hipStream_t stream);
hipError_t launch_keep_kernel(const KParams &P, const KVariant &v, hipStream_t stream);     // prop_keep_kernel.hip: without STOP_PHOTONS_ON_DETECTION
size_t prop_kernel_max_lanes();
size_t prop_kernel_lds_budget();

// RCCL gather of detected photons (comm.cpp)
struct Comm;
void comm_unique_id(uint8_t id[CLSIMHIP_UNIQUE_ID_BYTES]);
Comm *comm_create(int device, int rank, int world, const uint8_t id[CLSIMHIP_UNIQUE_ID_BYTES]);
void comm_destroy(Comm *c);
void comm_info(Comm *c, int *ranks, int *rank, int *device, char *pci_bus_id, size_t pci_bytes);
void comm_statistics(Comm *c, uint64_t *gathers, double *gather_ms, uint64_t *records_sent, uint64_t *records_received, bool reset);
void comm_gather_hits(Comm *c, const void *d_photons, const void *d_hit_count, size_t capacity, int root, void *d_gathered,
                      size_t gathered_capacity, uint64_t *counts_out, hipStream_t stream);

// Result of Compile(): kernel parameters without buffer pointers, the LDS image,
// the DOM templates and the named tables the parity tests read back.
struct CompiledTables {
    KParams params{};
    KVariant variant{};
    std::vector<uint32_t> lds_image;
    std::vector<float> len_table;       // KParams::len_table (TABLE lengths)
    std::vector<uint32_t> prox_map;     // KParams::prox_map
    std::vector<uint32_t> dom_prox;     // KParams::dom_prox
    std::vector<float> dom_centres;     // KParams::dom_centres (4 floats per DOM)
    std::vector<uint32_t> dom_named;    // KParams::dom_named (4 words per DOM)
    GeoTables geo;
    std::map<std::string, std::vector<double>> named;
};
// what Compile() may be told about the search filter's tables ("string_map_cells", "dom_map_cells", "named_search" of
// clsimhip_set_tuning; measurement and the filter-off parity points of tests/): no result depends on it
struct TableTuning {
    int string_map_cells = 512;     // string proximity map: cells per axis (8 ... 4096)
    int dom_map_cells = 256;        // DOM proximity map: at most this many cubic cells per axis (4 ... 512)
    bool named_search = true;       // false: every DOM is marked "not nameable" and takes the full search
};
CompiledTables compile_tables(const MediumData &medium, const GeometryInput &geometry,
                              const std::vector<RandomValueData> &generators, const FunctionData &bias,
                              double pancake_factor, const TableTuning &tuning = TableTuning());

// bounded blocking queue (I3CLSimQueue.h:48-195).  Like the reference's, a consumer that waits in get() counts as
// one free place: with capacity 0 the queue is a rendezvous -- put() returns only once a get() is there to take the
// item (queueFromOpenCL_, OpenCL.cxx:77-78), so size() / empty() show an item only while it is being handed over.
template <class T>
class BoundedQueue {
public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    void put(T v)
    {
        std::unique_lock<std::mutex> lk(m_);
        cond_.wait(lk, [&] { return q_.size() < cap_ + waiting_ || closed_; });
        if (closed_) return;
        q_.push_back(std::move(v));
        cond_.notify_all();
    }
    bool get(T &out)
    {
        std::unique_lock<std::mutex> lk(m_);
        ++waiting_;                     // I3CLSimQueue.h:85-87: tell the producer that somebody is waiting
        cond_.notify_all();
        cond_.wait(lk, [&] { return !q_.empty() || closed_; });
        --waiting_;
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        cond_.notify_all();
        return true;
    }
    // waits at most `us` microseconds; false if nothing arrived (or the queue was closed and is empty)
    bool get_for(T &out, long us)
    {
        std::unique_lock<std::mutex> lk(m_);
        ++waiting_;
        cond_.notify_all();
        const bool got = cond_.wait_for(lk, std::chrono::microseconds(us), [&] { return !q_.empty() || closed_; });
        --waiting_;
        if (!got || q_.empty()) { cond_.notify_all(); return false; }
        out = std::move(q_.front());
        q_.pop_front();
        cond_.notify_all();
        return true;
    }
    bool closed() const { std::lock_guard<std::mutex> lk(m_); return closed_; }
    size_t size() const { std::lock_guard<std::mutex> lk(m_); return q_.size(); }
    bool empty() const { return size() == 0; }
    void close() { std::lock_guard<std::mutex> lk(m_); closed_ = true; cond_.notify_all(); }
private:
    size_t cap_;
    mutable std::mutex m_;
    std::condition_variable cond_;
    std::deque<T> q_;
    size_t waiting_ = 0;
    bool closed_ = false;
};

// hipSetDevice for the duration of a call made on the CALLER's thread; the caller's current device is restored
struct DeviceGuard {
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&previous_) != hipSuccess) previous_ = -1;
        const hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    }
    ~DeviceGuard() { if (previous_ >= 0) (void)hipSetDevice(previous_); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
private:
    int previous_ = -1;
};

class Converter {
public:
    explicit Converter(int device);
    ~Converter();

    // configuration; each throws Error(STATE) once initialized (OpenCL.cxx:1322-1523)
    void set_wlen_generators(std::vector<RandomValueData> g);
    void set_wlen_bias(FunctionData b);
    void set_medium(MediumData m);
    void set_geometry(GeometryInput g);
    void set_double_buffering(bool v) { guard(); double_buffering_ = v; }
    void set_double_precision(bool v) { guard(); double_precision_ = v; }
    void set_stop_detected(bool v) { guard(); stop_detected_ = v; }
    void set_save_all(bool v) { guard(); save_all_ = v; }
    void set_save_all_prescale(double v) { guard(); save_all_prescale_ = v; }
    void set_fixed_abs_lengths(double v) { guard(); fixed_abs_lengths_ = v; }
    void set_pancake(double v) { guard(); pancake_ = v; }
    void set_history_entries(uint32_t v) { guard(); history_entries_ = v; }
    void set_workgroup_size(size_t v);
    void set_max_num_workitems(size_t v);

    void compile();
    void initialize(uint64_t seed);
    void initialize_with_streams(const uint64_t *x, const uint32_t *a, size_t count);
    bool initialized() const { return initialized_; }
    size_t max_workgroup_size() const { return static_cast<size_t>(prop_kernel_block_size()); }
    size_t workgroup_size() const { return workgroup_size_; }
    size_t max_num_workitems() const { return max_workitems_; }

    void enqueue_steps(const clsimhip_step *steps, size_t n, uint32_t identifier);
    void get_result(uint32_t *identifier, const clsimhip_photon **photons, size_t *n);
    void result_histories(const clsimhip_photon *photons, const float **histories, uint32_t *entries);
    void release_result(const clsimhip_photon *photons);
    size_t queue_size() const;
    bool more_photons_available() const;
    void statistics(double out[8]) const;
    double option(int which) const;

    void propagate_device(const void *d_steps, size_t n, size_t rng_offset, void *d_photons, size_t capacity,
                          void *d_hit_count, hipStream_t stream);
    void replace_indices(clsimhip_photon *photons, size_t n) const;
    void kernel_time(bool reset, double *total_ms, uint64_t *launches);
    long get_table(const std::string &name, double *out, size_t cap) const;
    void get_rng_state(uint64_t *x, size_t count);
    // the reference's tester classes (prop_eval_kernel.hip)
    void eval_device_function(int what, int layer, bool fast, const float *in4, size_t n, float *out4);
    void eval_device_random(int what, int generator, bool fast, uint64_t *x, const uint32_t *a, size_t n_streams, size_t draws, float *out);
    void debug_counters(uint32_t out[4]);
    // SetDevice (OpenCL.cxx:1322-1331): the HIP device ordinal, before Initialize()
    void set_device(int device);
    int device() const { return device_; }
    // the pooled kernel pays from ~3 steps per unit slot on: a launch that owns 1/k of the chip reaches that with 1/k of the steps
    bool pooled_for(size_t n_steps) const
    {
        need_init();
        std::lock_guard<std::mutex> lk(tuning_mutex_);
        return use_pool_ && n_steps <= pool_max_steps_ && n_steps * static_cast<size_t>(concurrent_launches_) >= pool_min_steps_;
    }
    // clsimhip_set_tuning / clsimhip_get_tuning (include/clsimhip.h has the table of keys): launcher parameters, any time between
    // launches; the three table keys before Compile().  The DEFAULT build reads no tuning from the environment.
    void set_tuning(const std::string &key, long long value);
    long long get_tuning(const std::string &key) const;
    void set_concurrent_device_launches(int k);
    int concurrent_device_launches() const { return concurrent_launches_; }
    bool uses_pooled_kernel() const { need_init(); std::lock_guard<std::mutex> lk(tuning_mutex_); return use_pool_ && pool_min_steps_ == 0; }     // for every bunch size

private:
    // A bunch on its way to the worker: the caller's steps are copied ONCE, in the caller's thread, into a page-locked buffer of the
    // step pool, which the worker uploads from (round 5; before, a vector here and a second copy into the slot's staging buffer on
    // the worker thread -- 8 ms per million steps in front of the first kernel of a run).  With every pool buffer in flight the
    // steps travel in a vector as before.
    struct StepBuffer { clsimhip_step *p = nullptr; size_t capacity = 0; };
    // (a pool buffer on loan: whoever drops it -- a job the closed queue refused, a queue destroyed with jobs inside -- returns it)
    struct StepLease {
        Converter *owner = nullptr;
        StepBuffer b;
        StepLease() = default;
        StepLease(Converter *o, StepBuffer buf) : owner(o), b(buf) {}
        StepLease(StepLease &&o) noexcept : owner(o.owner), b(o.b) { o.b = StepBuffer(); }
        StepLease &operator=(StepLease &&o) noexcept { if (this != &o) { drop(); owner = o.owner; b = o.b; o.b = StepBuffer(); } return *this; }
        StepLease(const StepLease &) = delete;
        StepLease &operator=(const StepLease &) = delete;
        ~StepLease() { drop(); }
        StepBuffer release() { const StepBuffer r = b; b = StepBuffer(); return r; }
        void drop() { if (owner && b.p) owner->give_step_buffer(b); b = StepBuffer(); }
    };
    struct Job { uint32_t id = 0; size_t n = 0; uint64_t generated = 0; StepLease pinned; std::vector<clsimhip_step> steps; };
    // The photons of a result live in a page-locked buffer of the converter's pool (the download lands there and the
    // caller reads them there until ReleaseResult: no host copy in between), or -- when the pool is exhausted because the
    // caller holds more results than it has buffers -- in a vector of their own.
    struct Result {
        uint32_t id = 0;
        struct HostFree { void operator()(clsimhip_photon *p) const; };
        std::unique_ptr<clsimhip_photon, HostFree> pinned;  // pool buffer (returned to the pool by release_result; freed with a result nobody fetched)
        size_t pinned_capacity = 0;                         // records `pinned` holds
        size_t count = 0;
        std::unique_ptr<std::vector<clsimhip_photon>> photons;
        std::unique_ptr<std::vector<float>> histories;     // [photons][history_entries_][4], forward order
        const clsimhip_photon *data() const { return pinned ? pinned.get() : (photons ? photons->data() : nullptr); }
    };

    void guard() const { if (initialized_) throw Error(CLSIMHIP_ERR_STATE, "I3CLSimStepToPhotonConverterHIP already initialized!"); }
    void need_init() const { if (!initialized_) throw Error(CLSIMHIP_ERR_STATE, "I3CLSimStepToPhotonConverterHIP is not initialized!"); }
    void setup_device_buffers();
    void release_device();      // frees every device / pinned allocation, stream and event (idempotent)
    void worker();
    void check_worker() const;  // throws the worker thread's device error, if it has died of one
    void hip_check(hipError_t e, const char *what) const;
    KParams launch_params(const void *d_steps, size_t n, size_t rng_offset, void *d_photons, size_t capacity, void *d_hits, hipStream_t stream);

    int device_;
    std::vector<RandomValueData> generators_;
    FunctionData bias_;
    MediumData medium_;
    GeometryInput geometry_;
    bool have_bias_ = false, have_medium_ = false, have_geometry_ = false;
    bool double_buffering_ = false, double_precision_ = false, stop_detected_ = false, save_all_ = false;     // (OpenCL.cxx:83-87: the class's own defaults;
                                                                                                          // initializeOpenCL's callers pass stopDetectedPhotons = true)
    double save_all_prescale_ = 0.001, fixed_abs_lengths_ = NAN, pancake_ = 1.0;
    uint32_t history_entries_ = 0;
    size_t workgroup_size_ = 0, max_workitems_ = 0;
    uint32_t max_output_photons_ = 0;

    bool compiled_ = false, initialized_ = false;
    CompiledTables tables_;

    // device state
    uint32_t *d_tables_ = nullptr;
    int16_t *d_dom_tx_ = nullptr, *d_dom_ty_ = nullptr;
    float *d_len_table_ = nullptr;
    uint32_t *d_prox_map_ = nullptr;
    uint32_t *d_dom_prox_ = nullptr;
    uint32_t *d_dom_named_ = nullptr;
    float *d_dom_centres_ = nullptr;
    float *d_hist_ring_ = nullptr;           // per resident lane: the last history_entries_ scatter points
    float *d_dom_tz_ = nullptr;
    uint64_t *d_rng_x_ = nullptr;
    uint32_t *d_rng_a_ = nullptr;
    // One buffer set per bunch in flight (OpenCL.cxx:296-340 allocates numBuffers = 1 or 2 of each): while
    // the kernel of bunch k+1 runs, the photons of bunch k are downloaded on the copy stream and converted.
    struct Slot {
        DevStep *d_steps = nullptr;
        DevPhoton *d_photons = nullptr;
        uint32_t *d_hit_count = nullptr;
        float *d_hist_out = nullptr;            // photon histories of the slot's hits (history_entries_ float4 each)
        float *h_hist = nullptr;
        clsimhip_step *h_steps = nullptr;       // pinned staging (for a job that came without a pool buffer)
        StepBuffer step_buffer;                 // the pool buffer the slot's upload reads; goes back to the pool when the slot is used again
        uint32_t *h_hit_count = nullptr;
        hipEvent_t start = nullptr, stop = nullptr, counted = nullptr, uploaded = nullptr;
        uint32_t id = 0;
        uint64_t generated = 0;
    };
    Slot slots_[2];
    int num_slots_ = 1;
    // Page-locked result buffers, sized by the photons that arrive: a bunch's download lands in a buffer of at least its own record
    // count (a quarter more, at least min_result_records_, never more than max_output_photons_), which then IS the result until the
    // caller releases it.  At most kResultBuffers exist at a time; a free one that is too small is given back to the host for a
    // larger one.  (Rounds 3-4 kept buffers of max_output_photons_ records each -- 840 MB at a million work items, six of them.)
    struct PinnedBuffer { clsimhip_photon *p = nullptr; size_t capacity = 0; };
    static constexpr int kResultBuffers = 6;
    size_t min_result_records_ = 65536;                 // ("result_min_records": tests make the buffers grow with small bunches)
    std::vector<PinnedBuffer> free_result_buffers_;
    int result_buffers_made_ = 0;
    bool pinning_refused_ = false;                      // the host would not page-lock more: results are copied out from then on
    std::mutex result_pool_mutex_;
    PinnedBuffer take_result_buffer(size_t records);    // {nullptr, 0} when every buffer is with the caller (or the host refuses)
    // Page-locked step buffers (see Job): input queue depth + one per slot + the one being filled, each sized by the bunch it first
    // carried (a quarter more) and replaced by a larger one when a later bunch needs it, at most kStepPoolBytes in all
    static constexpr int kStepBuffers = 8;
    static constexpr size_t kStepPoolBytes = size_t{1} << 30;
    std::vector<StepBuffer> free_step_buffers_;
    int step_buffers_made_ = 0;
    size_t step_pool_bytes_ = 0;
    bool step_pinning_refused_ = false;
    bool step_fallback_logged_ = false;                 // the pageable fallback is reported once (like finish() does for result buffers)
    std::mutex step_pool_mutex_;
    StepBuffer take_step_buffer(size_t steps);
    void give_step_buffer(StepBuffer b);
    // index -> ID tables on the device (host path: converted by assemble_hits_kernel); null when an ID does not fit the record
    int16_t *d_id_strings_ = nullptr;
    uint16_t *d_id_doms_ = nullptr;
    uint32_t *d_id_dom_start_ = nullptr;
    void submit(Slot &s, Job &job);
    void finish(Slot &s, std::chrono::steady_clock::time_point &last_done, bool &first);
    WorkRecord *d_work_ = nullptr;           // per step: work record (kparams.h), rebuilt by every launch
    uint32_t *d_queue_ = nullptr;            // ring of step-queue heads, one per launch in flight
    uint32_t queue_slot_ = 0;
    int concurrent_launches_ = 1;            // device path: launches the caller keeps in flight side by side (chip share of each)
    uint32_t *last_queue_ = nullptr;
#ifdef CLSIMHIP_CENSUS
    unsigned long long *d_census_ = nullptr;
#endif
    int k_wait_ = -1, k_aim_ = -1;               // -1 = automatic; "k_wait", "k_aim" (pooled kernel only: the classic and keep
                                                 // kernels have them as constants); 0 is honoured: never wait / never ask
    int k_search_ = 0;                           // lanes parked before a wave searches for DOMs, 0 = automatic ("k_search")
    int k_new_ = 0, k_slices_ = 0;               // creation threshold; slices per step, 0 = automatic ("k_new", "slices")
    int k_pop_ = 0, pool_ready_ = 0;             // pooled kernel: lanes serviced at once, ring entries per wave, 0 = automatic ("k_pop", "pool_ring")
    bool use_pool_ = false;                      // pooled kernel allowed (derived by apply_kernel_choice())
    size_t pool_min_steps_ = 0;                  // ... for bunches of at least this many steps
    size_t pool_max_steps_ = 0;                  // ... and at most this many (pool_kernel_max_steps(): the pending entries' 23-bit step index)
    int kernel_choice_ = 0;                      // "kernel": 0 automatic (by bunch size), 1 pooled, 2 classic for every bunch size
    long long tuned_pool_min_steps_ = -1;        // "pool_min_steps": -1 = the default threshold
    long long tuned_pool_max_steps_ = -1;        // "pool_max_steps": -1 = what the pending entries' step index can address; only ever lowers it
    int grid_ = 0;                               // "grid"
    bool generic_only_ = false;                  // "generic_kernels"
    TableTuning table_tuning_;
    bool pool_possible_ = false;                 // set by Initialize(): no photon histories, the table image leaves the pools their LDS
    void apply_kernel_choice();
    mutable std::mutex tuning_mutex_;            // the tuning members above: set_tuning() against the launches that read them
    hipError_t launch(const KParams &P, hipStream_t stream) const;
    hipStream_t stream_ = nullptr;           // upload + kernels (bunches serialise here: they share the RNG streams)
    hipStream_t copy_stream_ = nullptr;      // photon download
    hipStream_t upload_stream_ = nullptr;    // step upload of the next bunch while the previous kernel runs
    hipEvent_t ev_start_ = nullptr, ev_stop_ = nullptr;

    // worker + queues (in: capacity 5 like queueToOpenCL_, OpenCL.cxx:77)
    std::unique_ptr<BoundedQueue<Job>> in_queue_;
    std::unique_ptr<BoundedQueue<Result>> out_queue_;
    std::thread worker_;
    // a device error in the worker thread: kept for the callers (every later call returns CLSIMHIP_ERR_DEVICE with this text)
    mutable std::mutex fatal_mutex_;
    std::string fatal_error_;
    bool worker_failed_ = false;
    mutable std::mutex results_mutex_;
    std::map<const clsimhip_photon *, Result> handed_out_;

    // statistics (OpenCL.cxx:1088-1140, 1621-1640)
    mutable std::mutex stats_mutex_;
    uint64_t total_device_ns_ = 0, total_host_ns_ = 0, num_kernel_calls_ = 0, photons_generated_ = 0, photons_at_doms_ = 0;

    // device-resident path: event pairs on the caller's stream
    std::mutex ev_mutex_;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events_, free_events_;
    double dev_total_ms_ = 0;
    uint64_t dev_launches_ = 0;
};

} // namespace clsimhip
