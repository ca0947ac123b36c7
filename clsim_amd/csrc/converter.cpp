// Host runtime: see converter.h.  One worker thread owns the HIP streams
// (upload -> kernel on the compute stream, photon download on the copy stream,
// OpenCL.cxx:1142-1315); callers talk to it through two bounded queues, like the
// reference's queueToOpenCL_/queueFromOpenCL_.  With double buffering one bunch
// is on the GPU while the previous one is downloaded and converted.
#include "converter.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace clsimhip {
[[maybe_unused]] constexpr size_t kCensusBytes = 4u << 20;      // analysis build: counters + per-wave records + per-wave region counters


static_assert(sizeof(clsimhip_step) == sizeof(DevStep), "step layouts");
static_assert(sizeof(clsimhip_photon) == sizeof(DevPhoton), "photon layouts");
constexpr uint32_t kQueueSlots = 256;
// The pooled kernel keeps 64 + R units per wave in flight (about 0.6M on the chip): it wins on bunches that hold more
// steps than that by a margin (1M steps: +9 %, 4M: +8 %) and loses on smaller ones (0.5M: -17 %), where the classic
// kernel's smaller grids apply.  Chosen per launch; results do not depend on the choice.
// (round 4, with bunches just below the chip's unit slots sliced like larger ones -- prop_pool_kernel.hip -- and a ring of 45: cascade steps, classic /
// pooled 1e9 photons/s: 0.39M 2.57 / 2.56, 0.49M 2.89 / 2.90, 0.56M 2.98 / 3.18, 0.59M 3.00 / 3.30, 0.62M 3.02 / 3.42, 0.66M 3.03 / 3.50; flasher steps 0.31M 1.62 / 1.55,
// 0.63M 2.00 / 2.23.  Round 2's threshold was 614 400.)
constexpr size_t kPooledKernelMinSteps = 524288;

hipError_t Converter::launch(const KParams &P, hipStream_t stream) const
{
    KVariant v = tables_.variant;
    const bool pooled = pooled_for(P.n_steps);
    {
        std::lock_guard<std::mutex> lk(tuning_mutex_);
        v.grid = grid_;
        v.generic_only = generic_only_;
    }
    if (v.keep_detected) return pooled ? launch_pool_keep_kernel(P, v, stream) : launch_keep_kernel(P, v, stream);
    return pooled ? launch_pool_kernel(P, v, stream) : launch_prop_kernel(P, v, stream);
}

// ---- tuning (include/clsimhip.h: clsimhip_set_tuning) -----------------------------------------------------------------
void Converter::apply_kernel_choice()
{
    use_pool_ = pool_possible_ && kernel_choice_ != 2;
    pool_min_steps_ = (kernel_choice_ != 0) ? 0 : kPooledKernelMinSteps;
    if (tuned_pool_min_steps_ >= 0) pool_min_steps_ = static_cast<size_t>(tuned_pool_min_steps_);
    pool_max_steps_ = pool_kernel_max_steps();
    if (tuned_pool_max_steps_ >= 0) pool_max_steps_ = std::min(pool_max_steps_, static_cast<size_t>(tuned_pool_max_steps_));
}

namespace {
struct TuningKey { const char *name; long long lo, hi; };
// (the table of include/clsimhip.h, in its order)
const TuningKey kTuningKeys[] = {
    {"kernel", 0, 2}, {"pool_min_steps", -1, 1ll << 40}, {"pool_max_steps", -1, 1ll << 40}, {"pool_ring", 0, 4096},
    {"k_new", 0, 4096}, {"k_search", 0, 64}, {"slices", 0, 65535}, {"k_pop", 0, 64}, {"k_wait", -1, 255}, {"k_aim", -1, 64},
    {"grid", 0, 1 << 20}, {"generic_kernels", 0, 1}, {"result_min_records", 1, 1ll << 32},
    {"string_map_cells", 8, 4096}, {"dom_map_cells", 4, 512}, {"named_search", 0, 1},
};
}

void Converter::set_tuning(const std::string &key, long long value)
{
    const TuningKey *k = nullptr;
    for (const TuningKey &t : kTuningKeys) if (key == t.name) k = &t;
    if (!k) throw Error(CLSIMHIP_ERR_ARGUMENT, "no tuning key named " + key);
    if (value < k->lo || value > k->hi)
        throw Error(CLSIMHIP_ERR_ARGUMENT, "tuning " + key + ": " + std::to_string(value) + " is outside " + std::to_string(k->lo) + " ... " + std::to_string(k->hi));
    const bool table_key = (key == "string_map_cells" || key == "dom_map_cells" || key == "named_search");
    if (table_key && compiled_) throw Error(CLSIMHIP_ERR_STATE, "tuning " + key + " shapes a table of Compile(): set it before Compile()");
    std::lock_guard<std::mutex> lk(tuning_mutex_);          // (a launch reads its parameters under the same lock)
    const int v = static_cast<int>(value);
    if (key == "kernel") kernel_choice_ = v;
    else if (key == "pool_min_steps") tuned_pool_min_steps_ = value;
    else if (key == "pool_max_steps") tuned_pool_max_steps_ = value;
    else if (key == "pool_ring") pool_ready_ = v;
    else if (key == "k_new") k_new_ = v;
    else if (key == "k_search") k_search_ = v;
    else if (key == "slices") k_slices_ = v;
    else if (key == "k_pop") k_pop_ = v;
    else if (key == "k_wait") k_wait_ = v;
    else if (key == "k_aim") k_aim_ = v;
    else if (key == "grid") grid_ = v;
    else if (key == "generic_kernels") generic_only_ = (v != 0);
    else if (key == "result_min_records") min_result_records_ = static_cast<size_t>(value);
    else if (key == "string_map_cells") table_tuning_.string_map_cells = v;
    else if (key == "dom_map_cells") table_tuning_.dom_map_cells = v;
    else if (key == "named_search") table_tuning_.named_search = (v != 0);
    if (initialized_) apply_kernel_choice();
}

long long Converter::get_tuning(const std::string &key) const
{
    std::lock_guard<std::mutex> lk(tuning_mutex_);
    if (key == "kernel") return kernel_choice_;
    if (key == "pool_min_steps") return initialized_ ? static_cast<long long>(pool_min_steps_) : tuned_pool_min_steps_;
    if (key == "pool_max_steps") return initialized_ ? static_cast<long long>(pool_max_steps_) : tuned_pool_max_steps_;
    if (key == "pool_ring") return pool_ready_;
    if (key == "k_new") return k_new_;
    if (key == "k_search") return k_search_;
    if (key == "slices") return k_slices_;
    if (key == "k_pop") return k_pop_;
    if (key == "k_wait") return k_wait_;
    if (key == "k_aim") return k_aim_;
    if (key == "grid") return grid_;
    if (key == "generic_kernels") return generic_only_ ? 1 : 0;
    if (key == "result_min_records") return static_cast<long long>(min_result_records_);
    if (key == "string_map_cells") return table_tuning_.string_map_cells;
    if (key == "dom_map_cells") return table_tuning_.dom_map_cells;
    if (key == "named_search") return table_tuning_.named_search ? 1 : 0;
    throw Error(CLSIMHIP_ERR_ARGUMENT, "no tuning key named " + key);
}

void Converter::set_concurrent_device_launches(int k)
{
    if (k < 1 || k > 16) throw Error(CLSIMHIP_ERR_ARGUMENT, "concurrent device launches: 1 ... 16");
    concurrent_launches_ = k;
}

void Converter::hip_check(hipError_t e, const char *what) const
{
    if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

Converter::Converter(int device) : device_(device)
{
    // configuration and Compile() are host-only; the GPU is required from Initialize() on
    if (device < 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "device ordinal out of range");
}

Converter::~Converter()
{
    if (in_queue_) in_queue_->close();
    if (out_queue_) out_queue_->close();
    if (worker_.joinable()) worker_.join();
    in_queue_.reset();              // jobs nobody took give their page-locked step buffers back to the pool, which release_device() frees
    release_device();
}

void Converter::release_device()
{
    if (!stream_ && !d_tables_ && !d_rng_x_ && !slots_[0].d_steps) return;     // nothing was created
    int previous = -1;
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    (void)hipSetDevice(device_);
    for (auto &p : pending_events_) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : free_events_) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    pending_events_.clear(); free_events_.clear();
    if (ev_start_) (void)hipEventDestroy(ev_start_);
    if (ev_stop_) (void)hipEventDestroy(ev_stop_);
    if (stream_) (void)hipStreamDestroy(stream_);
    if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
    if (upload_stream_) (void)hipStreamDestroy(upload_stream_);
    ev_start_ = ev_stop_ = nullptr;
    stream_ = copy_stream_ = upload_stream_ = nullptr;
    for (Slot &sl : slots_) {
        if (sl.start) (void)hipEventDestroy(sl.start);
        if (sl.stop) (void)hipEventDestroy(sl.stop);
        if (sl.counted) (void)hipEventDestroy(sl.counted);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        (void)hipFree(sl.d_steps); (void)hipFree(sl.d_photons); (void)hipFree(sl.d_hit_count); (void)hipFree(sl.d_hist_out);
        if (sl.h_hist) (void)hipHostFree(sl.h_hist);
        if (sl.h_steps) (void)hipHostFree(sl.h_steps);
        if (sl.step_buffer.p) (void)hipHostFree(sl.step_buffer.p);
        if (sl.h_hit_count) (void)hipHostFree(sl.h_hit_count);
        sl = Slot();
    }
    for (const StepBuffer &b : free_step_buffers_) (void)hipHostFree(b.p);
    free_step_buffers_.clear(); step_buffers_made_ = 0; step_pool_bytes_ = 0; step_pinning_refused_ = false; step_fallback_logged_ = false;
    (void)hipFree(d_tables_); (void)hipFree(d_dom_tx_); (void)hipFree(d_dom_ty_); (void)hipFree(d_dom_tz_); (void)hipFree(d_len_table_); (void)hipFree(d_prox_map_); (void)hipFree(d_dom_prox_); (void)hipFree(d_dom_centres_); (void)hipFree(d_dom_named_); (void)hipFree(d_id_strings_); (void)hipFree(d_id_doms_); (void)hipFree(d_id_dom_start_);
    for (const PinnedBuffer &b : free_result_buffers_) (void)hipHostFree(b.p);
    free_result_buffers_.clear(); result_buffers_made_ = 0; pinning_refused_ = false;
    // results nobody released, and results nobody fetched: their page-locked buffers go with the converter
    handed_out_.clear();
    if (out_queue_) {
        Result r;
        while (out_queue_->get_for(r, 0)) {}
    }
    (void)hipFree(d_rng_x_); (void)hipFree(d_rng_a_);
    (void)hipFree(d_queue_); (void)hipFree(d_work_); (void)hipFree(d_hist_ring_);
    d_tables_ = nullptr; d_dom_tx_ = d_dom_ty_ = nullptr; d_dom_tz_ = nullptr; d_len_table_ = nullptr; d_prox_map_ = nullptr; d_dom_prox_ = nullptr; d_dom_centres_ = nullptr; d_dom_named_ = nullptr; d_id_strings_ = nullptr; d_id_doms_ = nullptr; d_id_dom_start_ = nullptr;
    d_rng_x_ = nullptr; d_rng_a_ = nullptr; d_queue_ = nullptr; d_work_ = nullptr; d_hist_ring_ = nullptr;
    last_queue_ = nullptr;
#ifdef CLSIMHIP_CENSUS
    (void)hipFree(d_census_); d_census_ = nullptr;
#endif
    if (previous >= 0) (void)hipSetDevice(previous);
}

void Converter::set_device(int device)
{
    guard();
    if (device < 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "device ordinal out of range");
    device_ = device;
}

void Converter::check_worker() const
{
    std::lock_guard<std::mutex> lk(fatal_mutex_);
    if (worker_failed_) throw Error(CLSIMHIP_ERR_DEVICE, "the converter's worker thread stopped after a device error: " + fatal_error_);
}

void Converter::set_wlen_generators(std::vector<RandomValueData> g)
{
    guard();
    // a wavelength that is zero, negative or not finite gives a photon lengths of 0/0 = NaN, and a NaN absorption
    // budget never runs out (propagation_kernel.c.cl:536): refused here rather than spun on by the GPU
    for (const RandomValueData &r : g) {
        if (r.kind == CLSIMHIP_RANDOM_CONSTANT) {
            if (!std::isfinite(r.value) || !(r.value > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "a constant wavelength must be finite and positive");
        } else if (r.kind == CLSIMHIP_RANDOM_CHERENKOV_NO_DISPERSION) {
            // WlenCherenkovNoDispersion.cxx:47-51
            if (std::isnan(r.first)) throw Error(CLSIMHIP_ERR_ARGUMENT, "The \"fromWlen\" argument must not be NaN!");
            if (std::isnan(r.spacing)) throw Error(CLSIMHIP_ERR_ARGUMENT, "The \"toWlen\" argument must not be NaN!");
            if (r.first > r.spacing) throw Error(CLSIMHIP_ERR_ARGUMENT, "The \"fromWlen\" argument must not be greater than \"toWlen\".");
            if (!(r.first > 0.) || !std::isfinite(r.spacing)) throw Error(CLSIMHIP_ERR_ARGUMENT, "the wavelength range must be positive and finite");
        } else {
            if (r.kind == CLSIMHIP_RANDOM_INTERPOLATED_X) {
                if (r.x.size() != r.y.size()) throw Error(CLSIMHIP_ERR_ARGUMENT, "The \"x\" and \"y\" vectors must have the same size!");
                for (size_t i = 0; i < r.x.size(); ++i)
                    if (!std::isfinite(r.x[i]) || !(r.x[i] > 0.) || (i > 0 && !(r.x[i] > r.x[i - 1])))
                        throw Error(CLSIMHIP_ERR_ARGUMENT, "the wavelengths of a distribution must be positive, finite and ascending");
            } else if (!std::isfinite(r.first) || !(r.first > 0.) || !std::isfinite(r.spacing) || !(r.spacing > 0.))
                throw Error(CLSIMHIP_ERR_ARGUMENT, "a wavelength distribution needs a positive first wavelength and spacing");
            double sum = 0.;
            for (double y : r.y) {
                if (!std::isfinite(y) || y < 0.) throw Error(CLSIMHIP_ERR_ARGUMENT, "wavelength distribution values must be finite and non-negative");
                sum += y;
            }
            if (!(sum > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "a wavelength distribution must not be zero everywhere");
        }
    }
    compiled_ = false;
    generators_ = std::move(g);
}
void Converter::set_wlen_bias(FunctionData b)
{
    guard();
    if (!b.on_device()) throw Error(CLSIMHIP_ERR_ARGUMENT, "the wavelength bias must be a table with equal spacing or a constant (FromTable.cxx:169-170)");
    compiled_ = false; bias_ = std::move(b); have_bias_ = true;
}
void Converter::set_medium(MediumData m) { guard(); m.validate(); compiled_ = false; medium_ = std::move(m); have_medium_ = true; }
void Converter::set_geometry(GeometryInput g) { guard(); compiled_ = false; geometry_ = std::move(g); have_geometry_ = true; }

void Converter::set_workgroup_size(size_t v)
{
    guard();
    if (v == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "workgroup size must not be 0");
    if (v > max_workgroup_size()) throw Error(CLSIMHIP_ERR_ARGUMENT, "Workgroup size too large!");
    workgroup_size_ = v;
}
void Converter::set_max_num_workitems(size_t v)
{
    guard();
    if (v == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "maximum number of work items must not be 0");
    max_workitems_ = v;
}

// OpenCL.cxx:485-533
void Converter::compile()
{
    guard();
    if (compiled_) return;
    if (generators_.empty()) throw Error(CLSIMHIP_ERR_CONFIG, "WlenGenerators not set!");
    if (!have_bias_) throw Error(CLSIMHIP_ERR_CONFIG, "WlenBias not set!");
    if (!have_medium_) throw Error(CLSIMHIP_ERR_CONFIG, "MediumProperties not set!");
    if (!have_geometry_) throw Error(CLSIMHIP_ERR_CONFIG, "Geometry not set!");
    if (double_precision_) throw Error(CLSIMHIP_ERR_CONFIG, "DoublePrecision is not available in the HIP propagator");
    // SAVE_ALL_PHOTONS is unusable in the reference at this revision: it leaves out the geometry source (OpenCL.cxx:461-466)
    // that saveHit() needs for geometryGetDomPosition (propagation_kernel.c.cl:339).
    // StopDetectedPhotons=false (no STOP_PHOTONS_ON_DETECTION) runs the classic kernel's instantiations of their own
    // (prop_keep_kernel.hip).  The reference's collision code indexes dom_bitmask[stringNum/64] in an array of
    // (GEO_MAX_DOM_INDEX+63)/64 words there -- out of bounds from the 65th string on when no string has more than 64 DOMs
    // (sparse_collision_kernel.c.cl:85-104), undefined; here (and in the oracle) that word exists, which is also what the
    // kernel text compiled for x86-64 does (tools/verbatim_cl_check.py clear_keep).
    if (save_all_) throw Error(CLSIMHIP_ERR_CONFIG, "SaveAllPhotons is not available in the HIP propagator");
    if (history_entries_ > 1024) throw Error(CLSIMHIP_ERR_CONFIG, "PhotonHistoryEntries > 1024 is not supported");
#ifdef CLSIMHIP_DEVELOPER
    // (developer build only: the round 1-5 environment names of the three table keys)
    if (const char *e = std::getenv("CLSIMHIP_PROX_N")) set_tuning("string_map_cells", std::max(8, std::min(4096, std::atoi(e))));
    if (const char *e = std::getenv("CLSIMHIP_DOM_PROX_N")) set_tuning("dom_map_cells", std::max(4, std::min(512, std::atoi(e))));
    if (const char *e = std::getenv("CLSIMHIP_NO_NAMED_SEARCH")) set_tuning("named_search", e[0] == '1' ? 0 : 1);
#endif
    tables_ = compile_tables(medium_, geometry_, generators_, bias_, pancake_, table_tuning_);
    if (!std::isnan(fixed_abs_lengths_)) {                      // OpenCL.cxx:425-431
        tables_.params.has_fixed_abs = 1;
        tables_.params.fixed_abs = to_float_literal(fixed_abs_lengths_);
    }
    tables_.params.history_n = static_cast<int32_t>(history_entries_);   // OpenCL.cxx:416-419
    tables_.variant.keep_detected = !stop_detected_;                     // OpenCL.cxx:395-397
    compiled_ = true;
}

bool load_multipliers_from_file(const char *path, uint32_t *a, size_t count)
{
    // mwcrng_init.h:62-103: binary format = 17-byte tag "safeprimes_base32" + int64 LE each; else text, first column
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) return false;
    char tag[18] = {0};
    f.read(tag, 17);
    if (std::strcmp(tag, "safeprimes_base32") == 0) {
        for (size_t i = 0; i < count; ++i) {
            int64_t m = 0;
            f.read(reinterpret_cast<char *>(&m), sizeof m);
            if (f.fail() || m < 0 || m > 0xffffffffll) return false;
            a[i] = static_cast<uint32_t>(m);
        }
        return true;
    }
    f.close();
    std::ifstream t(path);
    std::string line;
    for (size_t i = 0; i < count; ++i) {
        if (!std::getline(t, line)) return false;
        char *end = nullptr;
        const long long m = std::strtoll(line.c_str(), &end, 10);
        if (end == line.c_str() || m < 0 || m > 0xffffffffll) return false;
        a[i] = static_cast<uint32_t>(m);
    }
    return true;
}

void Converter::initialize(uint64_t seed)
{
    guard();
    compile();
    if (max_workitems_ == 0) max_workitems_ = 1048576;
    std::vector<uint32_t> a(max_workitems_);
    std::vector<uint64_t> x(max_workitems_);
    const char *file = std::getenv("CLSIMHIP_SAFEPRIMES_FILE");
    if (!(file && load_multipliers_from_file(file, a.data(), a.size()))) mwc_multipliers(a.data(), a.size());
    seed_streams(a.data(), a.size(), seed, x.data());
    initialize_with_streams(x.data(), a.data(), a.size());
}

// OpenCL.cxx:217-388
void Converter::initialize_with_streams(const uint64_t *x, const uint32_t *a, size_t count)
{
    guard();
    if (!x || !a || count == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "RNG streams are (null) or empty");
    compile();
    if (workgroup_size_ == 0) workgroup_size_ = max_workgroup_size();
    if (max_workitems_ == 0) max_workitems_ = count;
    if (count != max_workitems_) throw Error(CLSIMHIP_ERR_ARGUMENT, "number of RNG streams must equal the maximum number of work items");
    if (max_workitems_ % workgroup_size_ != 0)
        throw Error(CLSIMHIP_ERR_ARGUMENT, "The maximum number of work items (" + std::to_string(max_workitems_) +
                                               ") must be a multiple of the workgroup size (" + std::to_string(workgroup_size_) + ").");
    if (max_workitems_ > 0x7fffffffull) throw Error(CLSIMHIP_ERR_ARGUMENT, "too many work items");
    for (size_t i = 0; i < count; ++i) {
        // mwcrng_init.h:107: a state outside this range breaks the generator
        if ((x[i] == 0) | ((static_cast<uint32_t>(x[i] >> 32)) >= (a[i] - 1)) | ((static_cast<uint32_t>(x[i])) >= 0xfffffffful))
            throw Error(CLSIMHIP_ERR_ARGUMENT, "invalid MWC state word for stream " + std::to_string(i));
    }
    // OpenCL.cxx:266-277
    max_output_photons_ = static_cast<uint32_t>(std::min<size_t>(max_workitems_ * 10, 0xffffffffull));
    if (max_output_photons_ < 1000) max_output_photons_ = 1000;

    {
        int count = 0;
        const hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0)
            throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available (the propagator has no CPU fallback)");
        if (device_ >= count) throw Error(CLSIMHIP_ERR_ARGUMENT, "device ordinal out of range");
    }
    // everything below runs with this converter's device current; the caller's current device is restored on return,
    // and a failure half-way releases what was created (a retry starts from nothing)
    DeviceGuard on_device(device_);
    try {
        hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipStreamCreateWithFlags(&upload_stream_, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipEventCreate(&ev_start_), "hipEventCreate");
        hip_check(hipEventCreate(&ev_stop_), "hipEventCreate");
        setup_device_buffers();
        hip_check(hipMemcpy(d_rng_x_, x, count * sizeof(uint64_t), hipMemcpyHostToDevice), "upload rng x");
        hip_check(hipMemcpy(d_rng_a_, a, count * sizeof(uint32_t), hipMemcpyHostToDevice), "upload rng a");
    } catch (...) {
        release_device();
        throw;
    }

    in_queue_.reset(new BoundedQueue<Job>(5));          // queueToOpenCL_(5), OpenCL.cxx:77
    out_queue_.reset(new BoundedQueue<Result>(0));      // queueFromOpenCL_(0): rendezvous, OpenCL.cxx:78
    initialized_ = true;
    worker_ = std::thread([this] { worker(); });
}

void Converter::setup_device_buffers()
{
    const GeoTables &G = tables_.geo;
    auto upload = [&](void **dst, const void *src, size_t bytes, const char *what) {
        hip_check(hipMalloc(dst, std::max<size_t>(bytes, 16)), what);
        if (bytes) hip_check(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice), what);
    };
    upload(reinterpret_cast<void **>(&d_tables_), tables_.lds_image.data(), tables_.lds_image.size() * 4, "tables");
    upload(reinterpret_cast<void **>(&d_len_table_), tables_.len_table.data(), tables_.len_table.size() * 4, "length tables");
    upload(reinterpret_cast<void **>(&d_prox_map_), tables_.prox_map.data(), tables_.prox_map.size() * sizeof(uint32_t), "string proximity map");
    {   // the DOM proximity map as the kernels read it: every cell with the centre of the DOM it names (kparams.h: dom_cells).  Built and
        // uploaded in pieces of a million cells: the whole image is up to 256 MB, which the host need not hold beside the words
        const size_t cells = tables_.dom_prox.size();
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_dom_prox_), std::max<size_t>(cells * 16, 16)), "DOM proximity map");
        constexpr size_t kPiece = size_t{1} << 20;
        std::vector<uint32_t> fused(4 * std::min(cells, kPiece));
        for (size_t first = 0; first < cells; first += kPiece) {
            const size_t count = std::min(kPiece, cells - first);
            std::fill(fused.begin(), fused.begin() + static_cast<std::ptrdiff_t>(4 * count), 0u);
            for (size_t k = 0; k < count; ++k) {
                const uint32_t w = tables_.dom_prox[first + k], id = w & 0xffffu;
                fused[4 * k] = w;
                if (id != 0xffffu && 4 * static_cast<size_t>(id) + 2 < tables_.dom_centres.size())
                    std::memcpy(&fused[4 * k + 1], &tables_.dom_centres[4 * static_cast<size_t>(id)], 12);
            }
            hip_check(hipMemcpy(d_dom_prox_ + 4 * first, fused.data(), count * 16, hipMemcpyHostToDevice), "DOM proximity map");
        }
    }
    upload(reinterpret_cast<void **>(&d_dom_centres_), tables_.dom_centres.data(), tables_.dom_centres.size() * 4, "DOM centres");
    upload(reinterpret_cast<void **>(&d_dom_named_), tables_.dom_named.data(), tables_.dom_named.size() * 4, "named-DOM records");
    upload(reinterpret_cast<void **>(&d_dom_tx_), G.dom_tx.data(), G.dom_tx.size() * 2, "dom_tx");
    upload(reinterpret_cast<void **>(&d_dom_ty_), G.dom_ty.data(), G.dom_ty.size() * 2, "dom_ty");
    upload(reinterpret_cast<void **>(&d_dom_tz_), G.dom_tz.data(), G.dom_tz.size() * 4, "dom_tz");
    {   // index -> ID tables for assemble_hits_kernel; an ID that does not fit the record's short / ushort keeps the
        // conversion on the host, which reports it when a photon carries it (OpenCL.cxx:1577-1586)
        std::vector<int16_t> sid(G.string_index_to_id.size());
        std::vector<uint32_t> start(G.string_index_to_id.size() + 1);         // (+ the end of the last string's run)
        std::vector<uint16_t> did;
        bool fits = true;
        for (size_t k = 0; k < G.string_index_to_id.size(); ++k) {
            fits = fits && G.string_index_to_id[k] >= -32768 && G.string_index_to_id[k] <= 32767;
            sid[k] = static_cast<int16_t>(G.string_index_to_id[k]);
            start[k] = static_cast<uint32_t>(did.size());
            for (uint32_t v : G.dom_index_to_id[k]) { fits = fits && v <= 65535u; did.push_back(static_cast<uint16_t>(v)); }
        }
        start.back() = static_cast<uint32_t>(did.size());
        if (fits && !sid.empty()) {
            upload(reinterpret_cast<void **>(&d_id_strings_), sid.data(), sid.size() * 2, "string IDs");
            upload(reinterpret_cast<void **>(&d_id_doms_), did.data(), did.size() * 2, "OM IDs");
            upload(reinterpret_cast<void **>(&d_id_dom_start_), start.data(), start.size() * 4, "OM ID offsets");
        }
    }
    if (history_entries_) {
        const size_t bytes = prop_kernel_max_lanes() * history_entries_ * 16;
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_hist_ring_), bytes), "history ring");
        hip_check(hipMemset(d_hist_ring_, 0, bytes), "history ring");
    }
    hip_check(hipMalloc(reinterpret_cast<void **>(&d_rng_x_), max_workitems_ * sizeof(uint64_t)), "rng x");
    hip_check(hipMalloc(reinterpret_cast<void **>(&d_rng_a_), max_workitems_ * sizeof(uint32_t)), "rng a");
    num_slots_ = double_buffering_ ? 2 : 1;
    for (int i = 0; i < num_slots_; ++i) {
        Slot &sl = slots_[i];
        hip_check(hipMalloc(reinterpret_cast<void **>(&sl.d_steps), max_workitems_ * sizeof(DevStep)), "steps");
        hip_check(hipMalloc(reinterpret_cast<void **>(&sl.d_photons), static_cast<size_t>(max_output_photons_) * sizeof(DevPhoton)), "photons");
        hip_check(hipMalloc(reinterpret_cast<void **>(&sl.d_hit_count), 16), "hit counter");
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&sl.h_steps), max_workitems_ * sizeof(clsimhip_step), hipHostMallocDefault), "pinned steps");
        hip_check(hipHostMalloc(reinterpret_cast<void **>(&sl.h_hit_count), 16, hipHostMallocDefault), "pinned counter");
        if (history_entries_) {
            const size_t bytes = static_cast<size_t>(max_output_photons_) * history_entries_ * 16;
            hip_check(hipMalloc(reinterpret_cast<void **>(&sl.d_hist_out), bytes), "photon histories");
            hip_check(hipHostMalloc(reinterpret_cast<void **>(&sl.h_hist), bytes, hipHostMallocDefault), "pinned photon histories");
        }
        hip_check(hipEventCreate(&sl.start), "hipEventCreate");
        hip_check(hipEventCreate(&sl.stop), "hipEventCreate");
        hip_check(hipEventCreateWithFlags(&sl.counted, hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming), "hipEventCreate");
    }
    hip_check(hipMalloc(reinterpret_cast<void **>(&d_queue_), static_cast<size_t>(kQueueWords) * kQueueSlots * sizeof(uint32_t)), "step queue");
    hip_check(hipMalloc(reinterpret_cast<void **>(&d_work_), max_workitems_ * sizeof(WorkRecord)), "work records");
#ifdef CLSIMHIP_DEVELOPER
    // DEVELOPER BUILD ONLY (make DEVELOPER=1; tools/build_variant.sh, tools/scan_env.sh): the round 1-5 environment names, through the
    // same door as clsimhip_set_tuning.  The default build reads none of them.
    {
        static const char *const names[][2] = {
            {"CLSIMHIP_K_NEW", "k_new"}, {"CLSIMHIP_K_SEARCH", "k_search"}, {"CLSIMHIP_SLICES", "slices"}, {"CLSIMHIP_K_POP", "k_pop"},
            {"CLSIMHIP_K_WAIT", "k_wait"}, {"CLSIMHIP_K_AIM", "k_aim"}, {"CLSIMHIP_RESULT_MIN_RECORDS", "result_min_records"},
            {"CLSIMHIP_POOL_R", "pool_ring"}, {"CLSIMHIP_POOL_MIN_STEPS", "pool_min_steps"}, {"CLSIMHIP_GRID", "grid"},
            {"CLSIMHIP_NO_FAST", "generic_kernels"}};
        for (const auto &n : names)
            if (const char *e = std::getenv(n[0])) set_tuning(n[1], std::atoll(e));
        if (const char *e = std::getenv("CLSIMHIP_KERNEL")) set_tuning("kernel", std::strcmp(e, "pool") == 0 ? 1 : 2);
        if (const char *e = std::getenv("CLSIMHIP_POOL_INDEX_BITS")) set_tuning("pool_max_steps", (1ll << std::atoi(e)) - 1);
    }
#endif
    // photon histories are kept per lane and the pooled kernel moves photons between lanes; a very large table image
    // leaves its pools no LDS
    pool_possible_ = history_entries_ == 0 &&
        pool_kernel_fits(static_cast<uint32_t>(tables_.lds_image.size()), stop_detected_ ? 0u : static_cast<uint32_t>(tables_.params.num_strings), tables_.params.num_layers);
    apply_kernel_choice();
}

KParams Converter::launch_params(const void *d_steps, size_t n, size_t rng_offset, void *d_photons, size_t capacity, void *d_hits, hipStream_t stream)
{
    KParams P = tables_.params;
    P.tables = d_tables_;
    P.steps = static_cast<const DevStep *>(d_steps);
    P.n_steps = static_cast<uint32_t>(n);
    P.rng_x = d_rng_x_ + rng_offset;
    P.rng_a = d_rng_a_ + rng_offset;
    P.out = static_cast<DevPhoton *>(d_photons);
    P.hit_count = static_cast<uint32_t *>(d_hits);
    P.max_hits = static_cast<uint32_t>(std::min<size_t>(capacity, 0xffffffffull));
    {   // step queue head for this launch (zeroed in stream order)
        std::lock_guard<std::mutex> lk(ev_mutex_);
        P.queue = d_queue_ + static_cast<size_t>(kQueueWords) * (queue_slot_++ % kQueueSlots);
        last_queue_ = P.queue;
    }
    hip_check(hipMemsetAsync(P.queue, 0, kQueueWords * sizeof(uint32_t), stream), "reset step queue");
    {
        std::lock_guard<std::mutex> lk(tuning_mutex_);
        P.k_new = k_new_;
        P.k_search = k_search_;
        P.slices = k_slices_;
        P.k_pop = k_pop_;
        P.k_wait = k_wait_;
        P.k_aim = k_aim_;
        P.pool_ready = pool_ready_;
    }
    P.chip_share = concurrent_launches_;
#ifdef CLSIMHIP_CENSUS
    if (!d_census_) hip_check(hipMalloc(reinterpret_cast<void **>(&d_census_), kCensusBytes), "census");
    P.census = d_census_;
    hip_check(hipMemsetAsync(P.census, 0, kCensusBytes, stream), "reset census");
    hip_check(hipMemsetAsync(P.census + 8, 0xff, 8, stream), "reset census");
#endif
    P.work = d_work_ + rng_offset;       // work records live with the stream slots: launches on disjoint slots may overlap
    P.len_table = d_len_table_;
    P.prox_map = d_prox_map_;
    P.dom_cells = reinterpret_cast<const uint4 *>(d_dom_prox_);
    P.dom_centres = reinterpret_cast<const float4 *>(d_dom_centres_);
    P.dom_named = reinterpret_cast<const uint4 *>(d_dom_named_);
    P.hist_ring = d_hist_ring_;
    P.hist_out = nullptr;               // set per slot by submit(); the device path has no history output
    P.dom_tx = d_dom_tx_;
    P.dom_ty = d_dom_ty_;
    P.dom_tz = d_dom_tz_;
    return P;
}

// OpenCL.cxx:1525-1544
void Converter::enqueue_steps(const clsimhip_step *steps, size_t n, uint32_t identifier)
{
    need_init();
    check_worker();
    if (!steps) throw Error(CLSIMHIP_ERR_ARGUMENT, "Steps pointer is (null)!");
    if (n == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "Steps are empty!");
    if (n > max_workitems_) throw Error(CLSIMHIP_ERR_ARGUMENT, "Number of steps is greater than maximum number of work items!");
    if (n % workgroup_size_ != 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "The number of steps is not a multiple of the workgroup size!");
    Job job;
    job.id = identifier;
    job.n = n;
    for (size_t i = 0; i < n; ++i) job.generated += steps[i].num_photons;
    job.pinned = StepLease(this, take_step_buffer(n));
    if (job.pinned.b.p) std::memcpy(job.pinned.b.p, steps, n * sizeof(clsimhip_step));
    else job.steps.assign(steps, steps + n);
    in_queue_->put(std::move(job));
}

Converter::StepBuffer Converter::take_step_buffer(size_t steps)
{
    StepBuffer b;
    b.capacity = std::min(max_workitems_, steps + steps / 4);
    if (b.capacity < steps) b.capacity = steps;
    const size_t bytes = b.capacity * sizeof(clsimhip_step);
    StepBuffer retired;                 // a free buffer that is too small, given up for one of the size that is needed
    {
        // one lock scope for "is there a buffer, may another be made, is one to be replaced" and the count (check and
        // increment were in two scopes, so callers side by side could exceed kStepBuffers)
        std::lock_guard<std::mutex> lk(step_pool_mutex_);
        size_t best = free_step_buffers_.size();
        for (size_t i = 0; i < free_step_buffers_.size(); ++i)
            if (free_step_buffers_[i].capacity >= steps && (best == free_step_buffers_.size() || free_step_buffers_[i].capacity < free_step_buffers_[best].capacity)) best = i;
        if (best != free_step_buffers_.size()) {
            const StepBuffer found = free_step_buffers_[best];
            free_step_buffers_.erase(free_step_buffers_.begin() + static_cast<std::ptrdiff_t>(best));
            return found;
        }
        if (step_pinning_refused_) return StepBuffer();
        const bool full = step_buffers_made_ >= kStepBuffers || (step_pool_bytes_ + bytes > kStepPoolBytes && step_buffers_made_ >= 3);
        if (full) {
            // A pool whose buffers were sized by small first bunches (warm-up, tests) must not stay too small for production bunches:
            // the smallest FREE buffer goes and one of the needed size takes its place.  With none free every buffer is in flight:
            // the steps travel in a vector this once.
            if (free_step_buffers_.empty()) {
                if (!step_fallback_logged_) {
                    step_fallback_logged_ = true;
                    std::fprintf(stderr, "clsimhip: all %d page-locked step buffers are in flight; this bunch of %zu steps is staged through pageable memory\n",
                                 step_buffers_made_, steps);
                }
                return StepBuffer();
            }
            size_t smallest = 0;
            for (size_t i = 1; i < free_step_buffers_.size(); ++i)
                if (free_step_buffers_[i].capacity < free_step_buffers_[smallest].capacity) smallest = i;
            retired = free_step_buffers_[smallest];
            free_step_buffers_.erase(free_step_buffers_.begin() + static_cast<std::ptrdiff_t>(smallest));
            step_pool_bytes_ -= retired.capacity * sizeof(clsimhip_step);
            --step_buffers_made_;
            if (step_pool_bytes_ + bytes > kStepPoolBytes && step_buffers_made_ >= 3) {     // (still over the byte budget: the pool shrinks by one)
                free_step_buffers_.push_back(retired);
                step_pool_bytes_ += retired.capacity * sizeof(clsimhip_step);
                ++step_buffers_made_;
                return StepBuffer();
            }
        }
        ++step_buffers_made_;
        step_pool_bytes_ += bytes;
    }
    DeviceGuard on_device(device_);
    if (retired.p) (void)hipHostFree(retired.p);
    if (hipHostMalloc(reinterpret_cast<void **>(&b.p), bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(step_pool_mutex_);
        --step_buffers_made_;
        step_pool_bytes_ -= bytes;
        step_pinning_refused_ = true;
        return StepBuffer();
    }
    return b;
}

void Converter::give_step_buffer(StepBuffer b)
{
    if (!b.p) return;
    std::lock_guard<std::mutex> lk(step_pool_mutex_);
    free_step_buffers_.push_back(b);
}

// OpenCL.cxx:824-934: upload, launch; everything is queued on the compute stream, nothing waits
void Converter::submit(Slot &s, Job &job)
{
    const size_t n = job.n;
    s.id = job.id;
    s.generated = job.generated;
    // (the slot's previous upload finished long ago: its kernel has, or the slot would not be free)
    give_step_buffer(s.step_buffer);
    s.step_buffer = job.pinned.release();
    const clsimhip_step *source = s.step_buffer.p;
    if (!source) {
        std::memcpy(s.h_steps, job.steps.data(), n * sizeof(clsimhip_step));
        source = s.h_steps;
    }
    // the upload has a stream of its own: with double buffering it runs while the previous bunch's kernel does
    hip_check(hipMemcpyAsync(s.d_steps, source, n * sizeof(DevStep), hipMemcpyHostToDevice, upload_stream_), "upload steps");
    hip_check(hipEventRecord(s.uploaded, upload_stream_), "event");
    hip_check(hipStreamWaitEvent(stream_, s.uploaded, 0), "wait for the upload");
    hip_check(hipMemsetAsync(s.d_hit_count, 0, 4, stream_), "reset hit counter");
    KParams P = launch_params(s.d_steps, n, 0, s.d_photons, max_output_photons_, s.d_hit_count, stream_);
    P.hist_out = s.d_hist_out;
    P.id_strings = d_id_strings_; P.id_doms = d_id_doms_; P.id_dom_start = d_id_dom_start_;
    hip_check(hipEventRecord(s.start, stream_), "event");
    hip_check(launch(P, stream_), "propagation kernel launch");
    hip_check(hipEventRecord(s.stop, stream_), "event");
    hip_check(hipMemcpyAsync(s.h_hit_count, s.d_hit_count, 4, hipMemcpyDeviceToHost, stream_), "download hit counter");
    hip_check(hipMemcpyAsync(s.h_hit_count + 1, P.queue + 2, 12, hipMemcpyDeviceToHost, stream_), "download skipped-step and bad-record counters");
    hip_check(hipEventRecord(s.counted, stream_), "event");
}

// OpenCL.cxx:994-1086: wait for the bunch, download and convert its photons (on the copy stream, so that the
// next bunch's kernel -- already queued on the compute stream -- runs meanwhile), hand the result out
void Converter::finish(Slot &s, std::chrono::steady_clock::time_point &last_done, bool &first)
{
    hip_check(hipEventSynchronize(s.counted), "propagation kernel");
    uint32_t hits = *s.h_hit_count;
    if (s.h_hit_count[1] != 0)
        std::fprintf(stderr, "clsimhip: %u steps of bunch %u have non-finite position/direction/length/beta or a source type without spectrum and were not propagated\n",
                     s.h_hit_count[1], s.id);
    if (s.h_hit_count[3] != 0)           // (queue[4]: assemble_hits_kernel met records whose string / DOM indices name no DOM)
        throw Error(CLSIMHIP_ERR_DEVICE, "photon record with out-of-range string/DOM index");
    if (hits > max_output_photons_) {
        // OpenCL.cxx:1027-1032: logged, truncated
        std::fprintf(stderr, "clsimhip: maximum number of photons exceeded, only receiving %u of %u photons\n", max_output_photons_, hits);
        hits = max_output_photons_;
    }
    // The download lands in a page-locked buffer of the result pool, which then IS the result (no copy, no conversion on this
    // thread: round 2 zero-filled a vector per result, copied the download into it and converted indices one by one -- 0.4 s
    // per bunch of 8.4 M photons in the reference's benchmark, the device idle 80 % of the time).  With every pool buffer in
    // the caller's hands the records go into a plain vector.
    Result r;
    r.count = hits;
    std::unique_ptr<std::vector<clsimhip_photon>> photons;
    clsimhip_photon *where = nullptr;
    if (hits) {
        const PinnedBuffer buf = take_result_buffer(hits);
        if (buf.p) {
            r.pinned.reset(buf.p);
            r.pinned_capacity = buf.capacity;
            where = buf.p;
        } else {
            // every pool buffer is with the caller (or the host refuses to page-lock more): the download goes into pageable memory,
            // which is correct and slower -- said once, because nothing else shows it
            static std::once_flag noted;
            std::call_once(noted, [] {
                std::fprintf(stderr, "clsimhip: every page-locked result buffer is in the caller's hands (release them with "
                                     "clsimhip_release_result); photons are downloaded into pageable memory until one is free\n");
            });
            photons.reset(new std::vector<clsimhip_photon>(hits));
            where = photons->data();
        }
        try {
            hip_check(hipMemcpyAsync(where, s.d_photons, static_cast<size_t>(hits) * sizeof(DevPhoton), hipMemcpyDeviceToHost, copy_stream_), "download photons");
            hip_check(hipStreamSynchronize(copy_stream_), "download photons");
        } catch (...) {
            // the pool buffer goes back to the pool (it stays counted in result_buffers_made_ and stays usable), not to hipHostFree
            if (r.pinned) {
                std::lock_guard<std::mutex> lk(result_pool_mutex_);
                free_result_buffers_.push_back(PinnedBuffer{r.pinned.release(), r.pinned_capacity});
            }
            throw;
        }
        if (!d_id_strings_) replace_indices(where, hits);      // OpenCL.cxx:1604-1619 does this on the caller thread
    }
    std::unique_ptr<std::vector<float>> histories;
    if (hits && history_entries_) {
        // ConvertPhotonHistories (OpenCL.cxx:940-989): unroll each ring into forward order
        const size_t N = history_entries_;
        hip_check(hipMemcpyAsync(s.h_hist, s.d_hist_out, static_cast<size_t>(hits) * N * 16, hipMemcpyDeviceToHost, copy_stream_), "download photon histories");
        hip_check(hipStreamSynchronize(copy_stream_), "download photon histories");
        histories.reset(new std::vector<float>(static_cast<size_t>(hits) * N * 4, 0.f));
        for (size_t i = 0; i < hits; ++i) {
            const uint32_t num_scatters = where[i].num_scatters;
            if (num_scatters == 0) continue;
            const size_t recorded = std::min<size_t>(num_scatters, N);
            size_t cur = (num_scatters <= N) ? 0 : (num_scatters % N);
            for (size_t j = 0; j < recorded; ++j) {
                std::memcpy(&(*histories)[(i * N + j) * 4], &s.h_hist[(i * N + cur) * 4], 16);
                if (++cur >= N) cur = 0;
            }
        }
    }
    float ms = 0.f;
    hip_check(hipEventElapsedTime(&ms, s.start, s.stop), "event time");
    const auto now = std::chrono::steady_clock::now();
    {
        std::lock_guard<std::mutex> lk(stats_mutex_);
        total_device_ns_ += static_cast<uint64_t>(static_cast<double>(ms) * 1e6);
        if (!first) total_host_ns_ += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(now - last_done).count());
        else total_host_ns_ += static_cast<uint64_t>(static_cast<double>(ms) * 1e6);
        ++num_kernel_calls_;
        photons_generated_ += s.generated;
        photons_at_doms_ += hits;
    }
    first = false;
    last_done = now;
    r.id = s.id;
    r.photons = std::move(photons);
    r.histories = std::move(histories);
    out_queue_->put(std::move(r));
}

// OpenCL.cxx:1142-1315 (thread body).  With double buffering the worker keeps one bunch on the GPU and one in
// post-processing; it never holds a finished bunch back waiting for more input.
void Converter::worker()
{
    (void)hipSetDevice(device_);
    auto last_done = std::chrono::steady_clock::now();
    bool first = true;
    int cur = 0, pending = -1;
    Job job;
    try {
        for (;;) {
            bool have;
            if (pending < 0) {
                have = in_queue_->get(job);
                if (!have) break;
            } else {
                // a bunch is running: take the next one as soon as it arrives, but stop waiting when the kernel is done
                have = false;
                while (!(have = in_queue_->get_for(job, 200))) {
                    if (hipEventQuery(slots_[pending].counted) == hipSuccess || in_queue_->closed()) break;
                }
                if (!have) {
                    finish(slots_[pending], last_done, first);
                    pending = -1;
                    continue;
                }
            }
            submit(slots_[cur], job);
            if (pending >= 0) finish(slots_[pending], last_done, first);
            if (num_slots_ == 2) {
                pending = cur;
                cur ^= 1;
            } else {
                finish(slots_[cur], last_done, first);
            }
        }
        if (pending >= 0) finish(slots_[pending], last_done, first);
    } catch (const Error &e) {
        // The reference's worker log_fatal()s on device errors, which ends the process (OpenCL.cxx:768-774).  A C
        // library must not take its host down: the error is logged and kept, both queues are closed so that blocked
        // callers wake up, and every later EnqueueSteps / GetConversionResult returns CLSIMHIP_ERR_DEVICE with this
        // text (the C++ adapter turns that into the exception / log_fatal of the host framework).
        std::fprintf(stderr, "clsimhip: fatal device error in worker thread: %s\n", e.what());
        {
            std::lock_guard<std::mutex> lk(fatal_mutex_);
            fatal_error_ = e.what();
            worker_failed_ = true;
        }
        in_queue_->close();
        out_queue_->close();
    }
}

// OpenCL.cxx:1604-1619
void Converter::get_result(uint32_t *identifier, const clsimhip_photon **photons, size_t *n)
{
    need_init();
    if (!identifier || !photons || !n) throw Error(CLSIMHIP_ERR_ARGUMENT, "output pointers are (null)");
    Result r;
    if (!out_queue_->get(r)) {
        check_worker();
        throw Error(CLSIMHIP_ERR_STATE, "converter is shutting down");
    }
    *identifier = r.id;
    *n = r.count;
    static const clsimhip_photon empty_sentinel{};
    const clsimhip_photon *key = r.count ? r.data() : nullptr;
    *photons = key ? key : &empty_sentinel;
    if (key) {
        std::lock_guard<std::mutex> lk(results_mutex_);
        handed_out_[key] = std::move(r);
    }
}

void Converter::result_histories(const clsimhip_photon *photons, const float **histories, uint32_t *entries)
{
    need_init();
    if (!histories || !entries) throw Error(CLSIMHIP_ERR_ARGUMENT, "output pointers are (null)");
    *histories = nullptr;
    *entries = history_entries_;
    std::lock_guard<std::mutex> lk(results_mutex_);
    auto it = handed_out_.find(photons);
    if (it == handed_out_.end()) {
        if (photons) throw Error(CLSIMHIP_ERR_ARGUMENT, "not a result handed out by GetConversionResult (or already released)");
        return;
    }
    if (it->second.histories) *histories = it->second.histories->data();
}

void Converter::release_result(const clsimhip_photon *photons)
{
    PinnedBuffer back;
    {
        std::lock_guard<std::mutex> lk(results_mutex_);
        auto it = handed_out_.find(photons);
        if (it == handed_out_.end()) return;
        back.capacity = it->second.pinned_capacity;
        back.p = it->second.pinned.release();
        handed_out_.erase(it);
    }
    if (back.p) {
        std::lock_guard<std::mutex> lk(result_pool_mutex_);
        free_result_buffers_.push_back(back);
    }
}

void Converter::Result::HostFree::operator()(clsimhip_photon *p) const { if (p) (void)hipHostFree(p); }

Converter::PinnedBuffer Converter::take_result_buffer(size_t records)
{
    clsimhip_photon *too_small = nullptr;
    {
        std::lock_guard<std::mutex> lk(result_pool_mutex_);
        // the smallest free buffer that holds the records
        size_t best = free_result_buffers_.size(), smallest = free_result_buffers_.size();
        for (size_t i = 0; i < free_result_buffers_.size(); ++i) {
            const size_t c = free_result_buffers_[i].capacity;
            if (c >= records && (best == free_result_buffers_.size() || c < free_result_buffers_[best].capacity)) best = i;
            if (smallest == free_result_buffers_.size() || c < free_result_buffers_[smallest].capacity) smallest = i;
        }
        if (best != free_result_buffers_.size()) {
            const PinnedBuffer b = free_result_buffers_[best];
            free_result_buffers_.erase(free_result_buffers_.begin() + static_cast<std::ptrdiff_t>(best));
            return b;
        }
        if (pinning_refused_) return PinnedBuffer();
        if (result_buffers_made_ >= kResultBuffers) {
            if (free_result_buffers_.empty()) return PinnedBuffer();            // all of them are with the caller
            too_small = free_result_buffers_[smallest].p;                       // a free one makes room for a larger one
            free_result_buffers_.erase(free_result_buffers_.begin() + static_cast<std::ptrdiff_t>(smallest));
            --result_buffers_made_;
        }
        ++result_buffers_made_;
    }
    if (too_small) (void)hipHostFree(too_small);
    PinnedBuffer b;
    b.capacity = std::max(min_result_records_, (records + records / 4 + 4095) / 4096 * 4096);
    b.capacity = std::max(records, std::min(b.capacity, static_cast<size_t>(max_output_photons_)));
    if (hipHostMalloc(reinterpret_cast<void **>(&b.p), b.capacity * sizeof(clsimhip_photon), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(result_pool_mutex_);
        --result_buffers_made_;
        pinning_refused_ = true;
        return PinnedBuffer();
    }
    return b;
}

size_t Converter::queue_size() const { need_init(); return in_queue_->size(); }
bool Converter::more_photons_available() const { need_init(); return !out_queue_->empty(); }

void Converter::statistics(double out[8]) const
{
    std::lock_guard<std::mutex> lk(stats_mutex_);
    const double dev = static_cast<double>(total_device_ns_), host = static_cast<double>(total_host_ns_);
    const double gen = static_cast<double>(photons_generated_);
    out[0] = dev; out[1] = host; out[2] = static_cast<double>(num_kernel_calls_); out[3] = gen;
    out[4] = static_cast<double>(photons_at_doms_);
    out[5] = dev / gen; out[6] = host / gen; out[7] = dev / host;
}

double Converter::option(int which) const
{
    switch (which) {
    case CLSIMHIP_OPTION_ENABLE_DOUBLE_BUFFERING: return double_buffering_ ? 1. : 0.;
    case CLSIMHIP_OPTION_DOUBLE_PRECISION: return double_precision_ ? 1. : 0.;
    case CLSIMHIP_OPTION_STOP_DETECTED_PHOTONS: return stop_detected_ ? 1. : 0.;
    case CLSIMHIP_OPTION_SAVE_ALL_PHOTONS: return save_all_ ? 1. : 0.;
    case CLSIMHIP_OPTION_SAVE_ALL_PHOTONS_PRESCALE: return save_all_prescale_;
    case CLSIMHIP_OPTION_FIXED_NUMBER_OF_ABSORPTION_LENGTHS: return fixed_abs_lengths_;
    case CLSIMHIP_OPTION_DOM_PANCAKE_FACTOR: return pancake_;
    case CLSIMHIP_OPTION_PHOTON_HISTORY_ENTRIES: return static_cast<double>(history_entries_);
    }
    throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown option");
}

// OpenCL.cxx:1565-1600
void Converter::replace_indices(clsimhip_photon *photons, size_t n) const
{
    const GeoTables &G = tables_.geo;
    for (size_t i = 0; i < n; ++i) {
        const int16_t s = photons[i].string_id;
        const uint16_t d = photons[i].om_id;
        if (s < 0 || static_cast<size_t>(s) >= G.string_index_to_id.size() || d >= G.dom_index_to_id[s].size())
            throw Error(CLSIMHIP_ERR_DEVICE, "photon record with out-of-range string/DOM index");
        const int string_id = G.string_index_to_id[s];
        const unsigned dom_id = G.dom_index_to_id[s][d];
        if (string_id < -32768 || string_id > 32767)
            throw Error(CLSIMHIP_ERR_CONFIG, "Your detector I3Geometry uses a string ID \"" + std::to_string(string_id) + "\". Large IDs like that are currently not supported by clsim.");
        if (dom_id > 65535u)
            throw Error(CLSIMHIP_ERR_CONFIG, "Your detector I3Geometry uses a OM ID \"" + std::to_string(dom_id) + "\". Large IDs like that are currently not supported by clsim.");
        photons[i].string_id = static_cast<int16_t>(string_id);
        photons[i].om_id = static_cast<uint16_t>(dom_id);
    }
}

void Converter::propagate_device(const void *d_steps, size_t n, size_t rng_offset, void *d_photons, size_t capacity,
                                 void *d_hit_count, hipStream_t stream)
{
    need_init();
    if (!d_steps || !d_photons || !d_hit_count) throw Error(CLSIMHIP_ERR_ARGUMENT, "device pointers are (null)");
    if (n == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "Steps are empty!");
    if (rng_offset + n > max_workitems_) throw Error(CLSIMHIP_ERR_ARGUMENT, "Number of steps is greater than maximum number of work items!");
    if (history_entries_) throw Error(CLSIMHIP_ERR_STATE, "photon histories are only delivered through EnqueueSteps/GetConversionResult");
    DeviceGuard on_device(device_);
    std::pair<hipEvent_t, hipEvent_t> ev;
    {
        std::lock_guard<std::mutex> lk(ev_mutex_);
        if (!free_events_.empty()) { ev = free_events_.back(); free_events_.pop_back(); }
        else { hip_check(hipEventCreate(&ev.first), "hipEventCreate"); hip_check(hipEventCreate(&ev.second), "hipEventCreate"); }
    }
    hip_check(hipMemsetAsync(d_hit_count, 0, 4, stream), "reset hit counter");
    const KParams P = launch_params(d_steps, n, rng_offset, d_photons, capacity, d_hit_count, stream);
    hip_check(hipEventRecord(ev.first, stream), "event");
    hip_check(launch(P, stream), "propagation kernel launch");
    hip_check(hipEventRecord(ev.second, stream), "event");
    std::lock_guard<std::mutex> lk(ev_mutex_);
    pending_events_.push_back(ev);
}

namespace {
struct Scratch {                       // device memory of one tester call
    void *p = nullptr;
    explicit Scratch(size_t bytes)
    {
        const hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    ~Scratch() { if (p) (void)hipFree(p); }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
};
} // namespace

void Converter::eval_device_function(int what, int layer, bool fast, const float *in4, size_t n, float *out4)
{
    need_init();
    if (!in4 || !out4) throw Error(CLSIMHIP_ERR_ARGUMENT, "in4 / out4 are (null)");
    if (what < CLSIMHIP_EVAL_LENGTHS || what > CLSIMHIP_EVAL_POST_SCATTER_TRANSFORM) throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown function");
    if (what == CLSIMHIP_EVAL_LENGTHS && (layer < 0 || layer >= tables_.params.num_layers)) throw Error(CLSIMHIP_ERR_ARGUMENT, "no such layer");
    if (fast && !tables_.variant.fast) throw Error(CLSIMHIP_ERR_STATE, "this configuration has no FAST instantiation (Compile() could not prove its ranges)");
    if (n > 0xffffffffull) throw Error(CLSIMHIP_ERR_ARGUMENT, "too many points");
    DeviceGuard on_device(device_);
    Scratch in(n * 16), out(n * 16);
    hip_check(hipMemcpy(in.p, in4, n * 16, hipMemcpyHostToDevice), "hipMemcpy");
    KParams P = tables_.params;
    P.tables = d_tables_;
    P.len_table = d_len_table_;
    hip_check(launch_eval_function(P, tables_.variant.lengths, tables_.variant.tilt, fast, what, layer, static_cast<const float4 *>(in.p), static_cast<uint32_t>(n),
                                   static_cast<float4 *>(out.p), nullptr), "function tester launch");
    hip_check(hipDeviceSynchronize(), "function tester");
    hip_check(hipMemcpy(out4, out.p, n * 16, hipMemcpyDeviceToHost), "hipMemcpy");
}

void Converter::eval_device_random(int what, int generator, bool fast, uint64_t *x, const uint32_t *a, size_t n_streams, size_t draws, float *out)
{
    need_init();
    if (!x || !a || !out) throw Error(CLSIMHIP_ERR_ARGUMENT, "x / a / out are (null)");
    if (what < CLSIMHIP_EVAL_RANDOM_UNIFORM || what > CLSIMHIP_EVAL_RANDOM_SCATTERING_COSINE) throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown distribution");
    if (what == CLSIMHIP_EVAL_RANDOM_WAVELENGTH && (generator < 0 || generator >= tables_.params.num_gen)) throw Error(CLSIMHIP_ERR_ARGUMENT, "no such wavelength generator");
    if (fast && !tables_.variant.fast) throw Error(CLSIMHIP_ERR_STATE, "this configuration has no FAST instantiation (Compile() could not prove its ranges)");
    if (n_streams > 0xffffffffull || draws > 0xffffffffull) throw Error(CLSIMHIP_ERR_ARGUMENT, "too many draws");
    DeviceGuard on_device(device_);
    Scratch dx(n_streams * 8), da(n_streams * 4), dout(n_streams * draws * 4);
    hip_check(hipMemcpy(dx.p, x, n_streams * 8, hipMemcpyHostToDevice), "hipMemcpy");
    hip_check(hipMemcpy(da.p, a, n_streams * 4, hipMemcpyHostToDevice), "hipMemcpy");
    KParams P = tables_.params;
    P.tables = d_tables_;
    P.len_table = d_len_table_;
    hip_check(launch_eval_random(P, fast, what, generator, static_cast<uint64_t *>(dx.p), static_cast<const uint32_t *>(da.p), static_cast<uint32_t>(n_streams),
                                 static_cast<uint32_t>(draws), static_cast<float *>(dout.p), nullptr), "distribution tester launch");
    hip_check(hipDeviceSynchronize(), "distribution tester");
    hip_check(hipMemcpy(out, dout.p, n_streams * draws * 4, hipMemcpyDeviceToHost), "hipMemcpy");
    hip_check(hipMemcpy(x, dx.p, n_streams * 8, hipMemcpyDeviceToHost), "hipMemcpy");
}

void Converter::kernel_time(bool reset, double *total_ms, uint64_t *launches)
{
    std::lock_guard<std::mutex> lk(ev_mutex_);
    for (auto &ev : pending_events_) {
        hip_check(hipEventSynchronize(ev.second), "hipEventSynchronize");
        float ms = 0.f;
        hip_check(hipEventElapsedTime(&ms, ev.first, ev.second), "hipEventElapsedTime");
        dev_total_ms_ += ms;
        ++dev_launches_;
        free_events_.push_back(ev);
    }
    pending_events_.clear();
    if (total_ms) *total_ms = dev_total_ms_;
    if (launches) *launches = dev_launches_;
    if (reset) { dev_total_ms_ = 0; dev_launches_ = 0; }
}

long Converter::get_table(const std::string &name, double *out, size_t cap) const
{
    if (!compiled_) throw Error(CLSIMHIP_ERR_STATE, "not compiled");
    if (name == "dom_proximity_map") {                  // 16 MB of bytes: converted on request, not kept as doubles
        const size_t n = tables_.dom_prox.size();
        if (out) for (size_t i = 0; i < std::min(n, cap); ++i) out[i] = tables_.dom_prox[i];
        return static_cast<long>(n);
    }
    if (name == "dom_named") {
        const size_t n = tables_.dom_named.size();
        if (out) for (size_t i = 0; i < std::min(n, cap); ++i) out[i] = tables_.dom_named[i];
        return static_cast<long>(n);
    }
    if (name == "dom_centres") {
        const size_t n = tables_.dom_centres.size();
        if (out) for (size_t i = 0; i < std::min(n, cap); ++i) out[i] = tables_.dom_centres[i];
        return static_cast<long>(n);
    }
    const auto it = tables_.named.find(name);
    if (it == tables_.named.end()) throw Error(CLSIMHIP_ERR_ARGUMENT, "no table named " + name);
    const size_t n = it->second.size();
    if (out) std::memcpy(out, it->second.data(), std::min(n, cap) * sizeof(double));
    return static_cast<long>(n);
}

void Converter::debug_counters(uint32_t out[4])
{
    need_init();
    DeviceGuard on_device(device_);
    hip_check(hipDeviceSynchronize(), "sync");
#ifdef CLSIMHIP_CENSUS
    hip_check(hipMemcpy(out, d_census_, kCensusBytes, hipMemcpyDeviceToHost), "download census");     // the caller passes kCensusBytes (4 MiB)
#else
    if (last_queue_) hip_check(hipMemcpy(out, last_queue_, 16, hipMemcpyDeviceToHost), "download counters");
#endif
}

void Converter::get_rng_state(uint64_t *x, size_t count)
{
    need_init();
    if (!x || count > max_workitems_) throw Error(CLSIMHIP_ERR_ARGUMENT, "bad rng state request");
    DeviceGuard on_device(device_);
    hip_check(hipDeviceSynchronize(), "sync");
    hip_check(hipMemcpy(x, d_rng_x_, count * sizeof(uint64_t), hipMemcpyDeviceToHost), "download rng state");
}

} // namespace clsimhip
