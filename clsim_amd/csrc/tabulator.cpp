// See tabulator.h.  Host logic restated from private/clsim/tabulator/ (file:line at each function).
#include "tabulator.h"
#include "lightsource.h"

#include <cstdio>
#include <cstring>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace clsimhip {

namespace {
constexpr uint32_t kQueueSlots = 256;
constexpr double kFixedAbsorptionLengths = 42.;      // StepToTableConverter.cxx:182
}

// Axis.cxx:93-103 (linear), 131-141 (power)
double AxisData::transform(double v) const { return kind == CLSIMHIP_AXIS_LINEAR ? v : std::pow(v, static_cast<double>(power)); }
double AxisData::inverse(double v) const { return kind == CLSIMHIP_AXIS_LINEAR ? v : std::pow(v, 1. / power); }
// Axis.cxx:76-83
double AxisData::bin_edge(unsigned i) const
{
    const double imin = inverse(min), imax = inverse(max);
    const double istep = (imax - imin) / n_bins;
    return transform(imin + i * istep);
}

void Tabulator::hip_check(hipError_t e, const char *what) const
{
    if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

// GetMinimumRefractiveIndex (StepToTableConverter.cxx:96-120), as written: the scan point is wmin + i*(wmax-wmin)
static std::pair<double, double> minimum_refractive_index(const MediumData &m)
{
    std::pair<double, double> best(INFINITY, INFINITY);
    if (m.group_kind == CLSIMHIP_REFINDEX_DISPERSION)           // :103-104
        throw Error(CLSIMHIP_ERR_CONFIG, "Medium properties don't know how to calculate group refractive indices");
    auto group = [&](double w) {
        if (m.group_kind == CLSIMHIP_REFINDEX_TABLE) return m.group_table.eval(w);
        const double x = w / units::micrometer;                 // RefIndexIceCube.cxx:84-101
        const double np = m.n[0] + x * (m.n[1] + x * (m.n[2] + x * (m.n[3] + x * m.n[4])));
        const double corr = m.g[0] + x * (m.g[1] + x * (m.g[2] + x * (m.g[3] + x * m.g[4])));
        return np * corr;
    };
    double gmin = -INFINITY, gmax = INFINITY;
    if (m.group_kind == CLSIMHIP_REFINDEX_TABLE) {
        gmin = m.group_table.start;
        gmax = m.group_table.start + m.group_table.step * static_cast<double>(m.group_table.values.size() - 1);
    }
    const double wmin = std::max(m.min_wlen, gmin), wmax = std::min(m.max_wlen, gmax);
    for (unsigned i = 0; i < 1000; i++) {                       // the same for every layer
        const double w = wmin + i * (wmax - wmin);
        const double n = group(w);
        if (n > 1 && n < best.first) best = std::make_pair(n, m.phase_ref_index(w));
    }
    return best;
}

Tabulator::Tabulator(int device, int axes_kind, std::vector<AxisData> axes, bool store_squared_weights, const MediumData &medium,
                     const FunctionData &wavelength_acceptance, const PolynomialData &angular, double reference_area,
                     double step_length, const uint64_t *x, const uint32_t *a, size_t streams)
    : device_(device), axes_kind_(axes_kind), axes_(std::move(axes)), squared_(store_squared_weights),
      reference_area_(reference_area), step_length_(step_length), streams_(streams)
{
    if (axes_kind_ != CLSIMHIP_AXES_SPHERICAL && axes_kind_ != CLSIMHIP_AXES_CYLINDRICAL) throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown axes kind");
    // 4 axes, or 5 = TABULATE_IMPACT_ANGLE (StepToTableConverter.cxx:187-188)
    if (axes_.size() != 4 && axes_.size() != 5) throw Error(CLSIMHIP_ERR_CONFIG, "a table has 4 axes, or 5 with the impact angle");
    for (const AxisData &ax : axes_) {
        if (ax.n_bins == 0 || !(ax.max > ax.min)) throw Error(CLSIMHIP_ERR_ARGUMENT, "axis needs bins and max > min");
        if (ax.kind == CLSIMHIP_AXIS_POWER && ax.power < 1)
            throw Error(CLSIMHIP_ERR_CONFIG, "a power axis needs a power >= 1");
    }
    if (!x || !a || streams == 0 || streams % 256 != 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "RNG streams: need a non-zero multiple of 256");
    if (angular.coefficients.size() > 64) throw Error(CLSIMHIP_ERR_ARGUMENT, "too many polynomial coefficients");
    if (!(step_length > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "step length must be positive");

    // Axes::Axes (Axes.cxx:51-64): every axis has an under- and an overflow bin
    const size_t nd = axes_.size();
    shape_.assign(nd, 0); strides_.assign(nd, 0);
    shape_[nd - 1] = axes_[nd - 1].n_bins + 2; strides_[nd - 1] = 1;
    for (size_t i = nd - 1; i-- > 0;) { shape_[i] = axes_[i].n_bins + 2; strides_[i] = strides_[i + 1] * shape_[i + 1]; }
    n_bins_ = strides_[0] * shape_[0];
    if (n_bins_ >= 0xffffffffull) throw Error(CLSIMHIP_ERR_CONFIG, "table has more than 2^32 bins");
    // The device's own bin order for four axes (round 5): 4 x 2 x 1 bins of axes 0, 2 and 3 -- distance, polar angle, time -- share one
    // 64-byte sector; azimuth (axis 1) stays whole.  The table's sums are memory-side atomic requests of one sector each, and a path's
    // consecutive samples then meet fewer sectors than with eight time bins to a sector: the distance and polar-angle bins are the
    // ones a path crosses (profiles/r05/tab_tile_scan.txt: 2 x 2 x 2 1.5 % behind, eight time bins 14 %).  bin_content_double() puts the
    // sums into the reference's order (Axes.cxx:51-64).  (Developer build: CLSIMHIP_TAB_LAYOUT=linear keeps the reference's order on the
    // device too, CLSIMHIP_TAB_TILE=e0e2e3 another tile shape -- the measurements behind profiles/r05/tab_tile_scan.txt; five-axis tables
    // are always linear.)
    tiled_ = (nd == 4);
    n_device_bins_ = n_bins_;
#ifdef CLSIMHIP_DEVELOPER
    if (const char *e = std::getenv("CLSIMHIP_TAB_LAYOUT")) tiled_ = tiled_ && (std::strcmp(e, "linear") != 0);
    if (const char *e = std::getenv("CLSIMHIP_TAB_TILE")) {
        // (the tile's shape as three digits e0 e2 e3 with e0 + e2 + e3 = 3, e.g. 210 = 4 x 2 x 1 bins of distance, polar angle, time)
        if (std::strlen(e) == 3 && e[0] >= '0' && e[1] >= '0' && e[2] >= '0' && (e[0] - '0') + (e[1] - '0') + (e[2] - '0') == 3)
            for (int k = 0; k < 3; ++k) tile_bits_[k] = static_cast<unsigned>(e[k] - '0');
    }
    if (const char *e = std::getenv("CLSIMHIP_TAB_FAST")) fast_kernels_ = (e[0] == '1');
    if (const char *e = std::getenv("CLSIMHIP_GRID")) grid_ = std::atoi(e);
#endif
    if (tiled_) {
        const size_t t0 = size_t(1) << tile_bits_[0], t2 = size_t(1) << tile_bits_[1], t3 = size_t(1) << tile_bits_[2];
        const size_t h0 = (shape_[0] + t0 - 1) / t0, h2 = (shape_[2] + t2 - 1) / t2, h3 = (shape_[3] + t3 - 1) / t3;
        tile_stride_[2] = h3 * 8; tile_stride_[1] = h2 * tile_stride_[2]; tile_stride_[0] = shape_[1] * tile_stride_[1];
        n_device_bins_ = h0 * tile_stride_[0];
        if (n_device_bins_ >= 0xffffffffull) throw Error(CLSIMHIP_ERR_CONFIG, "table has more than 2^32 bins");
    }

    {   // spectralBiasFactor_ (StepToTableConverter.cxx:142-152): photons of a bare Cherenkov spectrum between 300 and 600 nm per
        // photon drawn from the acceptance-weighted spectrum (ConverterUtils.cxx:44-105; lightsource.cpp)
        FunctionData one;
        one.kind = CLSIMHIP_FUNCTION_CONSTANT;
        one.value = 1.;
        spectral_bias_factor_ = photons_per_meter(medium, one, 300e-9, 600e-9) / photons_per_meter(medium, wavelength_acceptance, medium.min_wlen, medium.max_wlen);
    }
    // StepToTableConverter.cxx:126-141: generator 0 = Cherenkov spectrum biased with the wavelength acceptance
    std::vector<RandomValueData> gens(1, make_cherenkov_generator(wavelength_acceptance, medium));
    tables_ = compile_tables(medium, GeometryInput(), gens, wavelength_acceptance, 1.0);
    KParams &P = tables_.params;
    tables_.variant.tabulate = true;
    tables_.variant.flasher = true;
    P.has_fixed_abs = 1;
    P.fixed_abs = to_float_literal(kFixedAbsorptionLengths);
    P.tab_axes_kind = axes_kind_;
    P.tab_full_azimuth = (axes_kind_ == CLSIMHIP_AXES_SPHERICAL && axes_[1].max > 180.) ? 1 : 0;     // Axes.cxx:96-97
    P.tab_ndim = static_cast<int32_t>(nd);
    for (size_t k = 0; k < 5; ++k) { P.tab_scale[k] = P.tab_offset[k] = 0.f; P.tab_inverse[k] = 0; P.tab_inv_exp[k] = 1.f; P.tab_nbins[k] = 0; P.tab_stride[k] = 0; }
    for (size_t k = 0; k < nd; ++k) {
        // Axis::GetIndexCode (Axis.cxx:45-60)
        const AxisData &ax = axes_[k];
        const double scale = ax.n_bins / (ax.inverse(ax.max) - ax.inverse(ax.min));
        const double offset = scale * ax.inverse(ax.min);
        P.tab_scale[k] = to_float_literal(scale);
        P.tab_offset[k] = to_float_literal(offset);
        P.tab_inverse[k] = (ax.kind == CLSIMHIP_AXIS_POWER) ? static_cast<int32_t>(ax.power) : 0;
        P.tab_inv_exp[k] = (ax.kind == CLSIMHIP_AXIS_POWER) ? to_float_literal(1. / ax.power) : 1.f;
        P.tab_nbins[k] = static_cast<int32_t>(ax.n_bins);
        P.tab_stride[k] = static_cast<uint32_t>(strides_[k]);
    }
    P.tab_tiled = tiled_ ? 1u : 0u;
    for (int k = 0; k < 3; ++k) P.tab_tile_stride[k] = tiled_ ? static_cast<uint32_t>(tile_stride_[k]) : 0u;
    for (int k = 0; k < 3; ++k) P.tab_tile_bits[k] = tile_bits_[k];
    // the reference's default table (python/tablemaker/tabulator.py:621-641) and every table of its shape: the specialised sampler (kparams.h: tab_std)
    P.tab_std = (axes_kind_ == CLSIMHIP_AXES_SPHERICAL && nd == 4 && !P.tab_full_azimuth && tiled_ && tile_bits_[0] == 2 && tile_bits_[1] == 1 &&
                 tile_bits_[2] == 0 && !squared_ && P.tab_inverse[0] == 2 && P.tab_inverse[1] <= 1 && P.tab_inverse[2] <= 1 && P.tab_inverse[3] == 2) ? 1u : 0u;
    tables_.named["TABULATOR_STANDARD_SAMPLER"] = {double(P.tab_std)};
    P.tab_max0 = to_float_literal(axes_[0].max);
    P.tab_max3 = to_float_literal(axes_[3].max);
    const auto n_min = minimum_refractive_index(medium);
    n_group_ = n_min.first; n_phase_ = n_min.second;
    P.tab_min_inv_groupvel = to_float_literal(n_group_ / units::c_light);                    // :192-193
    P.tab_tan_thetac = to_float_literal(std::sqrt(n_phase_ * n_phase_ - 1.));                 // :194-195
    P.tab_volume_step = to_float_literal(step_length_);                                       // :191
    {   // sampling-loop constants (kparams.h: off_tab)
        std::vector<uint32_t> &img = tables_.lds_image;
        while (img.size() % 4) img.push_back(0u);
        P.off_tab = static_cast<uint32_t>(img.size());
        auto putf = [&](float f) { uint32_t u; std::memcpy(&u, &f, 4); img.push_back(u); };
        for (int k = 0; k < 5; ++k) putf(P.tab_scale[k]);
        for (int k = 0; k < 5; ++k) putf(P.tab_offset[k]);
        for (int k = 0; k < 5; ++k) img.push_back(static_cast<uint32_t>(P.tab_nbins[k]));
        for (int k = 0; k < 5; ++k) img.push_back(P.tab_stride[k]);
        for (int k = 0; k < 5; ++k) img.push_back(static_cast<uint32_t>(P.tab_inverse[k]));
        putf(P.tab_max0); putf(P.tab_max3); putf(P.tab_min_inv_groupvel); putf(P.tab_tan_thetac); putf(P.tab_volume_step);
        img.push_back(static_cast<uint32_t>(nd));
        while (img.size() % 4) img.push_back(0u);
    }
    {   // getAngularAcceptance: coefficients are appended to the LDS image
        std::vector<uint32_t> &img = tables_.lds_image;
        P.off_ang = static_cast<uint32_t>(img.size());
        P.ang_n = static_cast<int32_t>(angular.coefficients.size());
        for (double c : angular.coefficients) { const float f = to_float_literal(c); uint32_t u; std::memcpy(&u, &f, 4); img.push_back(u); }
        P.table_words = static_cast<uint32_t>(img.size());
        P.ang_has_min = std::isinf(angular.range_min) ? 0 : 1;
        P.ang_has_max = std::isinf(angular.range_max) ? 0 : 1;
        P.ang_min = to_float_literal(angular.range_min); P.ang_max = to_float_literal(angular.range_max);
        P.ang_underflow = to_float_literal(angular.underflow); P.ang_overflow = to_float_literal(angular.overflow);
        tables_.named["getAngularAcceptance"] = angular.coefficients;
    }
    tables_.named["TABULATOR"] = {n_group_, n_phase_, P.tab_min_inv_groupvel, P.tab_tan_thetac, double(n_bins_)};
    tables_.named["LDS_IMAGE_WORDS"] = {double(P.table_words)};          // (what a workgroup stages; the waves' sample pools come on top)
    tables_.named["TABULATOR_SCALE"] = std::vector<double>(P.tab_scale, P.tab_scale + nd);
    tables_.named["TABULATOR_OFFSET"] = std::vector<double>(P.tab_offset, P.tab_offset + nd);

    for (size_t i = 0; i < streams; ++i)
        if ((x[i] == 0) | ((static_cast<uint32_t>(x[i] >> 32)) >= (a[i] - 1)) | ((static_cast<uint32_t>(x[i])) >= 0xfffffffful))
            throw Error(CLSIMHIP_ERR_ARGUMENT, "invalid MWC state word for stream " + std::to_string(i));

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available (the tabulator has no CPU fallback)");
    if (device_ < 0 || device_ >= count) throw Error(CLSIMHIP_ERR_ARGUMENT, "device ordinal out of range");
    DeviceGuard on_device(device_);
    hip_check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    hip_check(hipEventCreate(&ev_start_), "hipEventCreate");
    hip_check(hipEventCreate(&ev_stop_), "hipEventCreate");
    auto upload = [&](void **dst, const void *src, size_t bytes, const char *what) {
        hip_check(hipMalloc(dst, std::max<size_t>(bytes, 16)), what);
        if (bytes) hip_check(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice), what);
    };
    upload(reinterpret_cast<void **>(&d_tables_), tables_.lds_image.data(), tables_.lds_image.size() * 4, "tables");
    upload(reinterpret_cast<void **>(&d_len_table_), tables_.len_table.data(), tables_.len_table.size() * 4, "length tables");
    upload(reinterpret_cast<void **>(&d_rng_x_), x, streams * 8, "rng x");
    upload(reinterpret_cast<void **>(&d_rng_a_), a, streams * 4, "rng a");
    hip_check(hipMalloc(reinterpret_cast<void **>(&d_bins_), n_device_bins_ * sizeof(double)), "table bins");
    hip_check(hipMemset(d_bins_, 0, n_device_bins_ * sizeof(double)), "table bins");
    if (squared_) {
        hip_check(hipMalloc(reinterpret_cast<void **>(&d_sq_bins_), n_device_bins_ * sizeof(double)), "squared weights");
        hip_check(hipMemset(d_sq_bins_, 0, n_device_bins_ * sizeof(double)), "squared weights");
    }
    hip_check(hipMalloc(reinterpret_cast<void **>(&d_steps_), streams * sizeof(DevStep)), "steps");
    hip_check(hipMalloc(reinterpret_cast<void **>(&d_queue_), static_cast<size_t>(kQueueWords) * kQueueSlots * sizeof(uint32_t)), "step queue");
    hip_check(hipMalloc(reinterpret_cast<void **>(&d_work_), streams * sizeof(WorkRecord)), "work records");
    hip_check(hipHostMalloc(reinterpret_cast<void **>(&h_steps_), streams * sizeof(clsimhip_step), hipHostMallocDefault), "pinned steps");
}

Tabulator::~Tabulator()
{
    (void)hipSetDevice(device_);
    if (stream_) { (void)hipStreamSynchronize(stream_); (void)hipStreamDestroy(stream_); }
    if (ev_start_) (void)hipEventDestroy(ev_start_);
    if (ev_stop_) (void)hipEventDestroy(ev_stop_);
    (void)hipFree(d_tables_); (void)hipFree(d_len_table_); (void)hipFree(d_bins_); (void)hipFree(d_sq_bins_);
    (void)hipFree(d_rng_x_); (void)hipFree(d_rng_a_); (void)hipFree(d_steps_); (void)hipFree(d_queue_); (void)hipFree(d_work_);
    if (h_steps_) (void)hipHostFree(h_steps_);
}

// EnqueueSteps (StepToTableConverter.cxx:272-285) + one pass of FetchSteps (:399-460) without the entry buffers
void Tabulator::enqueue_steps(const clsimhip_step *steps, size_t n, const double ref[7])
{
    if (!steps || !ref) throw Error(CLSIMHIP_ERR_ARGUMENT, "steps / reference particle are (null)");
    if (n == 0) return;
    if (n > streams_) throw Error(CLSIMHIP_ERR_ARGUMENT, "Number of steps is greater than the number of RNG streams!");
    std::lock_guard<std::mutex> lk(mutex_);
    DeviceGuard on_device(device_);
    // the previous bunch still reads the pinned staging buffer and the slice counters
    hip_check(hipStreamSynchronize(stream_), "previous bunch");
    if (pending_event_) {
        float ms = 0.f;
        hip_check(hipEventElapsedTime(&ms, ev_start_, ev_stop_), "event time");
        device_ms_ += ms;
        pending_event_ = false;
    }
    for (size_t i = 0; i < n; ++i) {
        num_photons_ += steps[i].num_photons;
        sum_of_photon_weights_ += static_cast<double>(steps[i].num_photons) * static_cast<double>(steps[i].weight);
    }
    std::memcpy(h_steps_, steps, n * sizeof(clsimhip_step));
    hip_check(hipMemcpyAsync(d_steps_, h_steps_, n * sizeof(DevStep), hipMemcpyHostToDevice, stream_), "upload steps");
    KParams P = tables_.params;
    P.tables = d_tables_;
    P.len_table = d_len_table_;
    P.steps = d_steps_;
    P.n_steps = static_cast<uint32_t>(n);
    P.rng_x = d_rng_x_;
    P.rng_a = d_rng_a_;
    P.queue = d_queue_ + static_cast<size_t>(kQueueWords) * (queue_slot_++ % kQueueSlots);
    hip_check(hipMemsetAsync(P.queue, 0, kQueueWords * sizeof(uint32_t), stream_), "reset step queue");
    P.k_new = 12;
    P.k_search = 1;
    P.slices = 0;
    P.work = d_work_;
    P.tab_bins = d_bins_;
    P.tab_sq_bins = d_sq_bins_;
    {   // I3CLSimReferenceParticle (StepToTableConverter.cxx:64-93)
        const double dx = ref[4], dy = ref[5], dz = ref[6];
        const double perpz = std::hypot(dx, dy);
        double px = 1., py = 0., pz = 0.;
        if (perpz > 0.) {
            // I3Direction(x, y, z) normalises its arguments
            px = -dx * dz / perpz; py = -dy * dz / perpz; pz = perpz;
            const double norm = std::sqrt(px * px + py * py + pz * pz);
            px /= norm; py /= norm; pz /= norm;
        }
        const double v[12] = {ref[0], ref[1], ref[2], ref[3], dx, dy, dz, 0., px, py, pz, 0.};
        for (int k = 0; k < 12; ++k) P.tab_ref[k] = static_cast<float>(v[k]);
    }
    hip_check(hipEventRecord(ev_start_, stream_), "event");
    if (!standard_sampler_) P.tab_std = 0u;          // ("standard_sampler" 0: the generic sampler also for the standard table; tests compare the two)
    KVariant variant = tables_.variant;
    variant.tab_fast = fast_kernels_;
    variant.grid = grid_;
    hip_check(launch_tab_kernel(P, variant, stream_), "tabulation kernel launch");
    hip_check(hipEventRecord(ev_stop_, stream_), "event");
    pending_event_ = true;
    ++launches_;
}

// clsimhip_tabulator_set_tuning (include/clsimhip.h)
void Tabulator::set_tuning(const std::string &key, long long value)
{
    std::lock_guard<std::mutex> lk(mutex_);
    if (key == "fast_kernels" && (value == 0 || value == 1)) fast_kernels_ = (value != 0);
    else if (key == "standard_sampler" && (value == 0 || value == 1)) standard_sampler_ = (value != 0);
    else if (key == "grid" && value >= 0 && value <= (1 << 20)) grid_ = static_cast<int>(value);
    else throw Error(CLSIMHIP_ERR_ARGUMENT, "table maker tuning: no key " + key + " with the value " + std::to_string(value));
}

void Tabulator::finish()
{
    std::lock_guard<std::mutex> lk(mutex_);
    DeviceGuard on_device(device_);
    hip_check(hipStreamSynchronize(stream_), "tabulation kernel");
    if (pending_event_) {
        float ms = 0.f;
        hip_check(hipEventElapsedTime(&ms, ev_start_, ev_stop_), "event time");
        device_ms_ += ms;
        pending_event_ = false;
    }
}

// SphericalAxes / CylindricalAxes::GetBinVolume (Axes.cxx:118-133, 153-164)
double Tabulator::bin_volume(const size_t idxs[3]) const
{
    auto e = [&](size_t k, size_t i) { return axes_[k].bin_edge(static_cast<unsigned>(i)); };
    if (axes_kind_ == CLSIMHIP_AXES_SPHERICAL) {
        const double scalefactor = (axes_[1].max > 180.) ? 1 : 2;
        return ((std::pow(e(0, idxs[0] + 1), 3) - std::pow(e(0, idxs[0]), 3)) / 3.)
            * scalefactor * units::deg * (e(1, idxs[1] + 1) - e(1, idxs[1]))
            * (e(2, idxs[2] + 1) - e(2, idxs[2]));
    }
    return ((std::pow(e(0, idxs[0] + 1), 2) - std::pow(e(0, idxs[0]), 2)) / 2.)
        * 2 * (e(1, idxs[1] + 1) - e(1, idxs[1]))
        * (e(2, idxs[2] + 1) - e(2, idxs[2]));
}

void Tabulator::bin_content_double(double *out, size_t n, bool squared)
{
    if (!out || n != n_bins_) throw Error(CLSIMHIP_ERR_ARGUMENT, "output buffer must hold exactly n_bins values");
    if (squared && !d_sq_bins_) throw Error(CLSIMHIP_ERR_STATE, "squared weights are not recorded");
    finish();
    std::lock_guard<std::mutex> lk(mutex_);
    if (!tiled_) {
        hip_check(hipMemcpy(out, squared ? d_sq_bins_ : d_bins_, n * sizeof(double), hipMemcpyDeviceToHost), "download table");
        return;
    }
    // the device's tiled order -> the reference's (sample_bin in prop_kernel.hip forms the same index)
    std::vector<double> device(n_device_bins_);
    hip_check(hipMemcpy(device.data(), squared ? d_sq_bins_ : d_bins_, n_device_bins_ * sizeof(double), hipMemcpyDeviceToHost), "download table");
    size_t at = 0;
    for (size_t b0 = 0; b0 < shape_[0]; ++b0)
        for (size_t b1 = 0; b1 < shape_[1]; ++b1)
            for (size_t b2 = 0; b2 < shape_[2]; ++b2) {
                const unsigned e0 = tile_bits_[0], e2 = tile_bits_[1], e3 = tile_bits_[2];
                const size_t base = (b0 >> e0) * tile_stride_[0] + b1 * tile_stride_[1] + (b2 >> e2) * tile_stride_[2]
                                    + ((b0 & ((size_t(1) << e0) - 1)) << (e2 + e3)) + ((b2 & ((size_t(1) << e2) - 1)) << e3);
                for (size_t b3 = 0; b3 < shape_[3]; ++b3) out[at++] = device[base + ((b3 >> e3) << 3) + (b3 & ((size_t(1) << e3) - 1))];
            }
}

void Tabulator::bin_content(float *out, size_t n, bool squared, bool normalized)
{
    std::vector<double> sums(n_bins_);
    bin_content_double(sums.data(), n, squared);
    for (size_t i = 0; i < n; ++i) out[i] = static_cast<float>(sums[i]);
    if (!normalized) return;
    // Normalize (StepToTableConverter.cxx:512-543): the first 3 dimensions are spatial
    const size_t nd = axes_.size();
    const size_t spatial_stride = strides_[2];
    for (size_t offset = 0; offset < n_bins_; offset += spatial_stride) {
        size_t idxs[5];
        for (size_t j = 0; j < nd; ++j)
            idxs[j] = static_cast<size_t>(std::min(std::max(static_cast<int>(offset / strides_[j] % shape_[j]) - 1, 0), static_cast<int>(shape_[j]) - 3));
        double norm = bin_volume(idxs) / (step_length_ * reference_area_);
        if (squared) norm *= norm;
        for (size_t i = 0; i < spatial_stride; ++i) out[i + offset] = static_cast<float>(out[i + offset] / norm);
    }
}

void Tabulator::statistics(double out[8])
{
    finish();
    std::lock_guard<std::mutex> lk(mutex_);
    out[0] = static_cast<double>(num_photons_); out[1] = sum_of_photon_weights_; out[2] = n_group_; out[3] = n_phase_;
    out[4] = device_ms_; out[5] = static_cast<double>(launches_); out[6] = static_cast<double>(n_bins_); out[7] = 0.;
}

void Tabulator::get_rng_state(uint64_t *x, size_t count)
{
    if (!x || count > streams_) throw Error(CLSIMHIP_ERR_ARGUMENT, "bad rng state request");
    finish();
    std::lock_guard<std::mutex> lk(mutex_);
    hip_check(hipMemcpy(x, d_rng_x_, count * sizeof(uint64_t), hipMemcpyDeviceToHost), "download rng state");
}

long Tabulator::get_table(const std::string &name, double *out, size_t cap) const
{
    auto it = tables_.named.find(name);
    if (it == tables_.named.end()) throw Error(CLSIMHIP_ERR_ARGUMENT, "no table named " + name);
    const size_t n = it->second.size();
    if (out) std::memcpy(out, it->second.data(), std::min(n, cap) * sizeof(double));
    return static_cast<long>(n);
}

// ---- WriteFITSFile (StepToTableConverter.cxx:545-686) without cfitsio ----------------------------------------------
// A FITS file is a sequence of 2880-byte blocks: header units of 80-character cards ("KEYWORD = value"), data units of
// big-endian numbers.  cfitsio writes the image with the axis counts reversed ("like PyFITS does", :553-561), long
// keyword names through the HIERARCH convention (:621), the squared weights as an IMAGE extension "ERRORS" and one
// double IMAGE extension "EDGESi" per axis.  Same structure, keywords and values here; card comments are omitted.
namespace {

struct FitsWriter {
    FILE *f = nullptr;
    size_t in_block = 0;
    explicit FitsWriter(const std::string &path)
    {
        f = std::fopen(path.c_str(), "wbx");                // fits_create_diskfile refuses to overwrite, so does "x"
        if (!f) throw Error(CLSIMHIP_ERR_IO, "Could not create " + path);
    }
    ~FitsWriter() { if (f) std::fclose(f); }
    void raw(const void *p, size_t n)
    {
        if (std::fwrite(p, 1, n, f) != n) throw Error(CLSIMHIP_ERR_IO, "write failed");
        in_block = (in_block + n) % 2880;
    }
    void pad(char c)
    {
        static const std::string zeros(2880, '\0'), blanks(2880, ' ');
        if (in_block) raw((c == ' ' ? blanks : zeros).data(), 2880 - in_block);
    }
    void card(const std::string &text)
    {
        std::string c = text.substr(0, 80);
        c.resize(80, ' ');
        raw(c.data(), 80);
    }
    void key_logical(const char *k, bool v) { char b[81]; std::snprintf(b, sizeof b, "%-8s= %20s", k, v ? "T" : "F"); card(b); }
    void key_int(const char *k, long long v) { char b[81]; std::snprintf(b, sizeof b, "%-8s= %20lld", k, v); card(b); }
    void key_string(const char *k, const std::string &v) { char b[96]; std::snprintf(b, sizeof b, "%-8s= '%-8s'", k, v.c_str()); card(b); }
    void hierarch_int(const std::string &k, long long v) { char b[128]; std::snprintf(b, sizeof b, "HIERARCH %s = %lld", k.c_str(), v); card(b); }
    void hierarch_double(const std::string &k, double v) { char b[128]; std::snprintf(b, sizeof b, "HIERARCH %s = %.15G", k.c_str(), v); card(b); }
    void end_header() { card("END"); pad(' '); }
    template <class T, class U>
    void data(const std::vector<T> &v)
    {
        static_assert(sizeof(T) == sizeof(U), "size");
        std::vector<unsigned char> out(v.size() * sizeof(T));
        for (size_t i = 0; i < v.size(); ++i) {
            U bits;
            std::memcpy(&bits, &v[i], sizeof(T));
            for (size_t b = 0; b < sizeof(T); ++b) out[i * sizeof(T) + b] = static_cast<unsigned char>(bits >> (8 * (sizeof(T) - 1 - b)));
        }
        raw(out.data(), out.size());
        pad('\0');
    }
};

} // namespace

void Tabulator::write_fits_file(const std::string &path, const std::vector<HeaderEntry> &header)
{
    finish();
    const size_t nd = shape_.size();
    std::vector<float> content(n_bins_);
    bin_content(content.data(), n_bins_, false, true);          // this->Normalize() (:608)
    FitsWriter w(path);
    auto image_axes = [&](int bitpix, const std::vector<size_t> &shape) {
        w.key_int("BITPIX", bitpix);
        w.key_int("NAXIS", static_cast<long long>(shape.size()));
        for (size_t k = 0; k < shape.size(); ++k) {             // reversed: the last axis varies fastest in memory (:553-556)
            char name[16];
            std::snprintf(name, sizeof name, "NAXIS%zu", k + 1);
            w.key_int(name, static_cast<long long>(shape[shape.size() - 1 - k]));
        }
    };
    w.key_logical("SIMPLE", true);
    image_axes(-32, shape_);
    w.key_logical("EXTEND", true);
    {   // header keywords (:613-642): what only the converter knows, then the caller's
        double st[8];
        statistics(st);
        w.hierarch_double("_i3_n_photons", spectral_bias_factor_ * st[1]);      // spectralBiasFactor_ * sumOfPhotonWeights_
        w.hierarch_double("_i3_n_group", n_group_);
        w.hierarch_double("_i3_n_phase", n_phase_);
        for (const HeaderEntry &e : header) {
            if (e.key == "n_photons" || e.key == "n_group" || e.key == "n_phase") continue;      // overwritten by the converter (:614-616)
            if (e.is_int) w.hierarch_int("_i3_" + e.key, e.i);
            else w.hierarch_double("_i3_" + e.key, e.d);
        }
    }
    w.end_header();
    w.data<float, uint32_t>(content);
    if (squared_) {                                             // :647-651
        std::vector<float> sq(n_bins_);
        bin_content(sq.data(), n_bins_, true, true);
        w.key_string("XTENSION", "IMAGE");
        image_axes(-32, shape_);
        w.key_int("PCOUNT", 0);
        w.key_int("GCOUNT", 1);
        w.key_string("EXTNAME", "ERRORS");
        w.end_header();
        w.data<float, uint32_t>(sq);
    }
    for (size_t i = 0; i < nd; ++i) {                           // :656-680
        std::vector<double> edges(axes_[i].n_bins + 1);
        for (unsigned k = 0; k <= axes_[i].n_bins; ++k) edges[k] = axes_[i].bin_edge(k);
        w.key_string("XTENSION", "IMAGE");
        image_axes(-64, std::vector<size_t>(1, edges.size()));
        w.key_int("PCOUNT", 0);
        w.key_int("GCOUNT", 1);
        w.key_string("EXTNAME", "EDGES" + std::to_string(i));
        w.end_header();
        w.data<double, uint64_t>(edges);
    }
}

} // namespace clsimhip
