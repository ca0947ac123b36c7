// Photon table maker: the MI355X counterpart of I3CLSimStepToTableConverter
// (private/clsim/tabulator/I3CLSimStepToTableConverter.{h,cxx}, Axes.{h,cxx}, Axis.{h,cxx}).
//
// The reference compiles propKernel with -DTABULATE, runs it as ONE work item on a CPU OpenCL device
// (StepToTableConverter.cxx:259), has every stream write (bin, weight) entries into a buffer and adds them up on the
// host.  Here the same photon loop runs on the whole GPU (prop_kernel<..., TAB=true>) and every path sample is
// added to its bin with a hardware fp64 atomic: no entry buffers, no re-runs of streams that ran out of space.
#pragma once
#include <mutex>

#include "converter.h"

namespace clsimhip {

struct AxisData {                       // clsim::tabulator::LinearAxis / PowerAxis (Axis.h:36-91)
    int kind = CLSIMHIP_AXIS_LINEAR;
    double min = 0, max = 0;
    unsigned n_bins = 0, power = 1;
    double transform(double v) const;
    double inverse(double v) const;
    double bin_edge(unsigned i) const;  // Axis::GetBinEdge
};

struct PolynomialData {                 // I3CLSimFunctionPolynomial
    std::vector<double> coefficients;
    double range_min = -INFINITY, range_max = INFINITY, underflow = NAN, overflow = NAN;
};

class Tabulator {
public:
    Tabulator(int device, int axes_kind, std::vector<AxisData> axes, bool store_squared_weights, const MediumData &medium,
              const FunctionData &wavelength_acceptance, const PolynomialData &angular_acceptance, double reference_area,
              double step_length, const uint64_t *x, const uint32_t *a, size_t streams);
    ~Tabulator();

    void enqueue_steps(const clsimhip_step *steps, size_t n, const double reference[7]);
    void finish();
    void set_tuning(const std::string &key, long long value);
    size_t n_bins() const { return n_bins_; }
    const std::vector<size_t> &shape() const { return shape_; }
    const std::vector<AxisData> &axes() const { return axes_; }
    // binContent_ / squaredWeights_ as floats; normalized = after Normalize() (StepToTableConverter.cxx:512-543)
    void bin_content(float *out, size_t n, bool squared, bool normalized);
    void bin_content_double(double *out, size_t n, bool squared);
    double bin_volume(const size_t idxs[3]) const;
    void statistics(double out[8]);
    // WriteFITSFile (StepToTableConverter.cxx:595-686): normalised bin content as the primary image, the header keywords
    // (n_photons, n_group, n_phase and the caller's), squared weights as "ERRORS", one "EDGESi" extension per axis
    struct HeaderEntry { std::string key; bool is_int; long long i; double d; };
    void write_fits_file(const std::string &path, const std::vector<HeaderEntry> &header);
    double spectral_bias_factor() const { return spectral_bias_factor_; }
    void get_rng_state(uint64_t *x, size_t count);
    long get_table(const std::string &name, double *out, size_t cap) const;

private:
    void hip_check(hipError_t e, const char *what) const;
    int device_;
    int axes_kind_;
    std::vector<AxisData> axes_;
    std::vector<size_t> shape_, strides_;
    size_t n_bins_ = 0;
    bool tiled_ = false;                     // the device keeps the bins in tiles of eight of axes 0, 2, 3 (tabulator.cpp)
    size_t n_device_bins_ = 0, tile_stride_[3] = {0, 0, 0};
    bool fast_kernels_ = false;              // "fast_kernels" (set_tuning): the FAST instantiation, measured slower (prop_kernel.hip: launch_tab_kernel)
    int grid_ = 0;                           // "grid": workgroups of the launch, 0 = automatic
    bool standard_sampler_ = true;           // "standard_sampler": 0 keeps the generic sampler also for the standard table (KParams::tab_std)
    unsigned tile_bits_[3] = {2, 1, 0};      // bins per sector along axes 0, 2, 3 as powers of two (sum 3): 4 x 2 x 1 (tabulator.cpp)
    bool squared_;
    double reference_area_, step_length_;
    double n_group_ = 0, n_phase_ = 0;
    double spectral_bias_factor_ = 1.;  // spectralBiasFactor_ (StepToTableConverter.cxx:142-152)
    CompiledTables tables_;
    size_t streams_;
    uint32_t *d_tables_ = nullptr;
    float *d_len_table_ = nullptr;
    double *d_bins_ = nullptr, *d_sq_bins_ = nullptr;
    uint64_t *d_rng_x_ = nullptr;
    uint32_t *d_rng_a_ = nullptr;
    DevStep *d_steps_ = nullptr;
    uint32_t *d_queue_ = nullptr;
    WorkRecord *d_work_ = nullptr;
    clsimhip_step *h_steps_ = nullptr;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev_start_ = nullptr, ev_stop_ = nullptr;
    uint32_t queue_slot_ = 0;
    std::mutex mutex_;
    uint64_t num_photons_ = 0, launches_ = 0;
    double sum_of_photon_weights_ = 0, device_ms_ = 0;
    bool pending_event_ = false;
};

} // namespace clsimhip
