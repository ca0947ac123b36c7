// C++ adapter: the reference's converter interface implemented on the C ABI.
//
// `I3CLSimStepToPhotonConverterHIP` has the member functions of
// I3CLSimStepToPhotonConverter (public/clsim/I3CLSimStepToPhotonConverter.h:67-192)
// and the concrete setters the canonical caller uses
// (public/clsim/I3CLSimStepToPhotonConverterOpenCL.h:78-258,
// private/clsim/I3CLSimModuleHelper.cxx:303-372), with the same names, argument
// meaning and error behaviour (I3CLSimStepToPhotonConverter_exception).
//
// Two builds of the same class:
//  * -DCLSIMHIP_WITH_ICETRAY (inside IceTray): derives from the REAL interface -- the four configuration setters
//    take I3CLSimRandomValueConstPtr / I3CLSimFunctionConstPtr / I3CLSimMediumPropertiesConstPtr /
//    I3CLSimSimpleGeometryConstPtr exactly as public/clsim/I3CLSimStepToPhotonConverter.h:91-124 declares them and
//    translate through I3CLSimStepToPhotonConverterHIPGlue.h; the constructor takes the I3RandomService that seeds the
//    streams (OpenCL.cxx:68-69), SetDevice takes the HIP device ordinal (OpenCL.h:102 takes an I3CLSimOpenCLDevice).
//    This repository compiles and runs that build against tests/stubs/ (tests/test_icetray_adapter.py).
//  * stand-alone (no IceTray headers at all): derives from the minimal interface below, which has the same virtual
//    functions in the same order, and takes the C descriptors directly.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/clsimhip.h"

#ifdef CLSIMHIP_WITH_ICETRAY
#include <clsim/I3CLSimStepToPhotonConverter.h>
#include <phys-services/I3RandomService.h>
#include "I3CLSimStepToPhotonConverterHIPGlue.h"
#else
// ---- stand-alone counterparts of the reference types ----
typedef clsimhip_step I3CLSimStep;                       // public/clsim/I3CLSimStep.h (48 B blob)
typedef clsimhip_photon I3CLSimPhoton;                   // public/clsim/I3CLSimPhoton.h (80 B blob)
typedef std::vector<I3CLSimStep> I3CLSimStepSeries;
typedef std::vector<I3CLSimPhoton> I3CLSimPhotonSeries;
typedef std::shared_ptr<const I3CLSimStepSeries> I3CLSimStepSeriesConstPtr;
typedef std::shared_ptr<I3CLSimPhotonSeries> I3CLSimPhotonSeriesPtr;
struct I3CLSimPhotonHistory {                            // public/clsim/I3CLSimPhotonHistory.h:41-70 (accessors used here)
    std::size_t size() const { return posX_.size(); }
    float GetX(std::size_t i) const { return posX_[i]; }
    float GetY(std::size_t i) const { return posY_[i]; }
    float GetZ(std::size_t i) const { return posZ_[i]; }
    float GetDistanceInAbsorptionLengths(std::size_t i) const { return distanceInAbsorptionLengths_[i]; }
    void push_back(float x, float y, float z, float abslens)
    {
        posX_.push_back(x); posY_.push_back(y); posZ_.push_back(z); distanceInAbsorptionLengths_.push_back(abslens);
    }
private:
    std::vector<float> posX_, posY_, posZ_, distanceInAbsorptionLengths_;
};
typedef std::vector<I3CLSimPhotonHistory> I3CLSimPhotonHistorySeries;
typedef std::shared_ptr<I3CLSimPhotonHistorySeries> I3CLSimPhotonHistorySeriesPtr;

class I3CLSimStepToPhotonConverter_exception : public std::runtime_error {
public:
    explicit I3CLSimStepToPhotonConverter_exception(const std::string &msg) : std::runtime_error(msg) {}
};

struct I3CLSimStepToPhotonConverter {
    struct ConversionResult_t {
        ConversionResult_t() : identifier(0) {}
        uint32_t identifier;
        I3CLSimPhotonSeriesPtr photons;
        I3CLSimPhotonHistorySeriesPtr photonHistories;  // null unless PhotonHistoryEntries > 0
    };
    virtual ~I3CLSimStepToPhotonConverter() {}
    virtual void SetWlenGenerators(const std::vector<clsimhip_random_value> &wlenGenerators) = 0;
    virtual void SetWlenBias(const clsimhip_function &wlenBias) = 0;
    virtual void SetMediumProperties(const clsimhip_medium *mediumProperties) = 0;
    virtual void SetGeometry(const std::vector<int32_t> &stringIDs, const std::vector<uint32_t> &domIDs,
                             const std::vector<double> &x, const std::vector<double> &y, const std::vector<double> &z,
                             const std::vector<std::string> &subdetectors, double omRadius) = 0;
    virtual void Initialize() = 0;
    virtual bool IsInitialized() const = 0;
    virtual void EnqueueSteps(I3CLSimStepSeriesConstPtr steps, uint32_t identifier) = 0;
    virtual std::size_t GetWorkgroupSize() const = 0;
    virtual std::size_t GetMaxNumWorkitems() const = 0;
    virtual std::size_t QueueSize() const = 0;
    virtual bool MorePhotonsAvailable() const = 0;
    virtual ConversionResult_t GetConversionResult() = 0;
    virtual std::map<std::string, double> GetStatistics() const { return std::map<std::string, double>(); }
};
#endif

static_assert(sizeof(I3CLSimStep) == sizeof(clsimhip_step), "I3CLSimStep must be the 48-byte record of the C ABI");
static_assert(sizeof(I3CLSimPhoton) == sizeof(clsimhip_photon), "I3CLSimPhoton must be the 80-byte record of the C ABI");

class I3CLSimStepToPhotonConverterHIP : public I3CLSimStepToPhotonConverter {
public:
#ifdef CLSIMHIP_WITH_ICETRAY
    // I3CLSimStepToPhotonConverterOpenCL(I3RandomServicePtr, bool useNativeMath) (OpenCL.cxx:68-69); there is one math
    // library here, so the second argument has no counterpart
    explicit I3CLSimStepToPhotonConverterHIP(I3RandomServicePtr randomService, int device = 0) : handle_(nullptr), seed_(12345), randomService_(randomService)
    {
        if (clsimhip_create(device, &handle_) != CLSIMHIP_OK) throw I3CLSimStepToPhotonConverter_exception(clsimhip_last_error(nullptr));
        owner_.reset(handle_, [](clsimhip_converter *h) { clsimhip_destroy(h); });
    }
#else
    explicit I3CLSimStepToPhotonConverterHIP(int device = 0, uint64_t seed = 12345) : handle_(nullptr), seed_(seed)
    {
        if (clsimhip_create(device, &handle_) != CLSIMHIP_OK) throw I3CLSimStepToPhotonConverter_exception(clsimhip_last_error(nullptr));
        owner_.reset(handle_, [](clsimhip_converter *h) { clsimhip_destroy(h); });
    }
#endif
    // the reference interface declares no virtual destructor (I3CLSimStepToPhotonConverter.h:88 has it commented out):
    // converters live in shared_ptrs made from the concrete type, as I3CLSimStepToPhotonConverterOpenCLPtr does
    // (the library's converter goes when this object AND every ConversionResultView handed out by GetConversionResultInPlace() are
    // gone: a view keeps the converter it must return its buffer to alive -- ADVICE r5)
    virtual ~I3CLSimStepToPhotonConverterHIP() {}
    I3CLSimStepToPhotonConverterHIP(const I3CLSimStepToPhotonConverterHIP &) = delete;
    I3CLSimStepToPhotonConverterHIP &operator=(const I3CLSimStepToPhotonConverterHIP &) = delete;

    // ---- interface: configuration ----
#ifdef CLSIMHIP_WITH_ICETRAY
    void SetWlenGenerators(const std::vector<I3CLSimRandomValueConstPtr> &wlenGenerators) override
    {
        std::vector<clsimhip_glue::RandomValueHolder> held(wlenGenerators.size());
        std::vector<clsimhip_random_value> g(wlenGenerators.size());
        for (std::size_t i = 0; i < wlenGenerators.size(); ++i) {
            if (!wlenGenerators[i]) throw I3CLSimStepToPhotonConverter_exception("wavelength generator is (null)!");
            held[i] = clsimhip_glue::MakeHIPWlenGenerator(*wlenGenerators[i], i);
            g[i] = held[i].r;
        }
        check(clsimhip_set_wlen_generators(handle_, g.data(), g.size()));
    }
    void SetWlenBias(I3CLSimFunctionConstPtr wlenBias) override
    {
        if (!wlenBias) throw I3CLSimStepToPhotonConverter_exception("wavelength bias is (null)!");
        const clsimhip_glue::FunctionHolder h = clsimhip_glue::MakeHIPFunction(*wlenBias, "the wavelength bias");
        check(clsimhip_set_wlen_bias(handle_, &h.f));
    }
    void SetMediumProperties(I3CLSimMediumPropertiesConstPtr mediumProperties) override
    {
        if (!mediumProperties) throw I3CLSimStepToPhotonConverter_exception("medium properties are (null)!");
        clsimhip_glue::MediumHolder h;
        clsimhip_glue::MakeHIPMedium(*mediumProperties, h);
        check(clsimhip_set_medium_properties(handle_, h.m));
    }
    // the seven vectors the reference hands to its geometry code generator (GeometrySource.cxx:88-99)
    void SetGeometry(I3CLSimSimpleGeometryConstPtr geometry) override
    {
        if (!geometry) throw I3CLSimStepToPhotonConverter_exception("geometry is (null)!");
        const std::vector<std::string> &sub = geometry->GetSubdetectorVector();
        std::vector<const char *> names;
        for (const std::string &s : sub) names.push_back(s.c_str());
        check(clsimhip_set_geometry(handle_, geometry->size(), geometry->GetStringIDVector().data(), geometry->GetDomIDVector().data(),
                                    geometry->GetPosXVector().data(), geometry->GetPosYVector().data(), geometry->GetPosZVector().data(),
                                    names.data(), geometry->GetOMRadius()));
    }
    // streams seeded from the random service like init_MWC_RNG does (private/opencl/mwcrng_init.h:104-112); the
    // multipliers are the library's (the same safeprimes the reference reads from its file)
    void Initialize() override
    {
        if (!randomService_) { check(clsimhip_initialize(handle_, seed_)); return; }
        if (IsInitialized()) throw I3CLSimStepToPhotonConverter_exception("I3CLSimStepToPhotonConverterHIP already initialized!");
        check(clsimhip_compile(handle_));
        std::size_t n = 0;
        check(clsimhip_get_max_num_workitems(handle_, &n));
        if (n == 0) { n = 1048576; check(clsimhip_set_max_num_workitems(handle_, n)); }
        std::vector<uint32_t> a(n);
        std::vector<uint64_t> x(n, 0);
        check(clsimhip_mwc_multipliers(a.data(), n));
        for (std::size_t i = 0; i < n; ++i) {
            while ((x[i] == 0) | ((static_cast<uint32_t>(x[i] >> 32)) >= (a[i] - 1)) | ((static_cast<uint32_t>(x[i])) >= 0xfffffffful)) {
                x[i] = static_cast<uint32_t>(randomService_->Integer(0xffffffff));
                x[i] = x[i] << 32;
                x[i] += static_cast<uint32_t>(randomService_->Integer(0xffffffff));
            }
        }
        check(clsimhip_initialize_with_streams(handle_, x.data(), a.data(), n));
    }
#else
    void SetWlenGenerators(const std::vector<clsimhip_random_value> &g) override { check(clsimhip_set_wlen_generators(handle_, g.data(), g.size())); }
    void SetWlenBias(const clsimhip_function &b) override { check(clsimhip_set_wlen_bias(handle_, &b)); }
    void SetMediumProperties(const clsimhip_medium *m) override { check(clsimhip_set_medium_properties(handle_, m)); }
    void SetGeometry(const std::vector<int32_t> &stringIDs, const std::vector<uint32_t> &domIDs, const std::vector<double> &x,
                     const std::vector<double> &y, const std::vector<double> &z, const std::vector<std::string> &subdetectors,
                     double omRadius) override
    {
        std::vector<const char *> names;
        for (const std::string &s : subdetectors) names.push_back(s.c_str());
        check(clsimhip_set_geometry(handle_, stringIDs.size(), stringIDs.data(), domIDs.data(), x.data(), y.data(), z.data(),
                                    names.data(), omRadius));
    }
    void Initialize() override { check(clsimhip_initialize(handle_, seed_)); }
#endif
    void InitializeWithStreams(const std::vector<uint64_t> &x, const std::vector<uint32_t> &a) { check(clsimhip_initialize_with_streams(handle_, x.data(), a.data(), x.size())); }

    // ---- interface: steady state ----
    bool IsInitialized() const override { return clsimhip_is_initialized(handle_) != 0; }
    void EnqueueSteps(I3CLSimStepSeriesConstPtr steps, uint32_t identifier) override
    {
        if (!steps) {   // the null check comes after the initialisation check in the reference (OpenCL.cxx:1527-1531)
            if (!IsInitialized()) throw I3CLSimStepToPhotonConverter_exception("I3CLSimStepToPhotonConverterHIP is not initialized!");
            throw I3CLSimStepToPhotonConverter_exception("Steps pointer is (null)!");
        }
        check(clsimhip_enqueue_steps(handle_, reinterpret_cast<const clsimhip_step *>(steps->data()), steps->size(), identifier));
    }
    std::size_t GetWorkgroupSize() const override { size_t v = 0; check(clsimhip_get_workgroup_size(handle_, &v)); return v; }
    std::size_t GetMaxNumWorkitems() const override { size_t v = 0; check(clsimhip_get_max_num_workitems(handle_, &v)); return v; }
    std::size_t QueueSize() const override { size_t v = 0; check(clsimhip_queue_size(handle_, &v)); return v; }
    bool MorePhotonsAvailable() const override { int v = 0; check(clsimhip_more_photons_available(handle_, &v)); return v != 0; }
    ConversionResult_t GetConversionResult() override
    {
        ConversionResult_t r;
        const clsimhip_photon *p = nullptr;
        size_t n = 0;
        check(clsimhip_get_conversion_result(handle_, &r.identifier, &p, &n));
        r.photons = I3CLSimPhotonSeriesPtr(new I3CLSimPhotonSeries(n));
        if (n) {
            std::memcpy(static_cast<void *>(r.photons->data()), p, n * sizeof(clsimhip_photon));
            // I3CLSimPhotonHistory (public/clsim/I3CLSimPhotonHistory.h): per photon the recorded scatter points
            const float *h = nullptr;
            uint32_t entries = 0;
            check(clsimhip_get_result_histories(handle_, p, &h, &entries));
            if (h && entries) {
                r.photonHistories = I3CLSimPhotonHistorySeriesPtr(new I3CLSimPhotonHistorySeries(n));
                for (size_t i = 0; i < n; ++i) {
                    const uint32_t k = p[i].num_scatters < entries ? p[i].num_scatters : entries;
                    for (uint32_t j = 0; j < k; ++j) {
                        const float *e = h + (i * entries + j) * 4;
                        (*r.photonHistories)[i].push_back(e[0], e[1], e[2], e[3]);
                    }
                }
            }
            check(clsimhip_release_result(handle_, p));
        }
        return r;
    }
    // Extension (no reference counterpart): the result where the library left it -- the page-locked buffer the device's records were
    // downloaded into -- for a caller that consumes the records in place (a server that serialises them, a hit maker that walks
    // them once): no I3CLSimPhotonSeries is allocated, nothing is copied.  `photons` stays valid while `hold` (or a copy of it) lives
    // -- also beyond this adapter's own lifetime: `hold` shares ownership of the library's converter, which is destroyed (and its
    // page-locked buffers freed) only after the adapter and the last view are gone.  The buffer goes back to the converter's pool
    // when the last copy is dropped.  Photon histories, if recorded, are not part
    // of the view: use GetConversionResult().  GetConversionResult() copies because its interface type owns a std::vector.
    struct ConversionResultView {
        uint32_t identifier = 0;
        const I3CLSimPhoton *photons = nullptr;
        std::size_t size = 0;
        std::shared_ptr<const void> hold;
        const I3CLSimPhoton *begin() const { return photons; }
        const I3CLSimPhoton *end() const { return photons + size; }
    };
    ConversionResultView GetConversionResultInPlace()
    {
        ConversionResultView v;
        const clsimhip_photon *p = nullptr;
        check(clsimhip_get_conversion_result(handle_, &v.identifier, &p, &v.size));
        v.photons = reinterpret_cast<const I3CLSimPhoton *>(p);
        if (p) {
            std::shared_ptr<clsimhip_converter> keep = owner_;
            v.hold = std::shared_ptr<const void>(static_cast<const void *>(p), [keep](const void *q) { (void)clsimhip_release_result(keep.get(), static_cast<const clsimhip_photon *>(q)); });
        }
        return v;
    }
    std::map<std::string, double> GetStatistics() const override
    {
        double v[8];
        check(clsimhip_get_statistics(handle_, v));
        static const char *keys[8] = {"TotalDeviceTime", "TotalHostTime", "NumKernelCalls", "TotalNumPhotonsGenerated",
                                      "TotalNumPhotonsAtDOMs", "AverageDeviceTimePerPhoton", "AverageHostTimePerPhoton", "DeviceUtilization"};
        std::map<std::string, double> m;
        for (int i = 0; i < 8; ++i) m[keys[i]] = v[i];
        return m;
    }

    // the statistics accessors I3CLSimModule::Finish reads from the concrete class (OpenCL.h:377-381, I3CLSimModule.cxx:1626-1634):
    // times in nanoseconds
    double GetTotalDeviceTime() const { return stat(0); }
    double GetTotalHostTime() const { return stat(1); }
    uint64_t GetNumKernelCalls() const { return static_cast<uint64_t>(stat(2)); }
    uint64_t GetTotalNumPhotonsGenerated() const { return static_cast<uint64_t>(stat(3)); }
    uint64_t GetTotalNumPhotonsAtDOMs() const { return static_cast<uint64_t>(stat(4)); }

    // ---- tuning (no reference counterpart; keys and ranges: include/clsimhip.h, clsimhip_set_tuning).  No result depends on it; the library
    // reads none of it from the environment ----
    void SetTuning(const std::string &key, long long value) { check(clsimhip_set_tuning(handle_, key.c_str(), value)); }
    long long GetTuning(const std::string &key) const
    {
        long long v = 0;
        check(clsimhip_get_tuning(handle_, key.c_str(), &v));
        return v;
    }

    // ---- concrete setters and getters of the OpenCL converter (OpenCL.h:78-258) ----
    bool GetEnableDoubleBuffering() const { return option(CLSIMHIP_OPTION_ENABLE_DOUBLE_BUFFERING) != 0.; }
    bool GetDoublePrecision() const { return option(CLSIMHIP_OPTION_DOUBLE_PRECISION) != 0.; }
    bool GetStopDetectedPhotons() const { return option(CLSIMHIP_OPTION_STOP_DETECTED_PHOTONS) != 0.; }
    bool GetSaveAllPhotons() const { return option(CLSIMHIP_OPTION_SAVE_ALL_PHOTONS) != 0.; }
    double GetSaveAllPhotonsPrescale() const { return option(CLSIMHIP_OPTION_SAVE_ALL_PHOTONS_PRESCALE); }
    double GetFixedNumberOfAbsorptionLengths() const { return option(CLSIMHIP_OPTION_FIXED_NUMBER_OF_ABSORPTION_LENGTHS); }
    double GetDOMPancakeFactor() const { return option(CLSIMHIP_OPTION_DOM_PANCAKE_FACTOR); }
    uint32_t GetPhotonHistoryEntries() const { return static_cast<uint32_t>(option(CLSIMHIP_OPTION_PHOTON_HISTORY_ENTRIES)); }
    void SetDevice(int hipDeviceOrdinal) { check(clsimhip_set_device(handle_, hipDeviceOrdinal)); }
    void SetEnableDoubleBuffering(bool v) { check(clsimhip_set_enable_double_buffering(handle_, v)); }
    void SetDoublePrecision(bool v) { check(clsimhip_set_double_precision(handle_, v)); }
    void SetStopDetectedPhotons(bool v) { check(clsimhip_set_stop_detected_photons(handle_, v)); }
    void SetSaveAllPhotons(bool v) { check(clsimhip_set_save_all_photons(handle_, v)); }
    void SetSaveAllPhotonsPrescale(double v) { check(clsimhip_set_save_all_photons_prescale(handle_, v)); }
    void SetFixedNumberOfAbsorptionLengths(double v) { check(clsimhip_set_fixed_number_of_absorption_lengths(handle_, v)); }
    void SetDOMPancakeFactor(double v) { check(clsimhip_set_dom_pancake_factor(handle_, v)); }
    void SetPhotonHistoryEntries(uint32_t v) { check(clsimhip_set_photon_history_entries(handle_, v)); }
    void SetWorkgroupSize(std::size_t v) { check(clsimhip_set_workgroup_size(handle_, v)); }
    void SetMaxNumWorkitems(std::size_t v) { check(clsimhip_set_max_num_workitems(handle_, v)); }
    void Compile() { check(clsimhip_compile(handle_)); }
    std::size_t GetMaxWorkgroupSize() const { size_t v = 0; check(clsimhip_get_max_workgroup_size(handle_, &v)); return v; }
    clsimhip_converter *Handle() { return handle_; }

private:
    void check(int rc) const
    {
        if (rc == CLSIMHIP_OK) return;
#ifdef CLSIMHIP_WITH_ICETRAY
        // a device error is where the reference's worker log_fatal()s (OpenCL.cxx:768-774)
        if (rc == CLSIMHIP_ERR_DEVICE) log_fatal("%s", clsimhip_last_error(handle_));
#endif
        throw I3CLSimStepToPhotonConverter_exception(clsimhip_last_error(handle_));
    }
    double stat(int i) const { double v[8]; check(clsimhip_get_statistics(handle_, v)); return v[i]; }
    double option(int which) const { double v = 0.; check(clsimhip_get_option(handle_, which, &v)); return v; }
    clsimhip_converter *handle_;
    std::shared_ptr<clsimhip_converter> owner_;      // owns handle_ (deleter: clsimhip_destroy); shared with the in-place result views
    uint64_t seed_;
#ifdef CLSIMHIP_WITH_ICETRAY
    I3RandomServicePtr randomService_;
#endif
};

#ifdef CLSIMHIP_WITH_ICETRAY
typedef boost::shared_ptr<I3CLSimStepToPhotonConverterHIP> I3CLSimStepToPhotonConverterHIPPtr;

namespace I3CLSimModuleHelper {
// initializeOpenCL (private/clsim/I3CLSimModuleHelper.cxx:303-372) with a HIP device ordinal in the place of the
// I3CLSimOpenCLDevice (its GetApproximateNumberOfWorkItems() becomes an argument): the same calls in the same order
inline I3CLSimStepToPhotonConverterHIPPtr initializeHIP(int device, uint32_t approximateNumberOfWorkItems, I3RandomServicePtr rng,
                                                        I3CLSimSimpleGeometryConstPtr geometry, I3CLSimMediumPropertiesConstPtr medium,
                                                        I3CLSimFunctionConstPtr wavelengthGenerationBias,
                                                        const std::vector<I3CLSimRandomValueConstPtr> &wavelengthGenerators,
                                                        bool enableDoubleBuffering, bool doublePrecision, bool stopDetectedPhotons,
                                                        bool saveAllPhotons, double saveAllPhotonsPrescale, double fixedNumberOfAbsorptionLengths,
                                                        double pancakeFactor, uint32_t photonHistoryEntries, uint32_t limitWorkgroupSize)
{
    I3CLSimStepToPhotonConverterHIPPtr conv(new I3CLSimStepToPhotonConverterHIP(rng));
    conv->SetDevice(device);
    conv->SetWlenGenerators(wavelengthGenerators);
    conv->SetWlenBias(wavelengthGenerationBias);
    conv->SetMediumProperties(medium);
    conv->SetGeometry(geometry);
    conv->SetEnableDoubleBuffering(enableDoubleBuffering);
    conv->SetDoublePrecision(doublePrecision);
    conv->SetStopDetectedPhotons(stopDetectedPhotons);
    conv->SetSaveAllPhotons(saveAllPhotons);
    conv->SetSaveAllPhotonsPrescale(saveAllPhotonsPrescale);
    conv->SetFixedNumberOfAbsorptionLengths(fixedNumberOfAbsorptionLengths);
    conv->SetDOMPancakeFactor(pancakeFactor);
    conv->SetPhotonHistoryEntries(photonHistoryEntries);
    conv->Compile();
    std::size_t maxWorkgroupSize = conv->GetMaxWorkgroupSize();
    if (limitWorkgroupSize != 0 && limitWorkgroupSize < maxWorkgroupSize) maxWorkgroupSize = limitWorkgroupSize;
    conv->SetWorkgroupSize(maxWorkgroupSize);
    const std::size_t workgroupSize = conv->GetWorkgroupSize();
    std::size_t maxNumWorkitems = (static_cast<std::size_t>(approximateNumberOfWorkItems) / workgroupSize) * workgroupSize;
    if (maxNumWorkitems == 0) maxNumWorkitems = workgroupSize;
    conv->SetMaxNumWorkitems(maxNumWorkitems);
    conv->Initialize();
    return conv;
}
}
#endif
