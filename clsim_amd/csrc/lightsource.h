// Particle -> step requests (lightsource.cpp): the front end of I3CLSimLightSourceToStepConverterPPC.
#pragma once
#include <cstdint>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/clsimhip.h"
#include "host_model.h"

namespace clsimhip {

struct ShowerParameters {               // I3SimConstants::ShowerParameters (sim-services; not part of the reference tree)
    double a = 0., b = 0.;              // longitudinal profile: depth = b * Gamma(a) [m]
    double em_scale = 1., em_scale_sigma = 0.;
};
ShowerParameters shower_parameters(int32_t type, double energy_gev, double density_g_cm3);

// Which time an identifier is seen by a converter.  The random stream of a light source is a function of (seed, identifier):
// results then do not depend on the order light sources of DIFFERENT identifiers arrive in.  Identifiers that come back
// (restarting per event or frame, wrapping around) must not replay the same fluctuations, as the reference's one
// sequential I3RandomService never does: the stream also depends on how often the identifier has been seen before.  The
// first occurrence leaves the seed as it is.  (Bounded: after 2^20 distinct identifiers the table starts over in a new epoch.)
class OccurrenceCounter {
public:
    uint64_t mix(uint32_t identifier)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (seen_.size() >= (1u << 20)) { seen_.clear(); ++epoch_; }
        const uint64_t nth = seen_[identifier]++;
        return 0xA24BAED4963EE407ull * nth + 0x9FB21C651E98DF25ull * epoch_;
    }
private:
    std::mutex m_;
    std::unordered_map<uint32_t, uint32_t> seen_;
    uint64_t epoch_ = 0;
};

// ConverterUtils.cxx:44-105
double photons_per_meter(const MediumData &medium, const FunctionData &bias, double from_wlen, double to_wlen);

struct PPCConfig {                      // constructor arguments of I3CLSimLightSourceToStepConverterPPC (:51-70) + what it gets from setters
    uint32_t photons_per_step = 200, high_photons_per_step = 2000;
    double use_high_photons_per_step_from = 1e9;
    bool use_cascade_extension = true;
    double density = 0.9216;            // I3CLSimMediumProperties::GetMediumDensity() [g/cm3] (MakeIceCubeMediumProperties.py)
    uint64_t seed = 0;
};

class PPCConverter {
public:
    PPCConverter(const MediumData &medium, const FunctionData &bias, const PPCConfig &config);
    double mean_photons_per_meter(int layer) const;
    void enqueue(const clsimhip_particle &particle, std::vector<clsimhip_step_request> &out) const;

private:
    PPCConfig config_;
    double layers_z_start_ = 0., layers_height_ = 1.;
    std::vector<double> photons_per_meter_;
    mutable OccurrenceCounter occurrences_;
};

// ConverterUtils.cxx:113-214; spectrum == nullptr: delta peak at peak_wavelength
double flasher_correction_factor(const FunctionData *spectrum, double peak_wavelength, const FunctionData &bias, double from_wlen, double to_wlen);
// Flasher.cxx:214-265
void flasher_enqueue(double correction, uint64_t seed, const clsimhip_flasher_pulse *pulses, size_t n, std::vector<clsimhip_flasher_request> &out);

} // namespace clsimhip
