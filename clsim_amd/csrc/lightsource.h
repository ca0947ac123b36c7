// Particle -> step requests (lightsource.cpp): the front end of I3CLSimLightSourceToStepConverterPPC.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/clsimhip.h"
#include "host_model.h"

namespace clsimhip {

struct ShowerParameters {               // I3SimConstants::ShowerParameters (sim-services; not part of the reference tree)
    double a = 0., b = 0.;              // longitudinal profile: depth = b * Gamma(a) [m]
    double em_scale = 1., em_scale_sigma = 0.;
};
ShowerParameters shower_parameters(int32_t type, double energy_gev, double density_g_cm3);

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
};

// ConverterUtils.cxx:113-214; spectrum == nullptr: delta peak at peak_wavelength
double flasher_correction_factor(const FunctionData *spectrum, double peak_wavelength, const FunctionData &bias, double from_wlen, double to_wlen);
// Flasher.cxx:214-265
void flasher_enqueue(double correction, uint64_t seed, const clsimhip_flasher_pulse *pulses, size_t n, std::vector<clsimhip_flasher_request> &out);

} // namespace clsimhip
