// Host side of the flasher step producer: bunch plan of
// I3CLSimLightSourceToStepConverterFlasher::MakeSteps (private/clsim/I3CLSimLightSourceToStepConverterFlasher.cxx:329-440)
// and the time delay distribution python/I3CLSimRandomValueIceCubeFlasherTimeProfile.py builds per pulse width.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/clsimhip.h"

namespace clsimhip {

constexpr int kFlasherProfilePoints = 240;      // numpy.linspace(0, 120, 240, endpoint=False): 0.5 ns spacing

struct FlasherPlanEntry {       // one per request, uploaded to the device
    uint64_t first_out;         // index of the request's first output step
    uint64_t n_real;            // steps that carry photons (all photons_per_step, the last one last_real)
    uint32_t last_real;
    uint32_t profile;           // time profile table of the request's pulse width
};

// the LED pulse shape sampled at 0.5 ns (I3CLSimRandomValueIceCubeFlasherTimeProfile.py:118-155, _the_pulse(x, 2*width/ns))
std::vector<double> flasher_time_profile(double width_ns);
// InterpolatedDistribution.cxx:134-175 (InitTables): normalised density and cumulative values as float literals
void interpolated_distribution_tables(double spacing, const std::vector<double> &y, std::vector<float> &density, std::vector<float> &cumulative);
// MakeSteps for every request; returns the number of output steps (real + dummy)
uint64_t plan_flasher_steps(const clsimhip_flasher_config &cfg, const clsimhip_flasher_request *requests, size_t n,
                            std::vector<FlasherPlanEntry> &plan, std::vector<double> &widths);

} // namespace clsimhip
