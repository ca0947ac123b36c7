#include "flasher.h"

#include <algorithm>
#include <cmath>

#include "host_model.h"

namespace clsimhip {
namespace {

// LED output time profile for FB_WIDTH = 15 (I3CLSimRandomValueIceCubeFlasherTimeProfile.py:52-88, measured data from
// the IceCube wiki page the reference cites): 51 samples at 1 ns, before the offset/scale adjustment of :90
const double kPulseWidth15[51] = {
    1.18e-03, 2.769e-02, 1.2517e-01, 2.1484e-01, 3.2089e-01, 4.3239e-01, 4.6437e-01, 5.0023e-01, 4.3161e-01, 3.1621e-01, 2.2965e-01,
    1.3764e-01, 8.774e-02, 7.214e-02, 5.966e-02, 4.797e-02, 4.095e-02, 2.925e-02, 3.081e-02, 2.847e-02, 2.613e-02, 1.834e-02,
    1.834e-02, 1.99e-02, 1.288e-02, 1.288e-02, 1.288e-02, 1.6e-02, 1.444e-02, 1.678e-02, 7.42e-03, 6.64e-03, 9.76e-03, 1.132e-02,
    7.42e-03, 9.76e-03, 4.3e-03, 5.86e-03, 7.42e-03, 4.3e-03, 8.2e-03, 5.86e-03, 3.52e-03, 1.96e-03, 2.74e-03, 4.3e-03, 5.08e-03,
    2.74e-03, 3.52e-03, 4.3e-03, 2.74e-03};

// scipy interp1d(kind='linear', bounds_error=False, fill_value=0.) over x = 0..50 of the adjusted table (:90-91)
double pulse_narrow(double x)
{
    if (!(x >= 0.) || x > 50.) return 0.;
    auto value = [](int i) { return (kPulseWidth15[i] - 0.00118) / 0.49905; };
    if (x == 50.) return value(50);
    const int i = static_cast<int>(std::floor(x));
    const double lo = value(i), hi = value(i + 1);
    const double slope = (hi - lo) / ((i + 1.0) - i);
    return slope * (x - i) + lo;
}
double rising_edge(double x, double width)                         // :93-101
{
    const double template_width = 7.;
    double scaled = template_width * x / width;
    if (scaled > template_width) scaled = template_width;
    if (scaled < 0.) scaled = 0.;
    return pulse_narrow(scaled);
}
double falling_edge(double x)                                      // :103-110
{
    const double template_start = 7.;
    double scaled = x + template_start;
    if (scaled < template_start) scaled = template_start;
    return pulse_narrow(scaled);
}
} // namespace

std::vector<double> flasher_time_profile(double width_ns)
{
    const double fb_width = width_ns * 2.;                          // :154
    std::vector<double> y(kFlasherProfilePoints);
    for (int i = 0; i < kFlasherProfilePoints; ++i) {
        const double x = 0.5 * i;                                   // numpy.linspace(0., 120., 240, endpoint=False)
        if (fb_width <= 15.) {                                      // :126-127
            y[i] = pulse_narrow(x * (15. / fb_width));
        } else {                                                    // :128-131
            const double plateau = (fb_width - 15.) * 59.5 / (124. - 15.);
            const double rising = std::log(fb_width - 12.) * 1.91 + 5.;
            if (x <= rising) y[i] = rising_edge(x, rising);
            else if (x <= rising + plateau) y[i] = 1.;
            else y[i] = falling_edge(x - rising - plateau);
        }
    }
    return y;
}

void interpolated_distribution_tables(double spacing, const std::vector<double> &y, std::vector<float> &density, std::vector<float> &cumulative)
{
    const size_t n = y.size();
    if (n < 2) throw Error(CLSIMHIP_ERR_ARGUMENT, "At least two entries have to be specified for an interpolated distribution.");
    std::vector<double> acu(n, 0.);
    for (size_t j = 1; j < n; ++j) acu[j] = acu[j - 1] + spacing * (y[j] + y[j - 1]) / 2.;
    const double total = acu[n - 1];
    if (!(total > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "the distribution has no weight");
    density.resize(n); cumulative.resize(n);
    for (size_t j = 0; j < n; ++j) { density[j] = to_float_literal(y[j] / total); cumulative[j] = to_float_literal(acu[j] / total); }
}

uint64_t plan_flasher_steps(const clsimhip_flasher_config &cfg, const clsimhip_flasher_request *requests, size_t n,
                            std::vector<FlasherPlanEntry> &plan, std::vector<double> &widths)
{
    if (cfg.photons_per_step == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "photonsPerStep may not be <= 0!");
    if (cfg.bunch_size_granularity == 0 || cfg.max_bunch_size == 0 || cfg.max_bunch_size % cfg.bunch_size_granularity != 0)
        throw Error(CLSIMHIP_ERR_ARGUMENT, "maxBunchSize must be a non-zero multiple of the bunch size granularity");
    const clsimhip_distribution *dists[3] = {&cfg.polar, &cfg.azimuthal, &cfg.time_delay};
    for (const clsimhip_distribution *d : dists)
        if (d->kind < CLSIMHIP_DIST_CONSTANT || d->kind > CLSIMHIP_DIST_FLASHER_TIME_PROFILE) throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown distribution kind");
    if (cfg.polar.kind == CLSIMHIP_DIST_FLASHER_TIME_PROFILE || cfg.azimuthal.kind == CLSIMHIP_DIST_FLASHER_TIME_PROFILE)
        throw Error(CLSIMHIP_ERR_ARGUMENT, "the flasher time profile is a time delay distribution");
    plan.assign(n, FlasherPlanEntry());
    widths.clear();
    const uint64_t pps = cfg.photons_per_step, max_bunch = cfg.max_bunch_size, gran = cfg.bunch_size_granularity;
    uint64_t out = 0;
    for (size_t i = 0; i < n; ++i) {
        const clsimhip_flasher_request &q = requests[i];
        const float f[10] = {q.x, q.y, q.z, q.time, q.dx, q.dy, q.dz, q.sigma_polar, q.sigma_azimuthal, q.pulse_width};
        for (float v : f) if (!std::isfinite(v)) throw Error(CLSIMHIP_ERR_ARGUMENT, "flasher pulse with a non-finite field");
        if (!(q.dx * q.dx + q.dy * q.dy + q.dz * q.dz > 0.f)) throw Error(CLSIMHIP_ERR_ARGUMENT, "flasher pulse without a direction");
        FlasherPlanEntry &e = plan[i];
        e.first_out = out;
        // MakeSteps (Flasher.cxx:355-400), applied until the pulse is used up
        uint64_t photons = q.num_photons_with_bias, real = 0, total = 0;
        uint32_t last = static_cast<uint32_t>(pps);
        const uint64_t per_result = max_bunch * pps;
        const uint64_t whole = photons / per_result;                 // results of maxBunchSize full steps
        real += whole * max_bunch; total += whole * max_bunch;
        photons -= whole * per_result;
        if (photons > 0 || whole == 0) {
            uint64_t steps;
            uint32_t in_last;
            if (photons <= pps) { steps = 1; in_last = static_cast<uint32_t>(photons); }
            else {
                steps = photons / pps;
                in_last = static_cast<uint32_t>(photons % pps);
                if (in_last > 0) ++steps;
            }
            // a last step without photons is replaced by a dummy step (:407-408) -- also when the photons divide evenly
            // into steps, which drops photons_per_step photons of such a pulse (reference behaviour, kept)
            if (in_last == 0) { real += steps - 1; last = static_cast<uint32_t>(pps); }
            else { real += steps; last = in_last; }
            total += ((steps + gran - 1) / gran) * gran;              // :393-400
        }
        e.n_real = real;
        e.last_real = last;
        if (cfg.time_delay.kind == CLSIMHIP_DIST_FLASHER_TIME_PROFILE) {
            if (!(q.pulse_width > 0.f)) throw Error(CLSIMHIP_ERR_ARGUMENT, "the flasher time profile needs a positive pulse width");
            const double w = q.pulse_width;
            const auto it = std::find(widths.begin(), widths.end(), w);
            e.profile = static_cast<uint32_t>(it - widths.begin());
            if (it == widths.end()) widths.push_back(w);
        }
        out += total;
    }
    return out;
}

} // namespace clsimhip
