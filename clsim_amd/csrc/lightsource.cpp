// Particle -> step requests: the front end of I3CLSimLightSourceToStepConverterPPC
// (private/clsim/I3CLSimLightSourceToStepConverterPPC.cxx: Initialize :94-132, EnqueueLightSource :188-470).
//
// For every particle the reference decides how many Cherenkov photons it yields and cuts them into steps; the steps
// themselves are then made by FillStep / GenerateStep (:524-551, :785-842) -- in this library on the GPU
// (steps_kernel.hip), from the request records this file produces.  Host only, no device code.
//
// Three inputs of that decision live in dependencies that are NOT part of the reference tree:
//   * I3SimConstants::ShowerParameters (sim-services): longitudinal profile (a, b) and the electromagnetic fraction of
//     hadronic showers with its fluctuation.  Restated from its published parameterisation (L. Raedel, C. Wiebusch,
//     Astropart. Phys. 38 (2012) 53 and 44 (2013) 102; the constants PPC and sim-services carry).  PARITY UNPINNED: no
//     file of the reference tree holds these numbers or a test vector for them.
//   * I3RandomService::Gaus / Poisson (phys-services, GSL): replaced by a counter-based generator of this file
//     (splitmix64 stream per light source; Box-Muller; Poisson by inversion below mean 30, PTRS transformed rejection
//     above).  The reference's sequence depends on the GSL generator the user configures; only distributions can agree.
//   * gsl_integration_qag (ConverterUtils.cxx:71-105, relative tolerance 1e-5) for the photon yield per metre: here a
//     fixed composite Gauss-Legendre rule, converged far below that tolerance.
// Everything else follows the reference line by line, including the integer arithmetic of the step split and the
// `numStepsFromCascades % usePhotonsPerStep` of :455 (the muon's cascade-like photons in the last step are computed from the
// number of STEPS, not of photons).
#include <algorithm>
#include <cmath>
#include <limits>

#include "lightsource.h"

namespace clsimhip {

namespace {

// I3Units: metre = 1, GeV = 1, g/cm3 = 1e-3 kg / 1e-6 m3 in units where ... only ratios of densities enter
constexpr double kWaterEquivalentDensity = 0.924;       // g/cm3, :285
constexpr double kRadiationLengthTimesDensity = 0.358;  // g/cm3 * m: Lrad = 0.358 (g/cm3) / density

struct Rng {                                            // counter-based: splitmix64
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    double uniform() { return (static_cast<double>(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }      // (0, 1)
    double gauss(double mean, double sigma) { const double u = uniform(), v = uniform(); return mean + sigma * std::sqrt(-2. * std::log(u)) * std::cos(6.283185307179586 * v); }
    uint64_t poisson(double mean)
    {
        if (!(mean > 0.)) return 0;
        if (mean < 30.) {                               // inversion by sequential search
            const double limit = std::exp(-mean);
            double prod = uniform();
            uint64_t k = 0;
            while (prod > limit) { prod *= uniform(); ++k; }
            return k;
        }
        // W. Hoermann, "The transformed rejection method for generating Poisson random variables" (PTRS), 1993
        const double slam = std::sqrt(mean), loglam = std::log(mean);
        const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b, invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.);
        for (;;) {
            const double u = uniform() - 0.5, v = uniform();
            const double us = 0.5 - std::fabs(u);
            const double k = std::floor((2. * a / us + b) * u + mean + 0.43);
            if (us >= 0.07 && v <= vr) return static_cast<uint64_t>(k);
            if (k < 0. || (us < 0.013 && v > us)) continue;
            if (std::log(v) + std::log(invalpha) - std::log(a / (us * us) + b) <= -mean + k * loglam - std::lgamma(k + 1.)) return static_cast<uint64_t>(k);
        }
    }
};

// :322-330 and its three repetitions: Poisson below a mean of 1e7 photons (flashers: 1e6), a non-negative Gaussian above
uint64_t draw_photons(Rng &rng, double mean, double gauss_above = 1e7)
{
    if (mean > gauss_above) {
        double n = 0.;
        do { n = rng.gauss(mean, std::sqrt(mean)); } while (n < 0.);
        if (n > static_cast<double>(std::numeric_limits<uint64_t>::max())) throw Error(CLSIMHIP_ERR_ARGUMENT, "Too many photons for counter. internal limitation.");
        return static_cast<uint64_t>(n);
    }
    return rng.poisson(mean);
}

bool is_electron(int32_t t)
{
    return t == CLSIMHIP_PARTICLE_EMINUS || t == CLSIMHIP_PARTICLE_EPLUS || t == CLSIMHIP_PARTICLE_BREMS || t == CLSIMHIP_PARTICLE_DELTAE ||
           t == CLSIMHIP_PARTICLE_PAIRPROD || t == CLSIMHIP_PARTICLE_GAMMA || t == CLSIMHIP_PARTICLE_PI0;
}
bool is_muon(int32_t t) { return t == CLSIMHIP_PARTICLE_MUMINUS || t == CLSIMHIP_PARTICLE_MUPLUS; }
bool is_tau(int32_t t) { return t == CLSIMHIP_PARTICLE_TAUMINUS || t == CLSIMHIP_PARTICLE_TAUPLUS; }

} // namespace

// I3SimConstants::ShowerParameters(type, E, density) (sim-services; NOT in the reference tree: PARITY UNPINNED).  E in GeV,
// density in g/cm3.  Where the numbers come from, and what could and could not be checked here:
//   * form: longitudinal profile dE/dt ~ t^(a-1) exp(-t/b') with a = alpha + beta ln E and b = X0 / b0 [m]; hadronic
//     cascades scale their light by F = 1 - (E/E0)^(-m) (1 - f0) with a relative width rms0 (ln E)^(-gamma).  This is the
//     parameterisation of L. Raedel & C. Wiebusch, Astropart. Phys. 44 (2013) 102 (electromagnetic cascades: gamma
//     distribution fit of the longitudinal profile, one (alpha, beta, b0) triple for e-, e+ and gamma) and of L. Raedel's
//     RWTH Aachen thesis (2012) behind it for the hadronic species (pi+-, K0L, p, n, pbar: (alpha, beta, b0) and
//     (E0, m, f0, rms0, gamma)), as IceCube's sim-services (I3SimConstants.cxx) and PPC (pro.cu) carry it;
//   * UNVERIFIED: none of the 9 + 48 constants below could be compared with those publications or with I3SimConstants.cxx
//     in this environment (no network; no file of the reference tree holds them; the reference only CALLS the function,
//     I3CLSimLightSourceToStepConverterPPC.cxx:286, 349).  They are written down from the builder's recollection of that
//     source file.  Table and equation numbers are deliberately not quoted: they could not be checked either;
//   * what IS checked: the functional form (tests/test_lightsource.py: closed forms, 1/density scaling of b, clamping of
//     ln E at 0, no extension below 1 GeV, the electromagnetic fraction's mean and width), and that nothing but the number
//     of photons and steps of the benchmark workload depends on these values -- the propagator's parity does not.
// A maintainer adopting the producer side should diff this switch against I3SimConstants::ShowerParameters first.
ShowerParameters shower_parameters(int32_t type, double E, double density)
{
    ShowerParameters p;
    const double logE = std::max(0., std::log(E));
    const double Lrad = kRadiationLengthTimesDensity / density;
    if (is_electron(type)) {
        switch (type) {
        case CLSIMHIP_PARTICLE_EPLUS: p.a = 2.00035 + 0.63190 * logE; p.b = Lrad / 0.63008; break;
        case CLSIMHIP_PARTICLE_GAMMA:
        case CLSIMHIP_PARTICLE_PI0: p.a = 2.83923 + 0.58209 * logE; p.b = Lrad / 0.64526; break;
        default: p.a = 2.01849 + 0.63176 * logE; p.b = Lrad / 0.63207; break;                  // e-, and the stochastic losses
        }
    } else if (!is_muon(type) && !is_tau(type)) {
        double E0, m, f0, rms0, gamma;
        switch (type) {
        case CLSIMHIP_PARTICLE_PIMINUS: p.a = 1.69176636 + 0.40803489 * logE; p.b = Lrad / 0.34108075; E0 = 0.19826506; m = 0.16218006; f0 = 0.31859323; rms0 = 0.94033488; gamma = 1.35070162; break;
        case CLSIMHIP_PARTICLE_K0_LONG: p.a = 1.95948974 + 0.34934666 * logE; p.b = Lrad / 0.34535151; E0 = 0.21687243; m = 0.16861530; f0 = 0.27724987; rms0 = 1.00318874; gamma = 1.37528605; break;
        case CLSIMHIP_PARTICLE_PPLUS: p.a = 1.47495778 + 0.40450398 * logE; p.b = Lrad / 0.35226706; E0 = 0.29579368; m = 0.19373018; f0 = 0.02455403; rms0 = 1.01619344; gamma = 1.45477346; break;
        case CLSIMHIP_PARTICLE_NEUTRON: p.a = 1.57739060 + 0.40631102 * logE; p.b = Lrad / 0.35269455; E0 = 0.66725124; m = 0.19263595; f0 = 0.17559033; rms0 = 1.01414337; gamma = 1.45086895; break;
        // (antiproton: the proton's electromagnetic-fraction constants)
        case CLSIMHIP_PARTICLE_PMINUS: p.a = 1.92249171 + 0.33701751 * logE; p.b = Lrad / 0.34969748; E0 = 0.29579368; m = 0.19373018; f0 = 0.02455403; rms0 = 1.01619344; gamma = 1.45477346; break;
        default: p.a = 1.58357292 + 0.41886807 * logE; p.b = Lrad / 0.33833116; E0 = 0.18791678; m = 0.16267529; f0 = 0.30974123; rms0 = 0.95899551; gamma = 1.35589541; break;   // pi+ and everything else
        }
        const double e = std::max(2.71828183, E);
        p.em_scale = 1. - std::pow(e / E0, -m) * (1. - f0);
        p.em_scale_sigma = p.em_scale * rms0 * std::pow(std::log(e), -gamma);
    }
    if (E < 1.) p.b = 0.;                               // below 1 GeV: no cascade extension
    return p;
}

// NumberOfPhotonsPerMeter (ConverterUtils.cxx:44-105): Frank-Tamm yield for beta = 1, weighted with the wavelength
// generation bias, integrated over photon energy 1/lambda from 1/toWlen to 1/fromWlen
double photons_per_meter(const MediumData &medium, const FunctionData &bias, double from_wlen, double to_wlen)
{
    static const double node[8] = {-0.9602898564975363, -0.7966664774136267, -0.5255324099163290, -0.1834346424956498,
                                   0.1834346424956498, 0.5255324099163290, 0.7966664774136267, 0.9602898564975363};
    static const double weight[8] = {0.1012285362903763, 0.2223810344533745, 0.3137066458778873, 0.3626837833783620,
                                     0.3626837833783620, 0.3137066458778873, 0.2223810344533745, 0.1012285362903763};
    const double lo = 1. / to_wlen, hi = 1. / from_wlen;
    const int panels = 4096;
    const double h = (hi - lo) / panels;
    double sum = 0.;
    for (int i = 0; i < panels; ++i) {
        const double mid = lo + (i + 0.5) * h;
        for (int k = 0; k < 8; ++k) {
            const double energy = mid + 0.5 * h * node[k], wlen = 1. / energy;
            const double n = medium.phase_ref_index(wlen);
            sum += weight[k] * bias.eval(wlen) * (2. * M_PI / 137.) * (1. - 1. / (n * n));
        }
    }
    return sum * 0.5 * h;
}

PPCConverter::PPCConverter(const MediumData &medium, const FunctionData &bias, const PPCConfig &config) : config_(config)
{
    if (config.photons_per_step == 0 || config.high_photons_per_step == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "photonsPerStep may not be <= 0!");      // :60-65
    if (!(config.density > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "medium density must be positive");
    layers_z_start_ = medium.layers_z_start;
    layers_height_ = medium.layers_height;
    // :113-131: per layer; the phase refractive index is the same function in every layer of the supported media
    const double yield = photons_per_meter(medium, bias, medium.min_wlen, medium.max_wlen);
    photons_per_meter_.assign(static_cast<size_t>(medium.num_layers), yield);
}

double PPCConverter::mean_photons_per_meter(int layer) const
{
    if (layer < 0 || static_cast<size_t>(layer) >= photons_per_meter_.size()) throw Error(CLSIMHIP_ERR_ARGUMENT, "no such layer");
    return photons_per_meter_[static_cast<size_t>(layer)];
}

// EnqueueLightSource (:188-470): appends one request (cascade, cascade segment) or two (muon / tau: muon-like, then
// cascade-like steps) to `out`
void PPCConverter::enqueue(const clsimhip_particle &particle, std::vector<clsimhip_step_request> &out) const
{
    // :202-204
    double layer_f = std::max(0., (particle.z - layers_z_start_) / layers_height_);
    size_t layer = static_cast<size_t>(static_cast<uint32_t>(layer_f));
    if (layer >= photons_per_meter_.size()) layer = photons_per_meter_.size() - 1;
    const double density = config_.density;
    const double mean_per_meter = photons_per_meter_[layer];
    const int32_t type = particle.type;
    const bool electron = is_electron(type), muon = is_muon(type), tau = is_tau(type);
    // :272-279: anything else with a PDG code "is probably a hadron"
    const bool hadron = !electron && !muon && !tau;
    const double E = particle.energy;
    const double logE = std::max(0., std::log(E));
    if (!(E >= 0.) || !std::isfinite(E)) throw Error(CLSIMHIP_ERR_ARGUMENT, "particle energy must be finite and non-negative");
    // one random stream per light source: results do not depend on the order particles of different identifiers are
    // enqueued in, and an identifier that comes back gets fresh fluctuations (OccurrenceCounter)
    Rng rng{config_.seed ^ (0xD1B54A32D192ED03ull * (static_cast<uint64_t>(particle.identifier) + 1ull)) ^ occurrences_.mix(particle.identifier)};

    clsimhip_step_request r{};
    r.x = static_cast<float>(particle.x); r.y = static_cast<float>(particle.y); r.z = static_cast<float>(particle.z);
    r.time = static_cast<float>(particle.time);
    r.dx = static_cast<float>(particle.dx); r.dy = static_cast<float>(particle.dy); r.dz = static_cast<float>(particle.dz);
    r.identifier = particle.identifier;

    auto per_step = [&](uint64_t photons) {             // :333-335
        return (static_cast<double>(photons) > config_.use_high_photons_per_step_from) ? static_cast<uint64_t>(config_.high_photons_per_step)
                                                                                       : static_cast<uint64_t>(config_.photons_per_step);
    };
    if (electron || hadron) {
        const double nph = 5.21 * kWaterEquivalentDensity / density;               // :285
        const ShowerParameters sp = shower_parameters(type, E, density);
        double f = 1.;
        if (sp.em_scale_sigma != 0.) {
            do { f = sp.em_scale + sp.em_scale_sigma * rng.gauss(0., 1.); } while ((f < 0.) || (1. < f));      // :290-295
        }
        const double mean = f * mean_per_meter * nph * E;                           // :297
        const uint64_t photons = draw_photons(rng, mean);
        const uint64_t use = per_step(photons);
        r.photons_per_step = static_cast<uint32_t>(use);
        r.num_steps = photons / use;
        r.num_photons_in_last_step = static_cast<uint32_t>(photons % use);
        if (particle.shape == CLSIMHIP_SHAPE_CASCADE_SEGMENT) {                    // :342-357
            if (!(particle.length > 0)) throw Error(CLSIMHIP_ERR_ARGUMENT, "Found a cascade segment with length " + std::to_string(particle.length) + ". This should not be.");
            r.kind = CLSIMHIP_STEPS_MUON_CASCADE;
            r.length = static_cast<float>(particle.length);
        } else {                                                                    // :358-371
            r.kind = CLSIMHIP_STEPS_CASCADE;
            r.pa = static_cast<float>(sp.a);
            r.pb = static_cast<float>(config_.use_cascade_extension ? sp.b : 0.);
        }
        out.push_back(r);
        return;
    }
    // muons and taus (:373-462)
    const double length = std::isnan(particle.length) ? 2000. : particle.length;
    const double extr = 1. + std::max(0.0, 0.1880 + 0.0206 * logE);                 // PPC, June 2018
    const double muon_fraction = 1. / extr;
    const double mean_total = mean_per_meter * length * extr;
    const uint64_t from_muon = draw_photons(rng, mean_total * muon_fraction);
    const uint64_t from_cascades = draw_photons(rng, mean_total * (1. - muon_fraction));
    r.length = static_cast<float>(length);
    {
        const uint64_t use = per_step(from_muon);
        r.kind = CLSIMHIP_STEPS_MUON;
        r.photons_per_step = static_cast<uint32_t>(use);
        r.num_steps = from_muon / use;
        r.num_photons_in_last_step = static_cast<uint32_t>(from_muon % use);
        out.push_back(r);
    }
    {
        const uint64_t use = per_step(from_cascades);
        r.kind = CLSIMHIP_STEPS_MUON_CASCADE;
        r.photons_per_step = static_cast<uint32_t>(use);
        r.num_steps = from_cascades / use;
        r.num_photons_in_last_step = static_cast<uint32_t>(r.num_steps % use);      // :455, as written
        out.push_back(r);
    }
}

// PhotonNumberCorrectionFactorAfterBias (ConverterUtils.cxx:113-214): the bias at the peak for a delta-peak spectrum, else
// the ratio of the spectrum's integral with and without the bias over [from_wlen, to_wlen] (the reference: two
// gsl_integration_qag calls to 1e-5; here the same fixed Gauss-Legendre rule as for the Cherenkov yield)
double flasher_correction_factor(const FunctionData *spectrum, double peak_wavelength, const FunctionData &bias, double from_wlen, double to_wlen)
{
    if (!spectrum) return bias.eval(peak_wavelength);
    static const double node[8] = {-0.9602898564975363, -0.7966664774136267, -0.5255324099163290, -0.1834346424956498,
                                   0.1834346424956498, 0.5255324099163290, 0.7966664774136267, 0.9602898564975363};
    static const double weight[8] = {0.1012285362903763, 0.2223810344533745, 0.3137066458778873, 0.3626837833783620,
                                     0.3626837833783620, 0.3137066458778873, 0.2223810344533745, 0.1012285362903763};
    if (!(to_wlen > from_wlen)) throw Error(CLSIMHIP_ERR_ARGUMENT, "empty wavelength range");
    const int panels = 4096;
    const double h = (to_wlen - from_wlen) / panels;
    double with = 0., without = 0.;
    for (int i = 0; i < panels; ++i)
        for (int k = 0; k < 8; ++k) {
            const double w = from_wlen + (i + 0.5) * h + 0.5 * h * node[k];
            const double v = spectrum->eval(w);
            without += weight[k] * v;
            with += weight[k] * v * bias.eval(w);
        }
    if (!(without > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "flasher spectrum integrates to zero");
    return with / without;
}

// I3CLSimLightSourceToStepConverterFlasher::EnqueueLightSource (Flasher.cxx:214-265): pulses without photons are skipped
void flasher_enqueue(double correction, uint64_t seed, const clsimhip_flasher_pulse *pulses, size_t n, std::vector<clsimhip_flasher_request> &out)
{
    OccurrenceCounter occurrences;          // stateless entry point: the caller owns the seed and varies it from call to call
    for (size_t i = 0; i < n; ++i) {
        const clsimhip_flasher_pulse &p = pulses[i];
        if (!(p.num_photons_no_bias > 0.)) continue;
        const double with_bias = p.num_photons_no_bias * correction;
        if (!(with_bias > 0.)) continue;
        Rng rng{seed ^ (0xD1B54A32D192ED03ull * (static_cast<uint64_t>(p.identifier) + 1ull)) ^ occurrences.mix(p.identifier)};
        const uint64_t photons = draw_photons(rng, with_bias, 1e6);             // :237-253
        if (photons == 0) continue;
        clsimhip_flasher_request r{};
        r.x = p.x; r.y = p.y; r.z = p.z; r.time = p.time; r.dx = p.dx; r.dy = p.dy; r.dz = p.dz;
        r.sigma_polar = p.sigma_polar; r.sigma_azimuthal = p.sigma_azimuthal; r.pulse_width = p.pulse_width;
        r.identifier = p.identifier; r.source_type = p.source_type;
        r.num_photons_with_bias = photons;
        out.push_back(r);
    }
}

} // namespace clsimhip
