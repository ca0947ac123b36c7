// Step producer on the GPU (the part of it that is arithmetic of the reference tree).
//
// The reference turns a light source into I3CLSimSteps on ONE host thread
// (private/clsim/I3CLSimLightSourceToStepConverterPPC.cxx): per step a gamma-distributed position along the
// shower axis (FillStep :524-537, gammaDistributedNumber, ...ConverterUtils.h:72-105), a direction smeared around the axis
// with PPC's angular profile cos = 1 - (-ln(1 - u I)/b)^(1/a), a = 0.39, b = 2.61 (FeederThread :744-762, GenerateStep
// :785-819), or -- for the bare muon -- one step per photon bunch along the whole track (GenerateStepForMuon :821-842).
// At 9e6 steps per second that the propagator consumes, that thread is the bottleneck of a drop-in, so the same
// arithmetic runs here with one lane per output step and the steps are born in HBM.
//
// What is NOT here, because it lives in dependencies outside the reference tree: how many photons a particle yields and the
// shower parameters a, b (I3SimConstants::ShowerParameters, sim-services; NumberOfPhotonsPerMeter via gsl_integration_qag;
// Poisson/Gaus of I3RandomService).  The caller supplies them per request, exactly the fields of the reference's
// CascadeStepData_t / MuonStepData_t (PPC.h).  The reference draws positions from one MWC stream and angles from four
// racing feeder threads, so its step sequence is not reproducible even by itself; here every output step has its own
// MWC stream seeded from (seed, step index), and the single precision math is the deterministic library of the propagator
// (detmath.hip.h), restated in oracle/stepgen_oracle.c for the tests.
#include <hip/hip_runtime.h>

#include "../../include/clsimhip.h"
#include "detmath.hip.h"
#include "kparams.h"

namespace clsimhip {
namespace {

#define DS __device__ __forceinline__

constexpr float kPiS = 3.14159265359f;
constexpr uint32_t kStepMultiplier = 4294967118u;       // the first safeprime multiplier (mwcrng_init.h)

DS uint64_t splitmix(uint64_t &state)
{
    state += 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// a valid MWC state for multiplier a (mwcrng_init.h:105-113)
DS uint64_t stream_state(uint64_t seed, uint64_t index)
{
    uint64_t s = seed ^ (index * 0xD6E8FEB86659FD93ull);
    for (;;) {
        const uint64_t x = splitmix(s);
        const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
        if ((x != 0ull) && (hi < kStepMultiplier - 1u) && (lo < 0xffffffffu)) return x;
    }
}
DS float uniform_co(uint64_t &x)
{
    x = (x & 0xffffffffull) * (uint64_t)kStepMultiplier + (x >> 32);
    const uint32_t lo = (uint32_t)x;
    const int drop = 8 - (int)__clz(lo);
    const uint32_t t = (drop > 0) ? ((lo >> drop) << drop) : lo;
    return (float)t * 2.3283064365386963e-10f;
}
DS float uniform_oc(uint64_t &x) { return 1.0f - uniform_co(x); }

// ...ConverterUtils.h:72-105 (Weibull / Cheng, "stolen from PPC"), single precision; the rejection loops are bounded
DS float gamma_distributed(float shape, uint64_t &x)
{
    float v = 0.0f;
    if (shape < 1.0f) {
        const float c = 1.0f / shape;
        const float d = (1.0f - shape) * dm::powr_(shape, shape / (1.0f - shape));
        for (int tries = 0; tries < 256; ++tries) {
            const float z = -dm::log_(uniform_oc(x));
            const float e = -dm::log_(uniform_oc(x));
            v = dm::powr_(z, c);
            if (!(z + e < d + v)) break;
        }
    } else {
        const float b = shape - 1.3862943611198906f;              // log(4)
        const float l = dm::sqrt_(2.0f * shape - 1.0f);
        const float cheng = 2.504077396776274f;                   // 1 + log(4.5)
        for (int tries = 0; tries < 256; ++tries) {
            const float rx = uniform_oc(x);
            const float ry = uniform_oc(x);
            const float y = dm::log_(ry / (1.0f - ry)) / l;
            v = shape * dm::exp_(y);
            const float z = rx * ry * ry;
            const float r = b + (shape + l) * y - v;
            if (!((r < 4.5f * z - cheng) && (r < dm::log_(z)))) break;
        }
    }
    return v;
}

// ...ConverterUtils.h:140-175
DS void rotate_direction(float cosa, float sina, float &x, float &y, float &z, float u)
{
    float sinb, cosb;
    dm::sincos_(2.0f * kPiS * u, sinb, cosb);
    const float t = 1.0f - z * z;
    const float sinth = dm::sqrt_((t > 0.0f) ? t : 0.0f);
    if (sinth > 0.0f) {
        const float ox = x, oy = y, oz = z;
        x = ox * cosa - (oy * cosb + oz * ox * sinb) * sina / sinth;
        y = oy * cosa + (ox * cosb - oz * oy * sinb) * sina / sinth;
        z = oz * cosa + sina * sinb * sinth;
    } else {
        x = sina * cosb;
        y = sina * sinb;
        z = (z >= 0.0f) ? cosa : -cosa;
    }
    const float recip_length = 1.0f / dm::sqrt_(x * x + y * y + z * z);
    x *= recip_length; y *= recip_length; z *= recip_length;
}

__global__ void __launch_bounds__(256) generate_steps_kernel(const clsimhip_step_request *requests, const uint64_t *first_step,
                                                             uint32_t n_requests, uint64_t total_real, uint64_t total_padded,
                                                             uint64_t seed, DevStep *out)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= total_padded) return;
    DevStep s;
    if (g >= total_real) {
        // NoOpStepTemplate (Async.cxx:246-254): direction (0, 0, -1)
        s.x = s.y = s.z = s.t = 0.0f;
        s.theta = dm::acos_(-1.0f); s.phi = 0.0f; s.length = 0.0f; s.beta = 1.0f;
        s.num_photons = 0u; s.weight = 0.0f; s.identifier = 0u; s.source_type_and_pad = 0u;
        out[g] = s;
        return;
    }
    // the request this step belongs to: last r with first_step[r] <= g
    uint32_t lo = 0, hi = n_requests - 1u;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (first_step[mid] <= g) lo = mid; else hi = mid - 1u;
    }
    const clsimhip_step_request q = requests[lo];
    const uint64_t k = g - first_step[lo];
    uint64_t x = stream_state(seed, g);
    float dx = q.dx, dy = q.dy, dz = q.dz;
    float along = 0.0f;
    s.length = 0.001f;                                              // 1 mm (GenerateStep :805)
    if (q.kind == CLSIMHIP_STEPS_MUON) {
        s.length = q.length;                                        // GenerateStepForMuon :821-842
    } else {
        // FillStep :524-551
        along = (q.kind == CLSIMHIP_STEPS_CASCADE) ? q.pb * gamma_distributed(q.pa, x) : uniform_co(x) * q.length;
        // FeederThread :755-757 with angularDist a = 0.39, b = 2.61
        const float ang_a_inv = 1.0f / 0.39f, ang_b = 2.61f;
        const float ang_i = 1.0f - dm::exp_(-ang_b * dm::powr_(2.0f, 0.39f));
        const float inner = -dm::log_(1.0f - uniform_co(x) * ang_i) / ang_b;
        float cosv = 1.0f - dm::powr_(inner, ang_a_inv);
        cosv = (cosv > -1.0f) ? cosv : -1.0f;
        const float sinv = dm::sqrt_(1.0f - cosv * cosv);
        rotate_direction(cosv, sinv, dx, dy, dz, uniform_co(x));    // GenerateStep :812-816
    }
    s.x = q.x + along * q.dx;                                       // GenerateStep :799-802
    s.y = q.y + along * q.dy;
    s.z = q.z + along * q.dz;
    s.t = q.time + along / 0.299792458f;
    // I3CLSimStep::SetDir (I3CLSimStep.h:128-133): theta, phi of the direction of flight
    const float r_inv = 1.0f / dm::sqrt_(dx * dx + dy * dy + dz * dz);
    float cz = dz * r_inv;
    cz = (cz > 1.0f) ? 1.0f : ((cz < -1.0f) ? -1.0f : cz);
    s.theta = dm::acos_(cz);
    float phi = dm::atan2_(dy, dx);
    if (phi < 0.0f) phi += 2.0f * kPiS;
    s.phi = phi;
    s.beta = 1.0f;
    s.num_photons = (k < q.num_steps) ? q.photons_per_step : q.num_photons_in_last_step;
    s.weight = 1.0f;
    s.identifier = q.identifier;
    s.source_type_and_pad = 0u;                                     // Cherenkov emission
    out[g] = s;
}

// ---- flasher pulses: I3CLSimLightSourceToStepConverterFlasher::FillStep (Flasher.cxx:443-545) ----
struct FlasherPlanEntryDev { uint64_t first_out, n_real; uint32_t last_real, profile; };

// profile tables: [profile][0: density, 1: cumulative][240]
DS float sample_distribution(const clsimhip_distribution d, float parameter, uint64_t &x, const float *profile)
{
    if (d.kind == CLSIMHIP_DIST_CONSTANT) return parameter;                           // Constant.cxx:61-73
    if (d.kind == CLSIMHIP_DIST_NORMAL) {                                              // NormalDistribution.cxx:68-80
        const float rnd1 = uniform_oc(x);
        const float rnd2 = uniform_oc(x);
        float s, c;
        dm::sincos_(2.0f * kPiS * rnd2, s, c);
        return (dm::sqrt_(-2.0f * dm::log_(rnd1)) * s) * parameter + d.value;
    }
    if (d.kind == CLSIMHIP_DIST_UNIFORM) return uniform_co(x) * (parameter - d.value) + d.value;   // Uniform.cxx:101-103
    // InterpolatedDistribution.cxx:236-336 over the pulse shape (x0 = 0, spacing 0.5 ns)
    const float r = uniform_oc(x);
    const float *yv = profile, *cum = profile + 240;
    int lo = 1, hi = 239;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] >= r) hi = mid; else lo = mid + 1;
    }
    const int k = lo - 1;
    const float this_acu = (k == 0) ? 0.0f : cum[k];
    const float b = yv[k];
    const float sp = 0.5f;
    const float x0 = (float)k * sp + 0.0f;
    const float slope = (yv[k + 1] - b) / sp;
    const float dy = r - this_acu;
    if ((b == 0.0f) && (slope == 0.0f)) return x0;
    else if (b == 0.0f) return x0 + dm::sqrt_(2.0f * dy / slope);
    else if (slope == 0.0f) return x0 + dy / b;
    else return x0 + (dm::sqrt_(dy * (2.0f * slope) / (b * b) + 1.0f) - 1.0f) * b / slope;
}

__global__ void __launch_bounds__(256) generate_flasher_steps_kernel(const clsimhip_flasher_config cfg, const clsimhip_flasher_request *requests,
                                                                     const FlasherPlanEntryDev *plan, uint32_t n_requests,
                                                                     uint64_t total, uint64_t seed, const float *profiles, DevStep *out)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= total) return;
    uint32_t lo = 0, hi = n_requests - 1u;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (plan[mid].first_out <= g) lo = mid; else hi = mid - 1u;
    }
    const clsimhip_flasher_request q = requests[lo];
    const FlasherPlanEntryDev e = plan[lo];
    const uint64_t k = g - e.first_out;
    DevStep s;
    if (k >= e.n_real) {
        // dummy steps (:420-434): theta = phi = 0, the pulse's identifier
        s.x = s.y = s.z = s.t = 0.0f;
        s.theta = 0.0f; s.phi = 0.0f; s.length = 0.0f; s.beta = 1.0f;
        s.num_photons = 0u; s.weight = 0.0f; s.identifier = q.identifier; s.source_type_and_pad = 0u;
        out[g] = s;
        return;
    }
    uint64_t x = stream_state(seed, g);
    const float *profile = profiles + (size_t)e.profile * 480u;
    const float smear_polar = sample_distribution(cfg.polar, q.sigma_polar, x, profile);           // :450-460
    const float smear_azimuthal = sample_distribution(cfg.azimuthal, q.sigma_azimuthal, x, profile);
    float dx = q.dx, dy = q.dy, dz = q.dz;
    {
        const float r_inv = 1.0f / dm::sqrt_(dx * dx + dy * dy + dz * dz);
        dx *= r_inv; dy *= r_inv; dz *= r_inv;
    }
    if (!cfg.interpret_in_polar_coordinates) {
        // :468-486: azimuth smeared in the horizontal plane; then the horizontal unit vector is rotated about the
        // horizontal axis perpendicular to it by (90 deg - polar angle) + smearing (right-handed: zero smearing gives
        // the pulse's direction back)
        const float cz = (dz > 1.0f) ? 1.0f : ((dz < -1.0f) ? -1.0f : dz);
        const float polar = dm::acos_(cz);
        float azimuth = dm::atan2_(dy, dx);
        if (azimuth < 0.0f) azimuth += 2.0f * kPiS;
        const float smeared_azimuth = azimuth + smear_azimuthal;
        const float lift = (1.5707963267948966f - polar) + smear_polar;
        float sa, ca, sl, cl;
        dm::sincos_(smeared_azimuth, sa, ca);
        dm::sincos_(lift, sl, cl);
        dx = ca * cl; dy = sa * cl; dz = sl;
    } else {
        // :488-541: polar = how far from the old direction, azimuthal = at which orientation around it
        float sina, cosa, sinb, cosb;
        dm::sincos_(smear_polar, sina, cosa);
        dm::sincos_(smear_azimuthal, sinb, cosb);
        const float t = 1.0f - dz * dz;
        const float sinth = dm::sqrt_((t > 0.0f) ? t : 0.0f);
        if (sinth > 0.0f) {
            const float ox = dx, oy = dy, oz = dz;
            dx = ox * cosa - ((oy * cosb + oz * ox * sinb) * sina / sinth);
            dy = oy * cosa + ((ox * cosb - oz * oy * sinb) * sina / sinth);
            dz = oz * cosa + sina * sinb * sinth;
        } else {
            dx = sina * cosb;
            dy = sina * sinb;
            dz = cosa * ((dz < 0.0f) ? -1.0f : 1.0f);
        }
        const float recip_length = 1.0f / dm::sqrt_(dx * dx + dy * dy + dz * dz);
        dx *= recip_length; dy *= recip_length; dz *= recip_length;
    }
    const float delay = sample_distribution(cfg.time_delay, q.pulse_width, x, profile);           // :544-551
    s.x = q.x; s.y = q.y; s.z = q.z;
    s.t = q.time + delay;
    {   // I3CLSimStep::SetDir
        const float r_inv = 1.0f / dm::sqrt_(dx * dx + dy * dy + dz * dz);
        float cz = dz * r_inv;
        cz = (cz > 1.0f) ? 1.0f : ((cz < -1.0f) ? -1.0f : cz);
        s.theta = dm::acos_(cz);
        float phi = dm::atan2_(dy, dx);
        if (phi < 0.0f) phi += 2.0f * kPiS;
        s.phi = phi;
    }
    s.length = 0.0f;
    s.beta = 1.0f;
    s.num_photons = (k + 1u == e.n_real) ? e.last_real : cfg.photons_per_step;
    s.weight = 1.0f;
    s.identifier = q.identifier;
    s.source_type_and_pad = q.source_type & 0xffu;
    out[g] = s;
}

} // namespace

hipError_t launch_generate_flasher_steps(const clsimhip_flasher_config &cfg, const clsimhip_flasher_request *d_requests, const void *d_plan,
                                         uint32_t n_requests, uint64_t total, uint64_t seed, const float *d_profiles, void *d_out,
                                         hipStream_t stream)
{
    if (total == 0) return hipSuccess;
    const uint64_t blocks = (total + 255u) / 256u;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(generate_flasher_steps_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, cfg, d_requests,
                       static_cast<const FlasherPlanEntryDev *>(d_plan), n_requests, total, seed, d_profiles, static_cast<DevStep *>(d_out));
    return hipGetLastError();
}

hipError_t launch_generate_steps(const clsimhip_step_request *d_requests, const uint64_t *d_first_step, uint32_t n_requests,
                                 uint64_t total_real, uint64_t total_padded, uint64_t seed, void *d_out, hipStream_t stream)
{
    if (total_padded == 0) return hipSuccess;
    const uint64_t blocks = (total_padded + 255u) / 256u;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(generate_steps_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, d_requests, d_first_step, n_requests,
                       total_real, total_padded, seed, static_cast<DevStep *>(d_out));
    return hipGetLastError();
}

} // namespace clsimhip
