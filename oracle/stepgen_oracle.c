/*
 * oracle/stepgen_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the step producer (clsim_amd/csrc/steps_kernel.hip), i.e. of the step generation arithmetic of
 * private/clsim/I3CLSimLightSourceToStepConverterPPC.cxx (FillStep :524-551, FeederThread :744-762, GenerateStep :785-819,
 * GenerateStepForMuon :821-842) and ...ConverterUtils.h:72-105, 140-175, in single precision with the math library of
 * oracle_math.h.  PARITY STATUS: the reference draws from an MWC stream seeded by I3RandomService plus four racing
 * feeder threads and computes in double, so its step sequence is not reproducible (not even by itself) and no fixture
 * exists; this oracle pins the PRODUCT bit for bit, and tests/test_stepgen.py checks the distributions it produces against
 * the closed forms of the reference's formulas.  Only tests/ may use this file.
 */
#include <stdint.h>
#include <string.h>
#include "oracle_math.h"

typedef struct {
    float x, y, z, time, dx, dy, dz, length, pa, pb;
    uint32_t kind, identifier, photons_per_step, num_photons_in_last_step;
    uint64_t num_steps;
} sg_request;

typedef struct __attribute__((packed)) {
    float pos[4], dir[4];
    uint32_t numPhotons; float weight; uint32_t identifier; uint8_t sourceType, dummy1; uint16_t dummy2;
} sg_step;

#define SG_A 4294967118u
#define SG_PI 3.14159265359f

static uint64_t splitmix(uint64_t *state)
{
    *state += 0x9E3779B97F4A7C15ull;
    uint64_t z = *state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t stream_state(uint64_t seed, uint64_t index)
{
    uint64_t s = seed ^ (index * 0xD6E8FEB86659FD93ull);
    for (;;) {
        const uint64_t x = splitmix(&s);
        const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
        if ((x != 0ull) && (hi < SG_A - 1u) && (lo < 0xffffffffu)) return x;
    }
}
static float uniform_co(uint64_t *x)
{
    *x = (*x & 0xffffffffull) * (uint64_t)SG_A + (*x >> 32);
    const uint32_t lo = (uint32_t)*x;
    if (lo == 0) return 0.0f;
    const int drop = 8 - __builtin_clz(lo);
    const uint32_t t = (drop > 0) ? ((lo >> drop) << drop) : lo;
    return (float)t * 2.3283064365386963e-10f;
}
static float uniform_oc(uint64_t *x) { return 1.0f - uniform_co(x); }

static float gamma_distributed(float shape, uint64_t *x)
{
    float v = 0.0f;
    if (shape < 1.0f) {
        const float c = 1.0f / shape;
        const float d = (1.0f - shape) * om_powr(shape, shape / (1.0f - shape));
        for (int tries = 0; tries < 256; ++tries) {
            const float z = -om_log(uniform_oc(x));
            const float e = -om_log(uniform_oc(x));
            v = om_powr(z, c);
            if (!(z + e < d + v)) break;
        }
    } else {
        const float b = shape - 1.3862943611198906f;
        const float l = om_sqrt(2.0f * shape - 1.0f);
        const float cheng = 2.504077396776274f;
        for (int tries = 0; tries < 256; ++tries) {
            const float rx = uniform_oc(x);
            const float ry = uniform_oc(x);
            const float y = om_log(ry / (1.0f - ry)) / l;
            v = shape * om_exp(y);
            const float z = rx * ry * ry;
            const float r = b + (shape + l) * y - v;
            if (!((r < 4.5f * z - cheng) && (r < om_log(z)))) break;
        }
    }
    return v;
}
static void rotate_direction(float cosa, float sina, float *x, float *y, float *z, float u)
{
    float sinb, cosb;
    om_sincos(2.0f * SG_PI * u, &sinb, &cosb);
    const float t = 1.0f - *z * *z;
    const float sinth = om_sqrt((t > 0.0f) ? t : 0.0f);
    if (sinth > 0.0f) {
        const float ox = *x, oy = *y, oz = *z;
        *x = ox * cosa - (oy * cosb + oz * ox * sinb) * sina / sinth;
        *y = oy * cosa + (ox * cosb - oz * oy * sinb) * sina / sinth;
        *z = oz * cosa + sina * sinb * sinth;
    } else {
        *x = sina * cosb;
        *y = sina * sinb;
        *z = (*z >= 0.0f) ? cosa : -cosa;
    }
    const float recip_length = 1.0f / om_sqrt(*x * *x + *y * *y + *z * *z);
    *x *= recip_length; *y *= recip_length; *z *= recip_length;
}

void oracle_generate_steps(const sg_request *requests, const uint64_t *first_step, uint32_t n_requests, uint64_t total_real,
                           uint64_t total_padded, uint64_t seed, sg_step *out)
{
    for (uint64_t g = 0; g < total_padded; ++g) {
        sg_step s;
        memset(&s, 0, sizeof s);
        if (g >= total_real) {
            s.dir[0] = om_acos(-1.0f); s.dir[3] = 1.0f;
            out[g] = s;
            continue;
        }
        uint32_t lo = 0, hi = n_requests - 1u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (first_step[mid] <= g) lo = mid; else hi = mid - 1u;
        }
        const sg_request q = requests[lo];
        const uint64_t k = g - first_step[lo];
        uint64_t x = stream_state(seed, g);
        float dx = q.dx, dy = q.dy, dz = q.dz, along = 0.0f;
        s.dir[2] = 0.001f;
        if (q.kind == 2) {
            s.dir[2] = q.length;
        } else {
            along = (q.kind == 0) ? q.pb * gamma_distributed(q.pa, &x) : uniform_co(&x) * q.length;
            const float ang_a_inv = 1.0f / 0.39f, ang_b = 2.61f;
            const float ang_i = 1.0f - om_exp(-ang_b * om_powr(2.0f, 0.39f));
            const float inner = -om_log(1.0f - uniform_co(&x) * ang_i) / ang_b;
            float cosv = 1.0f - om_powr(inner, ang_a_inv);
            cosv = (cosv > -1.0f) ? cosv : -1.0f;
            const float sinv = om_sqrt(1.0f - cosv * cosv);
            rotate_direction(cosv, sinv, &dx, &dy, &dz, uniform_co(&x));
        }
        s.pos[0] = q.x + along * q.dx;
        s.pos[1] = q.y + along * q.dy;
        s.pos[2] = q.z + along * q.dz;
        s.pos[3] = q.time + along / 0.299792458f;
        const float r_inv = 1.0f / om_sqrt(dx * dx + dy * dy + dz * dz);
        float cz = dz * r_inv;
        cz = (cz > 1.0f) ? 1.0f : ((cz < -1.0f) ? -1.0f : cz);
        s.dir[0] = om_acos(cz);
        float phi = om_atan2(dy, dx);
        if (phi < 0.0f) phi += 2.0f * SG_PI;
        s.dir[1] = phi;
        s.dir[3] = 1.0f;
        s.numPhotons = (k < q.num_steps) ? q.photons_per_step : q.num_photons_in_last_step;
        s.weight = 1.0f;
        s.identifier = q.identifier;
        out[g] = s;
    }
}

/* ---- flasher pulses: clsim_amd/csrc/steps_kernel.hip generate_flasher_steps_kernel, i.e.
 * I3CLSimLightSourceToStepConverterFlasher::FillStep (Flasher.cxx:443-545) with the reference's OpenCL forms of the
 * distributions (NormalDistribution.cxx:68-80, Uniform.cxx, Constant.cxx, InterpolatedDistribution.cxx:236-336) ---- */
typedef struct { int32_t kind; float value; } sg_distribution;
typedef struct {
    sg_distribution polar, azimuthal, time_delay;
    int32_t interpret_in_polar_coordinates;
    uint32_t photons_per_step, max_bunch_size, bunch_size_granularity;
} sg_flasher_config;
typedef struct {
    float x, y, z, time, dx, dy, dz, sigma_polar, sigma_azimuthal, pulse_width;
    uint32_t identifier, source_type;
    uint64_t num_photons_with_bias;
} sg_flasher_request;
typedef struct { uint64_t first_out, n_real; uint32_t last_real, profile; } sg_flasher_plan;

static float sample_distribution(sg_distribution d, float parameter, uint64_t *x, const float *profile)
{
    if (d.kind == 0) return parameter;
    if (d.kind == 1) {
        const float rnd1 = uniform_oc(x);
        const float rnd2 = uniform_oc(x);
        float s, c;
        om_sincos(2.0f * SG_PI * rnd2, &s, &c);
        return (om_sqrt(-2.0f * om_log(rnd1)) * s) * parameter + d.value;
    }
    if (d.kind == 2) return uniform_co(x) * (parameter - d.value) + d.value;
    const float r = uniform_oc(x);
    const float *yv = profile, *cum = profile + 240;
    int k = 0;                                                   /* the reference's linear scan (:262-270) */
    float this_acu = 0.0f;
    for (;;) {
        const float next_acu = cum[k + 1];
        if (next_acu >= r || k + 1 >= 239) break;
        this_acu = next_acu;
        ++k;
    }
    const float b = yv[k];
    const float sp = 0.5f;
    const float x0 = (float)k * sp + 0.0f;
    const float slope = (yv[k + 1] - b) / sp;
    const float dy = r - this_acu;
    if ((b == 0.0f) && (slope == 0.0f)) return x0;
    else if (b == 0.0f) return x0 + om_sqrt(2.0f * dy / slope);
    else if (slope == 0.0f) return x0 + dy / b;
    else return x0 + (om_sqrt(dy * (2.0f * slope) / (b * b) + 1.0f) - 1.0f) * b / slope;
}

void oracle_generate_flasher_steps(const sg_flasher_config *cfg, const sg_flasher_request *requests, const sg_flasher_plan *plan,
                                   uint32_t n_requests, uint64_t total, uint64_t seed, const float *profiles, sg_step *out)
{
    uint32_t r = 0;
    for (uint64_t g = 0; g < total; ++g) {
        while (r + 1 < n_requests && plan[r + 1].first_out <= g) ++r;
        const sg_flasher_request q = requests[r];
        const sg_flasher_plan e = plan[r];
        const uint64_t k = g - e.first_out;
        sg_step s;
        memset(&s, 0, sizeof s);
        if (k >= e.n_real) {
            s.dir[3] = 1.0f;
            s.identifier = q.identifier;
            out[g] = s;
            continue;
        }
        uint64_t x = stream_state(seed, g);
        const float *profile = profiles + (size_t)e.profile * 480u;
        const float smear_polar = sample_distribution(cfg->polar, q.sigma_polar, &x, profile);
        const float smear_azimuthal = sample_distribution(cfg->azimuthal, q.sigma_azimuthal, &x, profile);
        float dx = q.dx, dy = q.dy, dz = q.dz;
        {
            const float r_inv = 1.0f / om_sqrt(dx * dx + dy * dy + dz * dz);
            dx *= r_inv; dy *= r_inv; dz *= r_inv;
        }
        if (!cfg->interpret_in_polar_coordinates) {
            const float cz = (dz > 1.0f) ? 1.0f : ((dz < -1.0f) ? -1.0f : dz);
            const float polar = om_acos(cz);
            float azimuth = om_atan2(dy, dx);
            if (azimuth < 0.0f) azimuth += 2.0f * SG_PI;
            const float smeared_azimuth = azimuth + smear_azimuthal;
            const float lift = (1.5707963267948966f - polar) + smear_polar;
            float sa, ca, sl, cl;
            om_sincos(smeared_azimuth, &sa, &ca);
            om_sincos(lift, &sl, &cl);
            dx = ca * cl; dy = sa * cl; dz = sl;
        } else {
            float sina, cosa, sinb, cosb;
            om_sincos(smear_polar, &sina, &cosa);
            om_sincos(smear_azimuthal, &sinb, &cosb);
            const float t = 1.0f - dz * dz;
            const float sinth = om_sqrt((t > 0.0f) ? t : 0.0f);
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
            const float recip_length = 1.0f / om_sqrt(dx * dx + dy * dy + dz * dz);
            dx *= recip_length; dy *= recip_length; dz *= recip_length;
        }
        const float delay = sample_distribution(cfg->time_delay, q.pulse_width, &x, profile);
        s.pos[0] = q.x; s.pos[1] = q.y; s.pos[2] = q.z;
        s.pos[3] = q.time + delay;
        {
            const float r_inv = 1.0f / om_sqrt(dx * dx + dy * dy + dz * dz);
            float cz = dz * r_inv;
            cz = (cz > 1.0f) ? 1.0f : ((cz < -1.0f) ? -1.0f : cz);
            s.dir[0] = om_acos(cz);
            float phi = om_atan2(dy, dx);
            if (phi < 0.0f) phi += 2.0f * SG_PI;
            s.dir[1] = phi;
        }
        s.dir[2] = 0.0f;
        s.dir[3] = 1.0f;
        s.numPhotons = (k + 1u == e.n_real) ? e.last_real : cfg->photons_per_step;
        s.weight = 1.0f;
        s.identifier = q.identifier;
        s.sourceType = (uint8_t)(q.source_type & 0xffu);
        out[g] = s;
    }
}
