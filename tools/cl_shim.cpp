// TEST INFRASTRUCTURE (tools/verbatim_cl_check.py): the OpenCL builtins that the reference's kernel files, compiled verbatim
// for x86-64, leave undefined -- and a serial driver that runs propKernel work item by work item.
//
// Math: oracle/oracle_math.h, this repository's single-precision math definition (the reference's would be the OpenCL runtime's
// library, which is not pinned).  Everything else is written from the OpenCL 1.2 specification, section
// 6.12 (conversions 6.2.3).  Each symbol gets the Itanium-mangled name clang gives the OpenCL builtin (address-space
// qualifiers mangle as vendor extensions, hence the asm labels).
#include <cmath>
#include <cstdint>
#include <cstring>

extern "C" {
#include "oracle_math.h"
}

typedef float float4 __attribute__((vector_size(16)));

#define CL(name) __asm__(name)

// ---- work-item functions, async copy, atomics (6.12.1, 6.12.10, cl_khr_global_int32_base_atomics) ----
static thread_local size_t g_global_id = 0;
size_t cl_get_global_id(unsigned) CL("_Z13get_global_idj");
size_t cl_get_global_id(unsigned) { return g_global_id; }
void *cl_async_copy(uint16_t *, const uint16_t *, size_t, void *) CL("_Z21async_work_group_copyPU7CLlocaltPU8CLglobalKtm9ocl_event");
void *cl_async_copy(uint16_t *dst, const uint16_t *src, size_t n, void *event) { std::memcpy(dst, src, n * sizeof(uint16_t)); return event; }
void cl_wait_group_events(int, void **) CL("_Z17wait_group_eventsiPU9CLprivate9ocl_event");
void cl_wait_group_events(int, void **) {}
unsigned cl_atom_inc(volatile unsigned *) CL("_Z8atom_incPU8CLglobalVj");
unsigned cl_atom_inc(volatile unsigned *p) { const unsigned old = *p; *p = old + 1u; return old; }      // returns the old value

// ---- conversions (6.2.3): default rounding of float -> int is toward zero; _rtn = toward minus infinity; _rtz of an
// unsigned word keeps its 24 leading bits ----
int cl_convert_int_f(float) CL("_Z11convert_intf");
int cl_convert_int_f(float v) { return (int)v; }
int cl_convert_int_rtn_f(float) CL("_Z15convert_int_rtnf");
int cl_convert_int_rtn_f(float v) { return (int)std::floor(v); }
unsigned cl_convert_uint_h(unsigned char) CL("_Z12convert_uinth");
unsigned cl_convert_uint_h(unsigned char v) { return v; }
unsigned cl_convert_uint_t(unsigned short) CL("_Z12convert_uintt");
unsigned cl_convert_uint_t(unsigned short v) { return v; }
unsigned long cl_convert_ulong_i(int) CL("_Z13convert_ulongi");          // the bit masks of the search without STOP_PHOTONS_ON_DETECTION
unsigned long cl_convert_ulong_i(int v) { return (unsigned long)(long)v; }
float cl_convert_float_f(float) CL("_Z13convert_floatf");
float cl_convert_float_f(float v) { return v; }
float cl_convert_float_i(int) CL("_Z13convert_floati");
float cl_convert_float_i(int v) { return (float)v; }
float cl_convert_float_s(short) CL("_Z13convert_floats");
float cl_convert_float_s(short v) { return (float)v; }
float cl_convert_float_t(unsigned short) CL("_Z13convert_floatt");
float cl_convert_float_t(unsigned short v) { return (float)v; }
short cl_convert_short_t(unsigned short) CL("_Z13convert_shortt");
short cl_convert_short_t(unsigned short v) { return (short)v; }
unsigned short cl_convert_ushort_t(unsigned short) CL("_Z14convert_ushortt");
unsigned short cl_convert_ushort_t(unsigned short v) { return v; }
float cl_convert_float_rtz_j(unsigned) CL("_Z17convert_float_rtzj");
float cl_convert_float_rtz_j(unsigned v)
{
    if (v == 0u) return 0.0f;
    const int drop = 8 - __builtin_clz(v);              // bits below the 24 leading ones
    if (drop > 0) v &= ~((1u << drop) - 1u);
    return (float)v;                                    // exact now
}

// ---- math (6.12.2) on the repository's math header ----
float cl_cos(float) CL("_Z3cosf");
float cl_cos(float x) { return om_cos(x); }
float cl_sin(float) CL("_Z3sinf");
float cl_sin(float x) { return om_sin(x); }
float cl_exp(float) CL("_Z3expf");
float cl_exp(float x) { return om_exp(x); }
float cl_log(float) CL("_Z3logf");
float cl_log(float x) { return om_log(x); }
float cl_sqrt(float) CL("_Z4sqrtf");
float cl_sqrt(float x) { return om_sqrt(x); }
float cl_rsqrt(float) CL("_Z5rsqrtf");
float cl_rsqrt(float x) { return om_rsqrt(x); }
// acos: this repository's definition differs by call site: the hit record's angles (sphDirFromCar) take
// the arccosine through binary64, the table maker's azimuth (getCoordinates) a single-precision polynomial.  A program has
// one or the other call site alive: saveHit is never called under TABULATE, getCoordinates exists only there.
#ifndef VERBATIM_TABULATE
#define VERBATIM_TABULATE 0
#endif
float cl_acos(float) CL("_Z4acosf");
#if VERBATIM_TABULATE
float cl_acos(float x) { return om_acos_f(x); }
#else
float cl_acos(float x) { return om_acos(x); }
#endif
float cl_atan2(float, float) CL("_Z5atan2ff");
float cl_atan2(float y, float x) { return om_atan2(y, x); }
float cl_fabs(float) CL("_Z4fabsf");
float cl_fabs(float x) { return om_fabs(x); }
// powr: the repository's definition has two branches: exponents in (0, 0.09] -- the simplified Liu
// function's beta, the only positive exponent the program uses -- take the single-word logarithm form, everything else
// (the ice model's negative exponents) the hi+lo form.  oracle/clsim_oracle.c: liu_cos / getScatteringLength / getAbsorptionLength.
float cl_powr(float, float) CL("_Z4powrff");
float cl_powr(float x, float y) { return (y > 0.0f && y <= 0.09f) ? om_powr_unit(x, y) : om_powr(x, y); }
float cl_pown(float, int) CL("_Z4pownfi");
float cl_pown(float x, int n)
{
    if (n == 2) return x * x;                           // the one use: pown(b, 2) in the wavelength generator
    float r = 1.0f;
    for (int i = 0; i < (n < 0 ? -n : n); ++i) r *= x;
    return n < 0 ? 1.0f / r : r;
}
float cl_modf(float, float *) CL("_Z4modffPU9CLprivatef");
float cl_modf(float x, float *ip) { const float t = std::trunc(x); *ip = t; return x - t; }

// ---- common / geometric / integer functions (6.12.4, 6.12.5, 6.12.3) ----
float cl_sign(float) CL("_Z4signf");
float cl_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : x); }         // +-0 for +-0 (and 0 for NaN: not met)
float cl_clamp(float, float, float) CL("_Z5clampfff");
float cl_clamp(float x, float lo, float hi) { const float t = (x > lo) ? x : lo; return (t < hi) ? t : hi; }     // fmin(fmax(x, lo), hi)
float cl_mix(float, float, float) CL("_Z3mixfff");
float cl_mix(float a, float b, float t) { return a + (b - a) * t; }
float cl_maxf(float, float) CL("_Z3maxff");
float cl_maxf(float a, float b) { return (a < b) ? b : a; }
float cl_minf(float, float) CL("_Z3minff");
float cl_minf(float a, float b) { return (b < a) ? b : a; }
int cl_maxi(int, int) CL("_Z3maxii");
int cl_maxi(int a, int b) { return (a < b) ? b : a; }
int cl_mini(int, int) CL("_Z3minii");
int cl_mini(int a, int b) { return (b < a) ? b : a; }
// dot of two float4: the specification leaves the summation order open; this repository fixes ((x + y) + z) + w
float cl_dot4(float4, float4) CL("_Z3dotDv4_fS_");
float cl_dot4(float4 a, float4 b) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }

// ---- builtins only the table maker's variant uses ----
int cl_convert_int_sat_rtn_f(float) CL("_Z19convert_int_sat_rtnf");
int cl_convert_int_sat_rtn_f(float v)
{
    const float f = std::floor(v);
    if (!(f >= -2147483648.0f)) return (f != f) ? 0 : INT32_MIN;       // NaN converts to 0 (6.2.3.3)
    if (f >= 2147483648.0f) return INT32_MAX;
    return (int)f;
}
int cl_clampi(int, int, int) CL("_Z5clampiii");
int cl_clampi(int x, int lo, int hi) { const int t = (x > lo) ? x : lo; return (t < hi) ? t : hi; }
float cl_cbrt(float) CL("_Z4cbrtf");
float cl_cbrt(float x) { return om_cbrt(x); }
float cl_pow(float, float) CL("_Z3powff");
float cl_pow(float x, float y) { return om_pow_frac(x, y); }
float4 cl_cross(float4, float4) CL("_Z5crossDv4_fS_");
float4 cl_cross(float4 a, float4 b)
{
    float4 r = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0], 0.0f};
    return r;
}

// ---- the kernel (STOP_PHOTONS_ON_DETECTION, no TABULATE; with or without SAVE_PHOTON_HISTORY: propagation_kernel.c.cl:406-430) ----
#ifndef VERBATIM_HISTORY
#define VERBATIM_HISTORY 0
#endif
#ifndef VERBATIM_TABULATE
#define VERBATIM_TABULATE 0
#endif
#if VERBATIM_TABULATE
// -DTABULATE (propagation_kernel.c.cl:406-430): steps, reference particle, table entries, entry counters, RNG
extern "C" void propKernel(void *inputSteps, void *referenceParticle, void *outputTableEntries, unsigned *numOutputEntries, uint64_t *MWC_RNG_x,
                           unsigned *MWC_RNG_a);
extern "C" void verbatim_tabulate(void *steps, void *reference, void *entries, unsigned *num_entries, uint64_t *rng_x, unsigned *rng_a, unsigned n_steps)
{
    for (unsigned i = 0; i < n_steps; ++i) {
        g_global_id = i;
        propKernel(steps, reference, entries, num_entries, rng_x, rng_a);
    }
}
#elif VERBATIM_HISTORY
extern "C" void propKernel(unsigned *hitIndex, unsigned maxHitIndex, unsigned short *geoLayerToOMNumIndexPerStringSet, void *inputSteps,
                           void *outputPhotons, float4 *photonHistory, uint64_t *MWC_RNG_x, unsigned *MWC_RNG_a);
#else
extern "C" void propKernel(unsigned *hitIndex, unsigned maxHitIndex, unsigned short *geoLayerToOMNumIndexPerStringSet, void *inputSteps,
                           void *outputPhotons, uint64_t *MWC_RNG_x, unsigned *MWC_RNG_a);
#endif

#if !VERBATIM_TABULATE
// serial NDRange: one work item after the other (the kernel's __local array is a static of the object)
extern "C" unsigned verbatim_run(void *photons, unsigned capacity, unsigned short *layer_to_om, void *steps, uint64_t *rng_x, unsigned *rng_a,
                                 void *histories, unsigned n_steps)
{
    (void)histories;
    unsigned hit_index = 0;
    for (unsigned i = 0; i < n_steps; ++i) {
        g_global_id = i;
#if VERBATIM_HISTORY
        propKernel(&hit_index, capacity, layer_to_om, steps, photons, static_cast<float4 *>(histories), rng_x, rng_a);
#else
        propKernel(&hit_index, capacity, layer_to_om, steps, photons, rng_x, rng_a);
#endif
    }
    return hit_index;
}
#endif
