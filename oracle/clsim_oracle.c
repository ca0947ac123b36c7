/*
 * clsim_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of clsim's photon propagator `propKernel`
 * (resources/kernels/propagation_kernel.c.cl:406-913) together with the DOM
 * collision search (resources/kernels/sparse_collision_kernel.c.cl:27-587), the
 * MWC random number generator (resources/kernels/mwcrng_kernel.cl:12-28) and
 * the run-time generated medium / spectrum / geometry functions, which the
 * reference emits as OpenCL text (private/opencl/I3CLSimHelperGenerate*.cxx and
 * the GetOpenCLFunction() bodies under private/clsim/function, random_value);
 * here they are evaluated from tables (struct oracle_tables) that
 * oracle/builders.py fills with the float literals the reference would print.
 *
 * Every function cites the reference file:line it follows.  The arithmetic is
 * the reference's single precision expression order with NO implicit fma
 * contraction (build with -ffp-contract=off) and the math library of
 * oracle_math.h in place of the OpenCL runtime builtins.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this file.  PARITY STATUS: the reference holds no golden vector for
 * propKernel output and cannot be built here (needs
 * IceTray, boost, an OpenCL CPU runtime and run-time generated code), so the
 * whole-kernel output of this oracle is UNPINNED by the reference; the
 * sub-functions that the reference's tests and data files do pin (safeprime
 * multipliers, anisotropy scaling, SPICE-Lea transforms, ice loader tables)
 * are pinned in tests/ against fixtures generated from those files.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_math.h"

/* Counting build (make liboracle_count.so, -DORACLE_COUNT_OPS; count_ops.hpp): this file compiled as C++ with its `float`s
 * replaced by a counting type.  REGION() names the part of the kernel the operations that follow belong to (until the end
 * of the enclosing block), EVENT() counts an occurrence; both vanish from the normal build, whose code is unchanged. */
#ifdef ORACLE_COUNT_OPS
#include "count_ops.hpp"
#include "count_ops_calls.hpp"
#else
#define REGION(r) ((void)0)
#define REGION_UNLESS_CREATING(r) ((void)0)
#define EVENT(e, n) ((void)0)
#define COUNT_OP(op) ((void)0)
#define COUNT_FOLD() ((void)0)
#endif

/* ---- wire structs: public/clsim/I3CLSimStep.h:141-155, I3CLSimPhoton.h:194-213;
 *      propagation_kernel.h.cl:52-81 ---- */
typedef struct __attribute__((packed)) {
    float pos[4];               /* x,y,z,time */
    float dir[4];               /* theta,phi,length,beta */
    uint32_t numPhotons;
    float weight;
    uint32_t identifier;
    uint8_t sourceType;
    uint8_t dummy1;
    uint16_t dummy2;
} oracle_step;                  /* 48 bytes */

typedef struct __attribute__((packed)) {
    float pos[4];
    float dir[2];
    float wavelength;
    float cherenkovDist;
    uint32_t numScatters;
    float weight;
    uint32_t identifier;
    int16_t stringID;
    uint16_t omID;
    float startPos[4];
    float startDir[2];
    float groupVelocity;
    float distInAbsLens;
} oracle_photon;                /* 80 bytes */

_Static_assert(sizeof(oracle_step) == 48, "step size");
_Static_assert(sizeof(oracle_photon) == 80, "photon size");

#define ORACLE_MAX_GEN 8
#define ORACLE_MAX_SUBDET 9

typedef struct {
    int32_t kind;               /* 0 interpolated (const spacing), 1 constant, 2 Cherenkov without dispersion,
                                   3 interpolated with its own x values (InterpolatedDistribution.cxx:41-55, :292-297) */
    int32_t n;
    float first, spacing;       /* literals of InterpolatedDistribution.cxx:250-266 */
    const float *yv;            /* _distYValues */
    const float *ycum;          /* _distYCumulativeValues */
    float value;                /* RandomValueConstant */
    const float *xv;            /* _distXValues (kind 3) */
} oracle_wlen_gen;

typedef struct {
    /* ---- compile-time switches of the reference (OpenCL.cxx:390-442) ---- */
    int32_t stop_detected;      /* STOP_PHOTONS_ON_DETECTION (OpenCL.cxx:395-397); 0: detected photons travel on */
    int32_t has_pancake;        /* PANCAKE_FACTOR defined (pancakeFactor != 1) */
    float pancake;
    /* ---- medium (MediumPropertiesSource.cxx:207-389) ---- */
    int32_t num_layers;
    float layer_bottom, layer_thickness;
    int32_t len_mode;           /* 0: per-layer FunctionConstant, 1: IceCube abs/scat, 2: per-layer FromTable */
    const float *abs_const, *sca_const;
    const float *aDust400, *deltaTau, *b400;
    float kappa, A, B, D, E, alpha, ref_wlen_recip, nanometer;
    float n[5], g[5], micrometer, c_light;
    int32_t scat_kind;          /* 0 HG, 1 SimplifiedLiu, 2 Mixed(Liu,HG) */
    float mix_frac, mix_frac_rest, liu_beta, hg_g, hg_g2;
    int32_t has_abs_corr;       /* ScalarFieldAnisotropyAbsLenScaling vs constant */
    float abs_corr_const;
    float an_l[3], an_rl[3], an_azx, an_azy, an_mazy, an_B2;
    int32_t has_pre, pre_renorm, has_post, post_renorm;
    float pre[9], post[9];
    int32_t has_tilt;           /* ScalarFieldIceTiltZShift vs constant */
    float tilt_const;
    int32_t tilt_nd, tilt_nz;
    float tilt_first_z, tilt_dz, tilt_lnx, tilt_lny;
    const float *tilt_dist, *tilt_zcorr;
    /* ---- spectra ---- */
    int32_t num_gen;
    oracle_wlen_gen gen[ORACLE_MAX_GEN];
    int32_t bias_kind;          /* 0 FromTable, 1 Constant */
    int32_t bias_n;
    float bias_start, bias_step, bias_value;
    const float *bias_data;
    /* ---- geometry (GeometrySource.cxx:1153-1269, 619-700) ---- */
    int32_t num_strings;
    float om_radius, string_max_radius;
    const float *str_x, *str_y, *str_minz, *str_maxz;
    const uint8_t *str_set;
    int32_t num_sets, max_layers;
    const uint16_t *set_nlayers;
    const float *set_startz, *set_height;
    const uint16_t *layer_to_om;        /* geoLayerToOMNumIndexPerStringSet */
    int32_t num_subdet;
    int32_t cell_nx[ORACLE_MAX_SUBDET], cell_ny[ORACLE_MAX_SUBDET];
    float cell_wx[ORACLE_MAX_SUBDET], cell_wy[ORACLE_MAX_SUBDET];
    float cell_sx[ORACLE_MAX_SUBDET], cell_sy[ORACLE_MAX_SUBDET];
    const uint16_t *cell_index[ORACLE_MAX_SUBDET];
    float dom_mul_x, dom_mul_y;
    const int16_t *dom_tx, *dom_ty;
    const float *dom_tz;
    const uint32_t *dom_start;
    const float *dom_meanx, *dom_meany;
    /* ---- tabulated medium functions (FunctionFromTable.cxx:167-300; MakeIceCubeMediumPropertiesPhotonics.py) ---- */
    int32_t tab_n;              /* len_mode 2: one FromTable function per layer, common binning */
    float tab_start, tab_step;
    int32_t tab_store16;        /* storeDataAsHalfPrecision: 16-bit linear quantisation */
    const uint16_t *abs_q, *sca_q;      /* [layer][tab_n] */
    const float *abs_lo, *abs_hi, *sca_lo, *sca_hi;     /* per layer _SMALLEST_ENTRY / _LARGEST_ENTRY */
    const float *abs_f, *sca_f;         /* [layer][tab_n] float literals when not quantised */
    int32_t phase_mode, group_mode;     /* 0 RefIndexIceCube, 1 FromTable (float data); group_mode 2: no override, from the dispersion */
    int32_t phase_n, group_n;
    float phase_start, phase_step, group_start, group_step;
    const float *phase_data, *group_data;
    /* ---- more compile-time switches (OpenCL.cxx:416-431) ---- */
    int32_t has_fixed_abs;      /* PROPAGATE_FOR_FIXED_NUMBER_OF_ABSORPTION_LENGTHS defined */
    float fixed_abs;
    int32_t history_n;          /* NUM_PHOTONS_IN_HISTORY (SAVE_PHOTON_HISTORY when > 0) */
    /* ---- TABULATE (tabulator/I3CLSimStepToTableConverter.cxx:178-207, Axes.cxx, Axis.cxx) ---- */
    int32_t tab_axes_kind;      /* 0 spherical_coordinates.c.cl, 1 cylindrical_coordinates.c.cl */
    int32_t tab_full_azimuth;   /* HAS_FULL_AZIMUTH_EXTENSION */
    int32_t tab_ndim;           /* 4, or 5 = TABULATE_IMPACT_ANGLE (StepToTableConverter.cxx:187-188) */
    float tab_scale[5], tab_offset[5];  /* Axis::GetIndexCode literals */
    int32_t tab_inverse[5];     /* inverse transform: the axis' power (0, 1 identity, 2 sqrt, 3 cbrt, above pow(x, tab_inv_exp)) */
    float tab_inv_exp[5];       /* ToFloatString(1./power), Axis.cxx:168 */
    int32_t tab_nbins[5];
    uint32_t tab_stride[5];
    float tab_max0, tab_max3;   /* isOutOfBounds */
    float tab_min_inv_groupvel, tab_tan_thetac;
    float tab_volume_step;      /* VOLUME_MODE_STEP */
    uint32_t tab_entries_per_stream;
    int32_t ang_n;              /* getAngularAcceptance: FunctionPolynomial (Polynomial.cxx:96-153) */
    float ang_coeff[16];
    int32_t ang_has_min, ang_has_max;
    float ang_min, ang_max, ang_underflow, ang_overflow;
} oracle_tables;

typedef struct { uint64_t x; uint32_t a; } rng_t;

#ifdef ORACLE_TRACE
/* divergence study (tools/divergence_model.py): per loop iteration of one work item:
 * [0] created a photon, [1] layer-walk trips, [2] cells visited, [3] strings tested,
 * [4] DOM-layer trips, [5] SimplifiedLiu branch, [6] scattered (not absorbed), [7] hit */
static __thread uint8_t *g_trace = 0;
static __thread uint64_t g_trace_cap = 0, g_trace_n = 0;
static __thread uint8_t g_cur[8];
#define TR(i, v) (g_cur[i] = (uint8_t)((g_cur[i] + (v)) > 255 ? 255 : (g_cur[i] + (v))))
#else
#define TR(i, v) ((void)0)
#endif

/* mwcrng_kernel.cl:12-20 */
static inline float rand_co(rng_t *r)
{
    COUNT_OP(OC_RNG);                               /* the draw is the unit; its conversion and scaling go to a region of their own */
    REGION(OCR_RNG_INTERNAL);
    r->x = (r->x & 0xffffffffull) * (uint64_t)r->a + (r->x >> 32);
    const uint32_t lo = (uint32_t)(r->x & 0xffffffffull);
    /* convert_float_rtz(uint): keep the top 24 significant bits */
    float f;
    if (lo == 0) f = 0.0f;
    else {
        const int lz = __builtin_clz(lo);
        const int drop = 8 - lz;
        const uint32_t t = (drop > 0) ? ((lo >> drop) << drop) : lo;
        f = (float)t;                              /* exact */
    }
    return f / 4294967296.0f;
}
/* mwcrng_kernel.cl:25-28 */
static inline float rand_oc(rng_t *r) { return 1.0f - rand_co(r); }

static inline float sqr(float a) { return a * a; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline float clampf(float v, float lo, float hi) { return fminf_(fmaxf_(v, lo), hi); }

/* ---------------- generated medium functions ---------------- */

/* FunctionFromTable.cxx:213-232: <name>_getInterpolationBinAndFraction */
static inline void tableBinAndFraction(float start, float step, int n, float wavelength, int *bin, float *fraction)
{
    const float q = (wavelength - start) / step;
    const float fbin = __builtin_truncf(q);
    *fraction = q - fbin;                           /* modf */
    int ibin = (int)fbin;
    if ((ibin < 0) || ((ibin == 0) && (*fraction < 0))) { ibin = 0; *fraction = 0.0f; }
    else if (ibin >= n - 1) { ibin = n - 2; *fraction = 1.0f; }
    *bin = ibin;
}
/* FunctionFromTable.cxx:279-291: float data */
static inline float fromTableFloat(float start, float step, int n, const float *data, float wavelength)
{
    int bin; float fraction;
    tableBinAndFraction(start, step, n, wavelength, &bin, &fraction);
    const float a = data[bin], b = data[bin + 1];
    return a + (b - a) * fraction;                  /* mix */
}
/* FunctionFromTable.cxx:262-275: 16-bit data between _SMALLEST_ENTRY and _LARGEST_ENTRY */
static inline float fromTable16(float start, float step, int n, const uint16_t *data, float lo, float hi, float wavelength)
{
    int bin; float fraction;
    tableBinAndFraction(start, step, n, wavelength, &bin, &fraction);
    const float a = (float)data[bin] * ((hi - lo) / 65535.f) + lo;
    const float b = (float)data[bin + 1] * ((hi - lo) / 65535.f) + lo;
    return a + (b - a) * fraction;
}

/* RefIndexIceCube.cxx:128-180 */
static inline float getPhaseRefIndex(const oracle_tables *T, float wlen)
{
    REGION(OCR_MEDIUM_PER_PHOTON);
    if (T->phase_mode == 1) return fromTableFloat(T->phase_start, T->phase_step, T->phase_n, T->phase_data, wlen);
    const float x = wlen / T->micrometer;
    return T->n[0] + x * (T->n[1] + x * (T->n[2] + x * (T->n[3] + x * T->n[4])));
}
static inline float getGroupRefIndex(const oracle_tables *T, float wlen)
{
    REGION(OCR_MEDIUM_PER_PHOTON);
    if (T->group_mode == 1) return fromTableFloat(T->group_start, T->group_step, T->group_n, T->group_data, wlen);
    const float x = wlen / T->micrometer;
    const float np = T->n[0] + x * (T->n[1] + x * (T->n[2] + x * (T->n[3] + x * T->n[4])));
    const float np_corr = T->g[0] + x * (T->g[1] + x * (T->g[2] + x * (T->g[3] + x * T->g[4])));
    return np * np_corr;
}
/* RefIndexIceCube.cxx:205-215, the phase index's GetOpenCLFunctionDerivative (`x*4.f*n4` is (x*4.f)*n4) */
static inline float getDispersion(const oracle_tables *T, float wlen)
{
    const float x = wlen / T->micrometer;
    const float dnp = (T->n[1] + x * (2.f * T->n[2] + x * (3.f * T->n[3] + x * 4.f * T->n[4]))) / T->micrometer;
    return dnp;
}
/* MediumPropertiesSource.cxx:255-272 (group index override path); group_mode 2 = no override, :274-300 (from the dispersion) */
static inline float getGroupVelocity(const oracle_tables *T, float wlen)
{
    REGION(OCR_MEDIUM_PER_PHOTON);
    if (T->group_mode == 2) {
        const float n_inv = 1.f / getPhaseRefIndex(T, wlen);
        const float y = getDispersion(T, wlen);
        return T->c_light * (1.0f + y * wlen * n_inv) * n_inv;
    }
    return T->c_light / getGroupRefIndex(T, wlen);
}
/* _Optimizers.cxx:195-250 / FunctionConstant.cxx:81-100 */
static inline float getScatteringLength(const oracle_tables *T, int layer, float wlen)
{
    REGION(OCR_LAYER_LENGTHS);
    EVENT(OCE_LAYER_LENGTH_EVALS, 1);
    if (T->len_mode == 0) return T->sca_const[layer];
    if (T->len_mode == 2) {                         /* MediumPropertiesSource.cxx:89-123: switch(layer) over _func<k> */
        const size_t o = (size_t)layer * (size_t)T->tab_n;
        if (T->tab_store16) return fromTable16(T->tab_start, T->tab_step, T->tab_n, T->sca_q + o, T->sca_lo[layer], T->sca_hi[layer], wlen);
        return fromTableFloat(T->tab_start, T->tab_step, T->tab_n, T->sca_f + o, wlen);
    }
    return 1.0f / (T->b400[layer] * om_powr(wlen * T->ref_wlen_recip, -T->alpha));
}
/* _Optimizers.cxx:123-190 */
static inline float getAbsorptionLength(const oracle_tables *T, int layer, float wlen)
{
    REGION(OCR_LAYER_LENGTHS);
    if (T->len_mode == 0) return T->abs_const[layer];
    if (T->len_mode == 2) {
        const size_t o = (size_t)layer * (size_t)T->tab_n;
        if (T->tab_store16) return fromTable16(T->tab_start, T->tab_step, T->tab_n, T->abs_q + o, T->abs_lo[layer], T->abs_hi[layer], wlen);
        return fromTableFloat(T->tab_start, T->tab_step, T->tab_n, T->abs_f + o, wlen);
    }
    const float x = wlen / T->nanometer;
    return 1.0f / ((T->D * T->aDust400[layer] + T->E) * om_powr(x, -T->kappa)
                   + T->A * om_exp(-T->B / x) * (1.0f + 0.01f * T->deltaTau[layer]));
}
/* HenyeyGreenstein.cxx:69-92 */
static inline float hg_cos(const oracle_tables *T, float rnd_co)
{
    const float s = 2.0f * rnd_co - 1.0f;
    const float ii = ((1.0f - T->hg_g2) / (1.0f + T->hg_g * s));
    return clampf((1.0f + T->hg_g2 - ii * ii) / (2.0f * T->hg_g), -1.0f, 1.0f);
}
/* SimplifiedLiu.cxx:64-88 */
static inline float liu_cos(const oracle_tables *T, float rnd_co)
{
    /* beta * 22.2 <= 2 (u >= 2^-32): the single-word logarithm form; otherwise the general powr */
    if (T->liu_beta <= 0.09f) return clampf(2.0f * om_powr_unit(rnd_co, T->liu_beta) - 1.0f, -1.0f, 1.0f);
    return clampf(2.0f * om_powr(rnd_co, T->liu_beta) - 1.0f, -1.0f, 1.0f);
}
/* Mixed.cxx:115-157 (single random number form) */
static inline float makeScatteringCosAngle(const oracle_tables *T, rng_t *rng)
{
    REGION(OCR_SCATTER_ANGLE);
    if (T->scat_kind == 0) return hg_cos(T, rand_co(rng));
    if (T->scat_kind == 1) return liu_cos(T, rand_co(rng));
    const float rr = rand_co(rng);
    if (rr < T->mix_frac) { TR(5, 1); EVENT(OCE_LIU, 1); return liu_cos(T, rr / T->mix_frac); }
    EVENT(OCE_HG, 1);
    return hg_cos(T, (1.0f - rr) / T->mix_frac_rest);
}
/* ScalarFieldAnisotropyAbsLenScaling.cxx:92-140 / ScalarFieldConstant.cxx:61-80 */
static inline float getDirectionalAbsLenCorrFactor(const oracle_tables *T, const float d[4])
{
    REGION(OCR_ANISO);
    if (!T->has_abs_corr) return T->abs_corr_const;
    const float n0 = (T->an_azx * d[0]) + (T->an_azy * d[1]);
    const float n1 = (T->an_mazy * d[0]) + (T->an_azx * d[1]);
    const float n2 = d[2];
    const float s0 = n0 * n0, s1 = n1 * n1, s2 = n2 * n2, s3 = 0.0f * 0.0f;
    /* dot(float4,float4): fixed as ((x+y)+z)+w */
    const float nB = ((s0 * T->an_rl[0] + s1 * T->an_rl[1]) + s2 * T->an_rl[2]) + s3 * 0.0f;
    const float An = ((s0 * T->an_l[0] + s1 * T->an_l[1]) + s2 * T->an_l[2]) + s3 * 0.0f;
    return 2.0f / ((T->an_B2 - nB) * An);
}
/* VectorTransformMatrix.cxx:101-135 / VectorTransformConstant.cxx:58-74 */
static inline void transformDirection(int has, int renorm, const float m[9], float d[4])
{
    REGION(OCR_TRANSFORM);
    if (!has) return;
    const float x = (m[0] * d[0]) + (m[1] * d[1]) + (m[2] * d[2]);
    const float y = (m[3] * d[0]) + (m[4] * d[1]) + (m[5] * d[2]);
    const float z = (m[6] * d[0]) + (m[7] * d[1]) + (m[8] * d[2]);
    d[0] = x; d[1] = y; d[2] = z;
    if (renorm) {
        const float norm = om_rsqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] = d[0] * norm; d[1] = d[1] * norm; d[2] = d[2] * norm;
    }
}
/* ScalarFieldIceTiltZShift.cxx:145-213 */
static inline float getTiltZShift(const oracle_tables *T, const float p[4])
{
    REGION(OCR_TILT);
    const float z_rescaled = (p[2] - T->tilt_first_z) / T->tilt_dz;
    const int k = imin(imax((int)__builtin_floorf(z_rescaled), 0), T->tilt_nz - 2);
    const float fraction_z_above = z_rescaled - (float)k;
    const float fraction_z_below = 1.0f - fraction_z_above;
    const float nr = T->tilt_lnx * p[0] + T->tilt_lny * p[1];
    for (int j = 1; j < T->tilt_nd; j++) {
        const float thisDist = T->tilt_dist[j];
        if ((nr < thisDist) || (j == T->tilt_nd - 1)) {
            const float previousDist = T->tilt_dist[j - 1];
            const float thisDistanceBinWidth = thisDist - previousDist;
            const float frac_at_lower = (thisDist - nr) / thisDistanceBinWidth;
            const float frac_at_upper = 1.0f - frac_at_lower;
            const float val_at_lower = (T->tilt_zcorr[(j - 1) * T->tilt_nz + k + 1] * fraction_z_above
                                        + T->tilt_zcorr[(j - 1) * T->tilt_nz + k] * fraction_z_below);
            const float val_at_upper = (T->tilt_zcorr[j * T->tilt_nz + k + 1] * fraction_z_above
                                        + T->tilt_zcorr[j * T->tilt_nz + k] * fraction_z_below);
            return (val_at_upper * frac_at_upper + val_at_lower * frac_at_lower);
        }
    }
    return 0.0f;
}
/* InterpolatedDistribution.cxx:236-336 / RandomValueConstant.cxx */
static inline float generateWavelength_k(const oracle_tables *T, int kgen, rng_t *rng)
{
    REGION(OCR_WAVELENGTH);
    const oracle_wlen_gen *G = &T->gen[kgen];
    if (G->kind == 1) return G->value;
    if (G->kind == 2) {                             /* WlenCherenkovNoDispersion.cxx:72-92: first = minVal, spacing = range */
        const float r = rand_oc(rng);
        return 1.f / (G->first + r * G->spacing);
    }
    const float randomNumber = rand_oc(rng);
    unsigned int k = 0;
    float this_acu = 0.0f;
    for (;;) {
        float next_acu = G->ycum[k + 1];
        if (next_acu >= randomNumber) break;
        this_acu = next_acu;
        ++k;
    }
    const float b = G->yv[k];
    float x0, slope;
    if (G->kind == 3) {                             /* :292-297 */
        x0 = G->xv[k];
        slope = (G->yv[k + 1] - b) / (G->xv[k + 1] - x0);
    } else {                                        /* :298-303 */
        x0 = (float)k * (G->spacing) + (G->first);
        slope = (G->yv[k + 1] - b) / (G->spacing);
    }
    const float dy = randomNumber - this_acu;
    if ((b == 0.0f) && (slope == 0.0f)) return x0;
    else if (b == 0.0f) return x0 + om_sqrt(2.0f * dy / slope);
    else if (slope == 0.0f) return x0 + dy / b;
    else return x0 + (om_sqrt(dy * (2.0f * slope) / (b * b) + 1.0f) - 1.0f) * b / slope;
}
/* MediumPropertiesSource.cxx:392-432 */
static inline float generateWavelength(const oracle_tables *T, uint32_t number, rng_t *rng)
{
    if (T->num_gen == 0) return 0.0f;
    if (T->num_gen == 1) return generateWavelength_k(T, 0, rng);
    if (number < (uint32_t)T->num_gen) return generateWavelength_k(T, (int)number, rng);
    return 0.0f;
}
/* FunctionFromTable.cxx:167-300 / FunctionConstant.cxx */
static inline float getWavelengthBias(const oracle_tables *T, float wavelength)
{
    if (T->bias_kind == 1) return T->bias_value;
    const float q = (wavelength - T->bias_start) / T->bias_step;
    const float fbin = __builtin_truncf(q);
    float fraction = q - fbin;                      /* modf */
    int ibin = (int)fbin;
    if ((ibin < 0) || ((ibin == 0) && (fraction < 0))) { ibin = 0; fraction = 0.0f; }
    else if (ibin >= T->bias_n - 1) { ibin = T->bias_n - 2; fraction = 1.0f; }
    const float a = T->bias_data[ibin], b = T->bias_data[ibin + 1];
    return a + (b - a) * fraction;                  /* mix */
}
/* GeometrySource.cxx:685-700 */
static inline void geometryGetDomPosition(const oracle_tables *T, unsigned s, unsigned d,
                                          float *x, float *y, float *z)
{
    const unsigned int index = T->dom_start[s] + d;
    *x = (float)T->dom_tx[index] * T->dom_mul_x + T->dom_meanx[s];
    *y = (float)T->dom_ty[index] * T->dom_mul_y + T->dom_meany[s];
    *z = T->dom_tz[index];
}

/* ---------------- propagation_kernel.c.cl ---------------- */

#define SPEED_OF_LIGHT 0.299792458f     /* h.cl:148 */
#define PI_F 3.14159265359f             /* h.cl:150 */
#define EPSILON 0.00001f                /* c.cl:505 */

/* c.cl:73-81 */
static inline int findLayerForGivenZPos(const oracle_tables *T, float z)
{
    return (int)((z - T->layer_bottom) / T->layer_thickness);
}
static inline float mediumLayerBoundary(const oracle_tables *T, int layer)
{
    return ((float)layer * T->layer_thickness) + T->layer_bottom;
}

/* c.cl:83-129 */
static void scatterDirectionByAngle(float cosa, float sina, float d[4], float randomNumber)
{
    REGION_UNLESS_CREATING(OCR_ROTATE);
    const float b = 2.0f * PI_F * randomNumber;
    float cosb, sinb;
    om_sincos(b, &sinb, &cosb);
    const float sinth = om_sqrt(fmaxf_(0.0f, 1.0f - d[2] * d[2]));
    if (sinth > 0.0f) {
        const float ox = d[0], oy = d[1], oz = d[2];
        d[0] = ox * cosa - ((oy * cosb + oz * ox * sinb) * sina) / sinth;
        d[1] = oy * cosa + ((ox * cosb - oz * oy * sinb) * sina) / sinth;
        d[2] = oz * cosa + sina * sinb * sinth;
    } else {
        const float sgn = (d[2] > 0.0f) ? (float)1.0f : ((d[2] < 0.0f) ? (float)-1.0f : d[2]);
        d[0] = sina * cosb;
        d[1] = sina * sinb;
        d[2] = cosa * sgn;
    }
    {
        const float recip_length = om_rsqrt(sqr(d[0]) + sqr(d[1]) + sqr(d[2]));
        d[0] *= recip_length; d[1] *= recip_length; d[2] *= recip_length;
    }
}

/* c.cl:132-184 */
static void createPhotonFromTrack(const oracle_tables *T, const oracle_step *step, const float stepDir[4],
                                  rng_t *rng, float pos[4], float dirw[4])
{
    REGION(OCR_CREATE);
    const float shiftMultiplied = step->dir[2] * rand_co(rng);
    const float inverseParticleSpeed = 1.0f / (SPEED_OF_LIGHT * step->dir[3]);
    pos[0] = step->pos[0] + stepDir[0] * shiftMultiplied;
    pos[1] = step->pos[1] + stepDir[1] * shiftMultiplied;
    pos[2] = step->pos[2] + stepDir[2] * shiftMultiplied;
    pos[3] = step->pos[3] + inverseParticleSpeed * shiftMultiplied;
    /* NO_FLASHER (OpenCL.cxx:648-650): with <=1 spectrum every step is Cherenkov */
    if (T->num_gen <= 1 || step->sourceType == 0) {
        const float wavelength = generateWavelength_k(T, 0, rng);
        const float cosCherenkov = fminf_(1.0f, 1.0f / (step->dir[3] * getPhaseRefIndex(T, wavelength)));
        const float sinCherenkov = om_sqrt(1.0f - cosCherenkov * cosCherenkov);
        dirw[0] = stepDir[0]; dirw[1] = stepDir[1]; dirw[2] = stepDir[2];
        dirw[3] = wavelength;
        scatterDirectionByAngle(cosCherenkov, sinCherenkov, dirw, rand_co(rng));
    } else {
        const float wavelength = generateWavelength(T, (uint32_t)step->sourceType, rng);
        dirw[0] = stepDir[0]; dirw[1] = stepDir[1]; dirw[2] = stepDir[2];
        dirw[3] = wavelength;
    }
}

/* c.cl:206-223 */
static void sphDirFromCar(const float d[4], float out[2])
{
    const float r_inv = om_rsqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float theta = 0.0f;
    if (om_fabs(d[2] * r_inv) <= 1.0f) theta = om_acos(d[2] * r_inv);
    else if (d[2] < 0.0f) theta = PI_F;
    if (theta < 0.0f) theta += 2.0f * PI_F;
    float phi = om_atan2(d[1], d[0]);
    if (phi < 0.0f) phi += 2.0f * PI_F;
    out[0] = theta; out[1] = phi;
}

#define ORACLE_MAX_HISTORY 64
typedef struct {
    oracle_photon *out;
    uint32_t max_hits;
    uint32_t count;             /* keeps counting past max_hits (c.cl:329-330) */
    float *hist_out;            /* photonHistory: history_n float4 per output slot, or NULL */
    const float *cur_hist;      /* currentPhotonHistory of the work item */
} hit_sink;

/* c.cl:307-404 */
static void saveHit(const oracle_tables *T, const float pos[4], const float dirw[4], float thisStepLength,
                    float inv_groupvel, float totalPath, uint32_t numScatters, float distAbsLens,
                    const float startPos[4], const float startDirw[4], const oracle_step *step,
                    unsigned hitOnString, unsigned hitOnDom, hit_sink *sink)
{
    REGION(OCR_HIT_RECORD);
    EVENT(OCE_HITS, 1);
    const uint32_t myIndex = sink->count++;
    if (myIndex >= sink->max_hits) return;
    oracle_photon *o = &sink->out[myIndex];
    float domPosX, domPosY, domPosZ;
    geometryGetDomPosition(T, hitOnString, hitOnDom, &domPosX, &domPosY, &domPosZ);
    if (T->has_pancake) {
        const float px = pos[0] - domPosX, py = pos[1] - domPosY, pz = pos[2] - domPosZ;
        const float parallel = px * dirw[0] + py * dirw[1] + pz * dirw[2];
        const float nx = px - parallel * dirw[0];
        const float ny = py - parallel * dirw[1];
        const float nz = pz - parallel * dirw[2];
        const float f = (T->pancake - 1.0f) / T->pancake;
        domPosX += f * nx; domPosY += f * ny; domPosZ += f * nz;
    }
    o->pos[0] = pos[0] + thisStepLength * dirw[0] - domPosX;
    o->pos[1] = pos[1] + thisStepLength * dirw[1] - domPosY;
    o->pos[2] = pos[2] + thisStepLength * dirw[2] - domPosZ;
    o->pos[3] = pos[3] + thisStepLength * inv_groupvel;
    sphDirFromCar(dirw, o->dir);
    o->wavelength = dirw[3];
    o->cherenkovDist = totalPath + thisStepLength;
    o->numScatters = numScatters;
    o->weight = step->weight / getWavelengthBias(T, dirw[3]);
    o->identifier = step->identifier;
    o->stringID = (int16_t)hitOnString;
    o->omID = (uint16_t)hitOnDom;
    memcpy(o->startPos, startPos, 16);
    sphDirFromCar(startDirw, o->startDir);
    o->groupVelocity = 1.0f / inv_groupvel;
    o->distInAbsLens = distAbsLens;
    if (T->history_n > 0 && sink->hist_out)        /* c.cl:387-392: the whole ring, valid or not */
        memcpy(sink->hist_out + (size_t)myIndex * 4u * (size_t)T->history_n, sink->cur_hist,
               (size_t)T->history_n * 16u);
}

/* sparse_collision_kernel.c.cl:27-192 (STOP_PHOTONS_ON_DETECTION branch) */
static void checkForCollision_OnString(const oracle_tables *T, unsigned stringNum, float dirLenXYSqr,
                                       const float pos[4], const float dirw[4], float *thisStepLength,
                                       int *hitRecorded, unsigned *hitOnString, unsigned *hitOnDom)
{
    REGION(OCR_SEARCH_STRING);
    EVENT(OCE_STRINGS, 1);
    const unsigned stringSet = T->str_set[stringNum];
    {
        const float smin = sqr(((pos[0] - T->str_x[stringNum]) * dirw[1]
                                - (pos[1] - T->str_y[stringNum]) * dirw[0])) / dirLenXYSqr;
        if (smin > sqr(T->string_max_radius)) return;
    }
    {
        if ((dirw[2] > 0.0f) && (pos[2] > T->str_maxz[stringNum] + T->om_radius)) return;
        if ((dirw[2] < 0.0f) && (pos[2] < T->str_minz[stringNum] - T->om_radius)) return;
    }
    int lowLayerZ = (int)((pos[2] - T->set_startz[stringSet]) / T->set_height[stringSet]);
    int highLayerZ = (int)((pos[2] + dirw[2] * (*thisStepLength) - T->set_startz[stringSet]) / T->set_height[stringSet]);
    if (highLayerZ < lowLayerZ) { int tmp = lowLayerZ; lowLayerZ = highLayerZ; highLayerZ = tmp; }
    lowLayerZ = imin(imax(lowLayerZ, 0), (int)T->set_nlayers[stringSet] - 1);
    highLayerZ = imin(imax(highLayerZ, 0), (int)T->set_nlayers[stringSet] - 1);

    const uint16_t *geoLayerToOMNumIndex = T->layer_to_om + (stringSet * (unsigned)T->max_layers) + lowLayerZ;
    TR(3, 1);
    for (int layer_z = lowLayerZ; layer_z <= highLayerZ; ++layer_z, ++geoLayerToOMNumIndex) {
        TR(4, 1);
        const unsigned domNum = *geoLayerToOMNumIndex;
        if (domNum == 0xFFFF) continue;
        REGION(OCR_SEARCH_DOM);
        EVENT(OCE_DOM_TESTS, 1);
        float domPosX, domPosY, domPosZ;
        geometryGetDomPosition(T, stringNum, domNum, &domPosX, &domPosY, &domPosZ);
        float urdot, discr;
        {
            const float dx = domPosX - pos[0], dy = domPosY - pos[1], dz = domPosZ - pos[2], dw = 0.0f;
            /* dot(): ((x*x+y*y)+z*z)+w*w */
            const float dr2 = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
            urdot = ((dx * dirw[0] + dy * dirw[1]) + dz * dirw[2]) + dw * dirw[3];
            discr = sqr(urdot) - dr2 + T->om_radius * T->om_radius;
        }
        if (discr < 0.0f) continue;
        if (T->has_pancake) discr = om_sqrt(discr) / T->pancake;
        else discr = om_sqrt(discr);
        {
            const float smin2 = urdot + discr;
            if (smin2 < 0.0f) continue;
        }
        const float smin1 = urdot - discr;
        if (smin1 < 0.0f) continue;
        if (smin1 < *thisStepLength) {
            *thisStepLength = smin1;
            *hitOnString = stringNum;
            *hitOnDom = domNum;
            *hitRecorded = 1;
        }
    }
}

/* sparse_collision_kernel.c.cl:194-303 */
static void checkForCollision_InCell(const oracle_tables *T, int sd, float dirLenXYSqr, const float pos[4],
                                     const float dirw[4], float *thisStepLength, int *hitRecorded,
                                     unsigned *hitOnString, unsigned *hitOnDom)
{
    REGION(OCR_SEARCH_CELLS);
    const float sx = T->cell_sx[sd], sy = T->cell_sy[sd], wx = T->cell_wx[sd], wy = T->cell_wy[sd];
    const int nx = T->cell_nx[sd], ny = T->cell_ny[sd];
    int lowCellX = (int)((pos[0] - sx) / wx);
    int lowCellY = (int)((pos[1] - sy) / wy);
    int highCellX = (int)((pos[0] + dirw[0] * (*thisStepLength) - sx) / wx);
    int highCellY = (int)((pos[1] + dirw[1] * (*thisStepLength) - sy) / wy);
    if (highCellX < lowCellX) { int tmp = lowCellX; lowCellX = highCellX; highCellX = tmp; }
    if (highCellY < lowCellY) { int tmp = lowCellY; lowCellY = highCellY; highCellY = tmp; }
    lowCellX = imin(imax(lowCellX, 0), nx - 1);
    lowCellY = imin(imax(lowCellY, 0), ny - 1);
    highCellX = imin(imax(highCellX, 0), nx - 1);
    highCellY = imin(imax(highCellY, 0), ny - 1);
    for (int cell_y = lowCellY; cell_y <= highCellY; ++cell_y) {
        for (int cell_x = lowCellX; cell_x <= highCellX; ++cell_x) {
            TR(2, 1);
            EVENT(OCE_CELLS, 1);
            const unsigned stringNum = T->cell_index[sd][cell_y * nx + cell_x];
            if (stringNum == 0xFFFF) continue;
            checkForCollision_OnString(T, stringNum, dirLenXYSqr, pos, dirw, thisStepLength,
                                       hitRecorded, hitOnString, hitOnDom);
        }
    }
}

/* sparse_collision_kernel.c.cl:462-587 */
static int checkForCollision(const oracle_tables *T, const float pos[4], const float dirw[4], float inv_groupvel,
                             float totalPath, uint32_t numScatters, float distAbsLens, const float startPos[4],
                             const float startDirw[4], const oracle_step *step, float *thisStepLength,
                             hit_sink *sink)
{
    REGION(OCR_SEARCH_CELLS);
    EVENT(OCE_SEARCH_CALLS, 1);
    const float dirLenXYSqr = sqr(dirw[0]) + sqr(dirw[1]);
    if (dirLenXYSqr <= 0.0f) return 0;
    int hitRecorded = 0;
    unsigned hitOnString = 0, hitOnDom = 0;
    for (int sd = 0; sd < T->num_subdet; ++sd)
        checkForCollision_InCell(T, sd, dirLenXYSqr, pos, dirw, thisStepLength, &hitRecorded,
                                 &hitOnString, &hitOnDom);
    if (hitRecorded)
        saveHit(T, pos, dirw, *thisStepLength, inv_groupvel, totalPath, numScatters, distAbsLens, startPos,
                startDirw, step, hitOnString, hitOnDom, sink);
    return hitRecorded;
}

/* ---- the same search without STOP_PHOTONS_ON_DETECTION (SetStopDetectedPhotons(false), OpenCL.cxx:395-397): every DOM the
 * segment enters is saved on the spot with the distance to it, the step is not shortened, the photon travels on
 * (sparse_collision_kernel.c.cl:165-186, :580-584; propagation_kernel.c.cl:704-750) ---- */
typedef struct {
    float inv_groupvel, totalPath, distAbsLens;
    uint32_t numScatters;
    const float *startPos, *startDirw;
    const oracle_step *step;
    hit_sink *sink;
} keep_ctx;

/* `1 << convert_ulong(n%64)` (c.cl:103-104, :252-253): the literal 1 is an int, so OpenCL takes the shift count modulo 32
 * (6.3.j) and the int result is widened to ulong with its sign: DOMs (strings) n and n+32 share a bit, and bit 31 drags bits
 * 32..63 along */
static inline uint64_t opencl_int_bit(unsigned n) { return (uint64_t)(int64_t)(int32_t)(1u << ((n % 64u) & 31u)); }

/* c.cl:27-192, #ifndef STOP_PHOTONS_ON_DETECTION */
static void checkForCollision_OnString_keep(const oracle_tables *T, unsigned stringNum, float dirLenXYSqr,
                                            const float pos[4], const float dirw[4], float thisStepLength, const keep_ctx *K)
{
    REGION(OCR_SEARCH_STRING);
    EVENT(OCE_STRINGS, 1);
    const unsigned stringSet = T->str_set[stringNum];
    {
        const float smin = sqr(((pos[0] - T->str_x[stringNum]) * dirw[1]
                                - (pos[1] - T->str_y[stringNum]) * dirw[0])) / dirLenXYSqr;
        if (smin > sqr(T->string_max_radius)) return;
    }
    {
        if ((dirw[2] > 0.0f) && (pos[2] > T->str_maxz[stringNum] + T->om_radius)) return;
        if ((dirw[2] < 0.0f) && (pos[2] < T->str_minz[stringNum] - T->om_radius)) return;
    }
    int lowLayerZ = (int)((pos[2] - T->set_startz[stringSet]) / T->set_height[stringSet]);
    int highLayerZ = (int)((pos[2] + dirw[2] * thisStepLength - T->set_startz[stringSet]) / T->set_height[stringSet]);
    if (highLayerZ < lowLayerZ) { int tmp = lowLayerZ; lowLayerZ = highLayerZ; highLayerZ = tmp; }
    lowLayerZ = imin(imax(lowLayerZ, 0), (int)T->set_nlayers[stringSet] - 1);
    highLayerZ = imin(imax(highLayerZ, 0), (int)T->set_nlayers[stringSet] - 1);

    /* :85-90: `ulong dom_bitmask[(GEO_MAX_DOM_INDEX+63)/64]`, zeroed per call, of which a call touches the one word
     * [stringNum/64] (:103-104 index with the STRING number).  That word exists for stringNum < 64*((GEO_MAX_DOM_INDEX+63)/64);
     * beyond (a 65th string of a detector with at most 64 DOMs per string) the reference reads and writes past its array,
     * which is undefined: the restatement behaves as an array that is long enough, i.e. one zeroed word per call. */
    uint64_t dom_bitmask = 0;
    const uint16_t *geoLayerToOMNumIndex = T->layer_to_om + (stringSet * (unsigned)T->max_layers) + lowLayerZ;
    for (int layer_z = lowLayerZ; layer_z <= highLayerZ; ++layer_z, ++geoLayerToOMNumIndex) {
        const unsigned domNum = *geoLayerToOMNumIndex;
        if (domNum == 0xFFFF) continue;
        if (dom_bitmask & opencl_int_bit(domNum)) continue;         /* a DOM named by several layers is tested once */
        dom_bitmask |= opencl_int_bit(domNum);
        REGION(OCR_SEARCH_DOM);
        EVENT(OCE_DOM_TESTS, 1);
        float domPosX, domPosY, domPosZ;
        geometryGetDomPosition(T, stringNum, domNum, &domPosX, &domPosY, &domPosZ);
        float urdot, discr;
        {
            const float dx = domPosX - pos[0], dy = domPosY - pos[1], dz = domPosZ - pos[2], dw = 0.0f;
            const float dr2 = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
            urdot = ((dx * dirw[0] + dy * dirw[1]) + dz * dirw[2]) + dw * dirw[3];
            discr = sqr(urdot) - dr2 + T->om_radius * T->om_radius;
        }
        if (discr < 0.0f) continue;
        if (T->has_pancake) discr = om_sqrt(discr) / T->pancake;
        else discr = om_sqrt(discr);
        {
            const float smin2 = urdot + discr;
            if (smin2 < 0.0f) continue;
        }
        const float smin1 = urdot - discr;
        if (smin1 < 0.0f) continue;
        if (smin1 < thisStepLength)                                  /* :165-186 */
            saveHit(T, pos, dirw, smin1, K->inv_groupvel, K->totalPath, K->numScatters, K->distAbsLens, K->startPos,
                    K->startDirw, K->step, stringNum, domNum, K->sink);
    }
}

/* c.cl:194-303, #ifndef STOP_PHOTONS_ON_DETECTION */
static void checkForCollision_InCell_keep(const oracle_tables *T, int sd, float dirLenXYSqr, const float pos[4],
                                          const float dirw[4], float thisStepLength, const keep_ctx *K)
{
    REGION(OCR_SEARCH_CELLS);
    const float sx = T->cell_sx[sd], sy = T->cell_sy[sd], wx = T->cell_wx[sd], wy = T->cell_wy[sd];
    const int nx = T->cell_nx[sd], ny = T->cell_ny[sd];
    int lowCellX = (int)((pos[0] - sx) / wx);
    int lowCellY = (int)((pos[1] - sy) / wy);
    int highCellX = (int)((pos[0] + dirw[0] * thisStepLength - sx) / wx);
    int highCellY = (int)((pos[1] + dirw[1] * thisStepLength - sy) / wy);
    if (highCellX < lowCellX) { int tmp = lowCellX; lowCellX = highCellX; highCellX = tmp; }
    if (highCellY < lowCellY) { int tmp = lowCellY; lowCellY = highCellY; highCellY = tmp; }
    lowCellX = imin(imax(lowCellX, 0), nx - 1);
    lowCellY = imin(imax(lowCellY, 0), ny - 1);
    highCellX = imin(imax(highCellX, 0), nx - 1);
    highCellY = imin(imax(highCellY, 0), ny - 1);
    /* :245-249: one bit per string, in words of 64, zeroed per subdetector */
    const int words = (T->num_strings + 63) / 64;
    uint64_t string_bitmask[words > 0 ? words : 1];
    for (int i = 0; i < words; ++i) string_bitmask[i] = 0;
    for (int cell_y = lowCellY; cell_y <= highCellY; ++cell_y) {
        for (int cell_x = lowCellX; cell_x <= highCellX; ++cell_x) {
            EVENT(OCE_CELLS, 1);
            const unsigned stringNum = T->cell_index[sd][cell_y * nx + cell_x];
            if (stringNum == 0xFFFF) continue;
            if (string_bitmask[stringNum / 64] & opencl_int_bit(stringNum)) continue;      /* :252 */
            string_bitmask[stringNum / 64] |= opencl_int_bit(stringNum);                   /* :253 */
            checkForCollision_OnString_keep(T, stringNum, dirLenXYSqr, pos, dirw, thisStepLength, K);
        }
    }
}

/* c.cl:462-587, #ifndef STOP_PHOTONS_ON_DETECTION: "this will always return false" */
static int checkForCollision_keep(const oracle_tables *T, const float pos[4], const float dirw[4], float inv_groupvel,
                                  float totalPath, uint32_t numScatters, float distAbsLens, const float startPos[4],
                                  const float startDirw[4], const oracle_step *step, float thisStepLength, hit_sink *sink)
{
    REGION(OCR_SEARCH_CELLS);
    EVENT(OCE_SEARCH_CALLS, 1);
    const float dirLenXYSqr = sqr(dirw[0]) + sqr(dirw[1]);
    if (dirLenXYSqr <= 0.0f) return 0;
    const keep_ctx K = { inv_groupvel, totalPath, distAbsLens, numScatters, startPos, startDirw, step, sink };
    for (int sd = 0; sd < T->num_subdet; ++sd)
        checkForCollision_InCell_keep(T, sd, dirLenXYSqr, pos, dirw, thisStepLength, &K);
    return 0;
}

/* ---------------- TABULATE ---------------- */
typedef struct __attribute__((packed)) { uint32_t index; float weight; } oracle_table_entry;   /* h.cl:83-87 */
typedef struct {
    float posAndTime[4], dir[4], perpDir[4];        /* I3CLSimReferenceParticle, h.cl:89-94 */
} oracle_reference;
typedef struct {
    const oracle_reference *source;
    oracle_table_entry *entries;                    /* this stream's TABLE_ENTRIES_PER_STREAM slots */
    uint32_t *entry_counter;
    uint32_t photons_left_out;                      /* inputSteps[i].numPhotons on return */
} tab_ctx;

static inline float dot4(const float a[4], const float b[4]) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }
static inline float magnitude(const float v[4]) { return om_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* Polynomial.cxx:96-153 */
static inline float getAngularAcceptance(const oracle_tables *T, float x)
{
    if (T->ang_has_min && x < T->ang_min) return T->ang_underflow;
    if (T->ang_has_max && x > T->ang_max) return T->ang_overflow;
    if (T->ang_n == 0) return 0.f;
    float r = T->ang_coeff[T->ang_n - 1];
    for (int i = T->ang_n - 2; i >= 0; --i) r = T->ang_coeff[i] + x * r;     /* c0 + x*(c1 + x*(...)) */
    return r;
}
static void scatterDirectionByAngle(float cosa, float sina, float d[4], float randomNumber);
/* spherical_coordinates.c.cl:39-81 / cylindrical_coordinates.c.cl:39-77; dirw is the function's own copy */
static inline void getCoordinates(const oracle_tables *T, const float absPos[4], float dirw[4], const oracle_reference *source,
                                  rng_t *rng, float coords[5])
{
    float pos[4], rho[4];
    for (int k = 0; k < 4; ++k) pos[k] = absPos[k] - source->posAndTime[k];
    const float l = dot4(pos, source->dir);
    for (int k = 0; k < 4; ++k) rho[k] = pos[k] - l * source->dir[k];
    if (T->tab_axes_kind == 0) {
        const float n_rho = magnitude(rho);
        coords[0] = magnitude(pos);
        const float azimuth = (n_rho > 0) ? om_acos_f(dot4(rho, source->perpDir) / n_rho) / (PI_F / 180) : (float)0.0f;
        if (T->tab_full_azimuth) {
            /* cross(rho, perpDir) . dir */
            const float cx = rho[1] * source->perpDir[2] - rho[2] * source->perpDir[1];
            const float cy = rho[2] * source->perpDir[0] - rho[0] * source->perpDir[2];
            const float cz = rho[0] * source->perpDir[1] - rho[1] * source->perpDir[0];
            const float cv[4] = { cx, cy, cz, 0.0f };
            const float azisign = dot4(cv, source->dir);
            coords[1] = (azisign > 0) ? 360.f - azimuth : azimuth;
        } else {
            coords[1] = azimuth;
        }
        coords[2] = (coords[0] > 0) ? (l / coords[0]) : (float)0.0f;
        coords[3] = pos[3] - coords[0] * T->tab_min_inv_groupvel;
    } else {
        coords[0] = magnitude(rho);
        coords[1] = (coords[0] > 0) ? om_acos_f(dot4(rho, source->perpDir) / coords[0]) : (float)0.0f;
        coords[2] = source->posAndTime[2] + l * source->dir[2];
        coords[3] = pos[3] - (l + coords[0] * T->tab_tan_thetac) * 3.33564095f;
    }
    coords[4] = 0.0f;
    if (T->tab_ndim > 4) {
        /* TABULATE_IMPACT_ANGLE (spherical :67-79, cylindrical :61-72): impact position randomised over the DOM's
         * cross-section; dot() runs over all four components (wavelength * time difference included) */
        const float sina = om_sqrt(rand_co(rng));
        scatterDirectionByAngle(om_sqrt(1 - sina * sina), sina, dirw, rand_co(rng));
        if (T->tab_axes_kind == 0) {
            coords[4] = (coords[0] > 0) ? (dot4(dirw, pos) / coords[0]) : (float)1.0f;
        } else {
            /* cylindrical :70-75: (l - rho*recip(tan_thetaC))*source->dir is evaluated component by component
             * (scalar - vector, vector * vector), as OpenCL does for this expression */
            const float recip_tan = 1.f / T->tab_tan_thetac;
            float cpos[4];
            for (int k = 0; k < 4; ++k)
                cpos[k] = absPos[k] - (source->posAndTime[k] + (l - rho[k] * recip_tan) * source->dir[k]);
            const float cdist = magnitude(cpos);
            coords[4] = (cdist > 0) ? (dot4(dirw, cpos) / cdist) : (float)1.0f;
        }
    }
}
/* Axes.cxx:104-151 */
static inline int isOutOfBounds(const oracle_tables *T, const float c[5])
{
    if (T->tab_axes_kind == 0) return (c[3] > T->tab_max3) || (c[0] > T->tab_max0);
    return (c[3] > T->tab_max3);
}
/* convert_int_sat_rtn */
static inline int convert_int_sat_rtn(float v)
{
    if (v != v) return 0;
    const float f = __builtin_floorf(v);
    if (f >= 2147483648.0f) return 2147483647;
    if (f < -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}
/* Axes.cxx:69-90 with Axis.cxx:45-60 */
static inline uint32_t getBinIndex(const oracle_tables *T, const float c[5])
{
    uint32_t index = 0;
    for (int k = 0; k < T->tab_ndim; ++k) {
        const int pw = T->tab_inverse[k];
        const float v = (pw <= 1) ? c[k] : (pw == 2) ? om_sqrt(c[k]) : (pw == 3) ? om_cbrt(c[k]) : om_pow_frac(c[k], T->tab_inv_exp[k]);
        int b = convert_int_sat_rtn(T->tab_scale[k] * v - T->tab_offset[k]);
        b = imin(imax(b, -1), T->tab_nbins[k]) + 1;
        index += T->tab_stride[k] * (uint32_t)b;
    }
    return index;
}
/* c.cl:228-303 */
static int savePath(const oracle_tables *T, const oracle_step *step, tab_ctx *tc, const float pos0[4], const float dirw[4],
                    float thisStepLength, float *prevStepLength, float inv_groupvel, float depth, float thisStepDepth, int *stop,
                    rng_t *rng)
{
    /* c.cl:246-251 */
    const float impactWeight = (T->tab_ndim > 4) ? step->weight : step->weight * getAngularAcceptance(T, dirw[2]);
    float d = *prevStepLength;
    uint32_t offset = *tc->entry_counter;
    for (; d < thisStepLength && offset < T->tab_entries_per_stream; d += T->tab_volume_step, offset++) {
        float pos[4];
        pos[0] = pos0[0] + d * dirw[0];
        pos[1] = pos0[1] + d * dirw[1];
        pos[2] = pos0[2] + d * dirw[2];
        pos[3] = pos0[3] + d * inv_groupvel;
        float coords[5];
        float dir_copy[4] = { dirw[0], dirw[1], dirw[2], dirw[3] };
        getCoordinates(T, pos, dir_copy, tc->source, rng, coords);
        if (isOutOfBounds(T, coords)) { *stop = 1; break; }
        tc->entries[offset].index = getBinIndex(T, coords);
        tc->entries[offset].weight = impactWeight * om_exp(-(depth + (d / thisStepLength) * thisStepDepth));
    }
    if (d < thisStepLength && !(*stop)) return 0;      /* ran out of space */
    *tc->entry_counter = offset;
    *prevStepLength = d - thisStepLength;
    return 1;
}

/* c.cl:406-913: one work item (tc != NULL: the TABULATE variant, which is built with SAVE_ALL_PHOTONS and a fixed
 * number of absorption lengths, tabulator/I3CLSimStepToTableConverter.cxx:178-186) */
static void propagate_step(const oracle_tables *T, const oracle_step *stepIn, rng_t *rng, hit_sink *sink,
                           uint64_t *iterations, tab_ctx *tc)
{
    oracle_step step = *stepIn;
    float stepDir[4];
    EVENT(OCE_STEPS, 1);
    {
        REGION(OCR_PER_STEP);
        float st, ct, sp, cp;
        om_sincos(step.dir[0], &st, &ct);
        om_sincos(step.dir[1], &sp, &cp);
        const float rho = st;
        stepDir[0] = rho * cp; stepDir[1] = rho * sp; stepDir[2] = ct; stepDir[3] = 0.0f;
    }
    uint32_t photonsLeftToPropagate = step.numPhotons;
    float abs_lens_left = 0.0f, abs_lens_initial = 0.0f;
    float startPos[4] = {0, 0, 0, 0}, startDirw[4] = {0, 0, 0, 0}, pos[4] = {0, 0, 0, 0}, dirw[4] = {0, 0, 0, 0};
    uint32_t numScatters = 0;
    float totalPath = 0.0f;
    int carriedLayer = 0;          /* getTiltZShift_IS_CONSTANT: c.cl:521-523 */
    float inv_groupvel = 0.0f;
    const float thickness = T->layer_thickness;
    const float recip_thickness = 1.0f / thickness;
    uint64_t iters = 0;
    /* c.cl:452-455: private array of the work item, never cleared between photons (the reference leaves it
     * uninitialised; only the entries ConvertPhotonHistories reads, OpenCL.cxx:940-989, are defined) */
    float currentPhotonHistory[ORACLE_MAX_HISTORY][4];
    memset(currentPhotonHistory, 0, sizeof currentPhotonHistory);
    sink->cur_hist = &currentPhotonHistory[0][0];

    rng_t prev_rng = *rng;
    float prevStepRemainder = 0.0f, depthPropagated = 0.0f;
    while (photonsLeftToPropagate > 0) {
        ++iters;
        EVENT(OCE_TRIPS, 1);
#ifdef ORACLE_TRACE
        memset(g_cur, 0, 8);
#endif
        if (abs_lens_left < EPSILON) {
            REGION(OCR_CREATE);
            EVENT(OCE_PHOTONS, 1);
            TR(0, 1);
            prev_rng = *rng;                                                /* c.cl:540-545 */
            createPhotonFromTrack(T, &step, stepDir, rng, pos, dirw);
            if (tc) prevStepRemainder = T->tab_volume_step * rand_oc(rng);  /* c.cl:559-563 */
            memcpy(startPos, pos, 16); memcpy(startDirw, dirw, 16);
            numScatters = 0; totalPath = 0.0f;
            if (!T->has_tilt)
                carriedLayer = imin(imax(findLayerForGivenZPos(T, pos[2]), 0), T->num_layers - 1);
            inv_groupvel = 1.0f / getGroupVelocity(T, dirw[3]);
            if (T->has_fixed_abs) abs_lens_initial = T->fixed_abs;          /* c.cl:582-588 */
            else abs_lens_initial = -om_log(rand_oc(rng));
            abs_lens_left = abs_lens_initial;
            depthPropagated = 0.0f;
        }
        float distancePropagated;
        {
            REGION(OCR_WALK);
            float effective_z; int currentPhotonLayer;
            if (!T->has_tilt) {
                effective_z = pos[2] - T->tilt_const;
                currentPhotonLayer = carriedLayer;
            } else {
                effective_z = pos[2] - getTiltZShift(T, pos);
                currentPhotonLayer = imin(imax(findLayerForGivenZPos(T, effective_z), 0), T->num_layers - 1);
            }
            const float photon_dz = dirw[2];
            const float abs_len_correction_factor = getDirectionalAbsLenCorrFactor(T, dirw);
            abs_lens_left *= abs_len_correction_factor;
            float mediumBoundary = (photon_dz < 0.0f) ? (mediumLayerBoundary(T, currentPhotonLayer))
                                                      : (mediumLayerBoundary(T, currentPhotonLayer) + thickness);
            float sca_step_left = -om_log(rand_oc(rng));
            float currentScaLen = getScatteringLength(T, currentPhotonLayer, dirw[3]);
            float currentAbsLen = getAbsorptionLength(T, currentPhotonLayer, dirw[3]);
            float ais = (photon_dz * sca_step_left - ((mediumBoundary - effective_z) / currentScaLen)) * recip_thickness;
            float aia = (photon_dz * abs_lens_left - ((mediumBoundary - effective_z) / currentAbsLen)) * recip_thickness;
            int j = currentPhotonLayer;
            if (photon_dz < 0) {
                for (; (j > 0) && (ais < 0.0f) && (aia < 0.0f);
                     mediumBoundary -= thickness,
                     currentScaLen = getScatteringLength(T, j, dirw[3]),
                     currentAbsLen = getAbsorptionLength(T, j, dirw[3]),
                     ais += 1.0f / currentScaLen,
                     aia += 1.0f / currentAbsLen) { --j; TR(1, 1); EVENT(OCE_LAYER_CROSSINGS, 1); }
            } else {
                for (; (j < T->num_layers - 1) && (ais > 0.0f) && (aia > 0.0f);
                     mediumBoundary += thickness,
                     currentScaLen = getScatteringLength(T, j, dirw[3]),
                     currentAbsLen = getAbsorptionLength(T, j, dirw[3]),
                     ais -= 1.0f / currentScaLen,
                     aia -= 1.0f / currentAbsLen) { ++j; TR(1, 1); EVENT(OCE_LAYER_CROSSINGS, 1); }
            }
            if (currentPhotonLayer != j) EVENT(OCE_CROSSING_TRIPS, 1);
            float distanceToAbsorption;
            if ((currentPhotonLayer == j) || ((om_fabs(photon_dz)) < EPSILON)) {
                distancePropagated = sca_step_left * currentScaLen;
                distanceToAbsorption = abs_lens_left * currentAbsLen;
            } else {
                const float recip_photon_dz = 1.0f / photon_dz;
                distancePropagated = (ais * thickness * currentScaLen + mediumBoundary - effective_z) * recip_photon_dz;
                distanceToAbsorption = (aia * thickness * currentAbsLen + mediumBoundary - effective_z) * recip_photon_dz;
            }
            if (!T->has_tilt) carriedLayer = j;
            if (distanceToAbsorption < distancePropagated) {
                distancePropagated = distanceToAbsorption;
                abs_lens_left = 0.0f;
            } else {
                abs_lens_left = (distanceToAbsorption - distancePropagated) / currentAbsLen;
            }
            abs_lens_left = abs_lens_left / abs_len_correction_factor;
        }
        if (!tc && !T->stop_detected) {                                     /* c.cl:704-750 without STOP_PHOTONS_ON_DETECTION */
            checkForCollision_keep(T, pos, dirw, inv_groupvel, totalPath, numScatters, abs_lens_initial - abs_lens_left,
                                   startPos, startDirw, &step, distancePropagated, sink);
        } else if (!tc) {
            const int collided = checkForCollision(T, pos, dirw, inv_groupvel, totalPath, numScatters,
                                                   abs_lens_initial - abs_lens_left, startPos, startDirw, &step,
                                                   &distancePropagated, sink);
            if (collided) { abs_lens_left = 0.0f; TR(7, 1); }
        } else {                                                            /* c.cl:755-785 */
            int stop = 0;
            if (!savePath(T, &step, tc, pos, dirw, distancePropagated, &prevStepRemainder, inv_groupvel, depthPropagated,
                          abs_lens_initial - abs_lens_left - depthPropagated, &stop, rng)) {
                tc->photons_left_out = photonsLeftToPropagate;              /* unfinished: restart this photon later */
                *rng = prev_rng;
                if (iterations) *iterations += iters;
                return;
            } else if (stop) {
                abs_lens_left = 0.0f;
            }
            depthPropagated = abs_lens_initial - abs_lens_left;
        }
        REGION(OCR_ADVANCE);                                                /* (to the end of the loop body) */
        pos[0] += dirw[0] * distancePropagated;
        pos[1] += dirw[1] * distancePropagated;
        pos[2] += dirw[2] * distancePropagated;
        pos[3] += inv_groupvel * distancePropagated;
        totalPath += distancePropagated;
        if (abs_lens_left < EPSILON) {
            --photonsLeftToPropagate;
        } else {
            EVENT(OCE_SCATTERS, 1);
            if (T->history_n > 0) {                                         /* c.cl:833-837 */
                float *h = currentPhotonHistory[numScatters % (uint32_t)T->history_n];
                h[0] = pos[0]; h[1] = pos[1]; h[2] = pos[2];
                h[3] = abs_lens_initial - abs_lens_left;
            }
            transformDirection(T->has_pre, T->pre_renorm, T->pre, dirw);
            const float cosScatAngle = makeScatteringCosAngle(T, rng);
            const float sinScatAngle = om_sqrt(1.0f - sqr(cosScatAngle));
            scatterDirectionByAngle(cosScatAngle, sinScatAngle, dirw, rand_co(rng));
            transformDirection(T->has_post, T->post_renorm, T->post, dirw);
            ++numScatters;
            TR(6, 1);
        }
#ifdef ORACLE_TRACE
        if (g_trace && g_trace_n < g_trace_cap) { memcpy(g_trace + 8 * g_trace_n, g_cur, 8); ++g_trace_n; }
#endif
    }
    if (tc) tc->photons_left_out = 0;                                       /* c.cl:905-908 */
    if (iterations) *iterations += iters;
}

/* ---------------- exported entry points (ctypes) ---------------- */
#ifdef __cplusplus
extern "C" {
#endif

/* One launch of the TABULATE kernel over steps[0..n): entries[i*EPS ..] / num_entries[i] per stream, photons_left[i] =
 * what the kernel writes back into inputSteps[i].numPhotons, x[] updated (restored to the photon's start on a miss). */
void oracle_tabulate(const oracle_tables *T, const oracle_step *steps, uint32_t n, uint64_t *x, const uint32_t *a,
                     const oracle_reference *source, oracle_table_entry *entries, uint32_t *num_entries,
                     uint32_t *photons_left, int threads)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (uint32_t i = 0; i < n; ++i) {
        rng_t r = { x[i], a[i] };
        hit_sink sink = { NULL, 0, 0, NULL, NULL };
        num_entries[i] = 0;
        tab_ctx tc = { source, entries + (size_t)i * T->tab_entries_per_stream, &num_entries[i], 0 };
        propagate_step(T, &steps[i], &r, &sink, NULL, &tc);
        photons_left[i] = tc.photons_left_out;
        x[i] = r.x;
    }
}

#ifndef ORACLE_COUNT_OPS     /* (the counting build has no use for it) */
/* The table maker's host loop around the kernel (tabulator/I3CLSimStepToTableConverter.cxx:399-460 FetchSteps, :495-507
 * binContent_[index] += weight), for bunches whose entries would not fit in memory at once: every step is run as the
 * reference re-runs an unfinished one -- `photons_per_call` photons of it per kernel call from the stream as the previous
 * call left it (c.cl:770-776: numPhotons = what is left, stream at the next photon's start) -- into one thread-private entry
 * buffer of TABLE_ENTRIES_PER_STREAM slots, and the call's entries are added to `bins` (binary64, shared: atomic adds;
 * may be NULL), to the step's own sum and count.  A call that runs out of entry space returns -1 at once: the reference
 * would count the interrupted photon's first segments twice (LABBOOK section 8, N3), which no comparison wants.
 * x[] is updated in place.  Returns 0. */
int oracle_tabulate_accumulate(const oracle_tables *T, const oracle_step *steps, uint32_t n, uint64_t *x, const uint32_t *a,
                               const oracle_reference *source, double *bins, double *step_sum, uint64_t *step_entries,
                               uint32_t photons_per_call, int threads)
{
    int failed = 0;
    if (photons_per_call == 0) return -2;
#pragma omp parallel num_threads(threads)
    {
        oracle_table_entry *buf = (oracle_table_entry *)malloc((size_t)T->tab_entries_per_stream * sizeof(oracle_table_entry));
#pragma omp for schedule(dynamic, 1)
        for (uint32_t i = 0; i < n; ++i) {
            int stop;
#pragma omp atomic read
            stop = failed;
            if (stop || !buf) { if (!buf) {
#pragma omp atomic write
                failed = 1;
                } continue; }
            rng_t r = { x[i], a[i] };
            hit_sink sink = { NULL, 0, 0, NULL, NULL };
            oracle_step s = steps[i];
            uint32_t left = steps[i].numPhotons;
            double sum = 0.0;
            uint64_t count = 0;
            while (left > 0) {
                uint32_t filled = 0;
                s.numPhotons = left < photons_per_call ? left : photons_per_call;
                tab_ctx tc = { source, buf, &filled, 0 };
                propagate_step(T, &s, &r, &sink, NULL, &tc);
                if (tc.photons_left_out != 0) {
#pragma omp atomic write
                    failed = 1;
                    break;
                }
                for (uint32_t k = 0; k < filled; ++k) {
                    const double w = (double)buf[k].weight;
                    sum += w;
                    if (bins) {
#pragma omp atomic
                        bins[buf[k].index] += w;
                    }
                }
                count += filled;
                left -= s.numPhotons;
            }
            x[i] = r.x;
            if (step_sum) step_sum[i] = sum;
            if (step_entries) step_entries[i] = count;
        }
        free(buf);
    }
    return failed ? -1 : 0;
}
#endif
void oracle_eval_tabulator(const oracle_tables *T, const oracle_reference *source, const float *pos_and_time, int n,
                           float *coords, uint32_t *index, int32_t *out_of_bounds)
{
    /* the four geometric coordinates; with a fifth axis the caller supplies direction + wavelength and a stream */
    for (int i = 0; i < n; ++i) {
        float c[5], dirw[4] = { 0.f, 0.f, 1.f, 0.f };
        rng_t r = { 1u, 4294967118u };
        oracle_tables T4 = *T;
        T4.tab_ndim = 4;
        getCoordinates(&T4, pos_and_time + 4 * i, dirw, source, &r, c);
        for (int k = 0; k < 4; ++k) coords[4 * i + k] = c[k];
        index[i] = getBinIndex(&T4, c);
        out_of_bounds[i] = isOutOfBounds(&T4, c);
    }
}
float oracle_eval_angular_acceptance(const oracle_tables *T, float x) { return getAngularAcceptance(T, x); }

/* Propagates steps[0..n) with streams (x[i], a[i]); appends hits to `out`
 * (capacity max_hits) in step order.  Returns the hit counter (which may exceed
 * max_hits, like the reference's atomic counter).  x[] is updated in place
 * (c.cl:911-912). */
uint32_t oracle_propagate_hist(const oracle_tables *T, const oracle_step *steps, uint32_t n, uint64_t *x,
                               const uint32_t *a, oracle_photon *out, uint32_t max_hits, uint64_t *iterations,
                               float *hist_out)
{
    if (T->history_n > ORACLE_MAX_HISTORY) return 0xffffffffu;
    hit_sink sink = { out, max_hits, 0, hist_out, NULL };
    uint64_t it = 0;
    for (uint32_t i = 0; i < n; ++i) {
        rng_t r = { x[i], a[i] };
        propagate_step(T, &steps[i], &r, &sink, &it, NULL);
        x[i] = r.x;
    }
    if (iterations) *iterations = it;
    COUNT_FOLD();
    return sink.count;
}
uint32_t oracle_propagate(const oracle_tables *T, const oracle_step *steps, uint32_t n, uint64_t *x,
                          const uint32_t *a, oracle_photon *out, uint32_t max_hits, uint64_t *iterations)
{
    return oracle_propagate_hist(T, steps, n, x, a, out, max_hits, iterations, NULL);
}

/* Multi-threaded variant used as the CPU baseline (bench.py cpu_baseline): one
 * step per task; hits are concatenated per thread, so their order differs from
 * oracle_propagate (the reference's order is not deterministic either, H3). */
uint32_t oracle_propagate_mt(const oracle_tables *T, const oracle_step *steps, uint32_t n, uint64_t *x,
                             const uint32_t *a, oracle_photon *out, uint32_t max_hits, int threads,
                             uint64_t *iterations)
{
#ifdef _OPENMP
    uint32_t total = 0;
    uint64_t it_total = 0;
#pragma omp parallel num_threads(threads)
    {
        uint32_t cap = 1024, cnt = 0;
        oracle_photon *buf = (oracle_photon *)malloc((size_t)cap * sizeof(oracle_photon));
        uint64_t it = 0;
#pragma omp for schedule(dynamic, 16)
        for (uint32_t i = 0; i < n; ++i) {
            /* a step of P photons records at most P hits -- with STOP_PHOTONS_ON_DETECTION; without it a photon is recorded
             * by every DOM on its way, so the step is run again with more room if its records did not fit */
            uint32_t need = cnt + steps[i].numPhotons;
            for (;;) {
                if (need > cap) {
                    cap = need * 2 + 4096;
                    buf = (oracle_photon *)realloc(buf, (size_t)cap * sizeof(oracle_photon));
                }
                rng_t r = { x[i], a[i] };
                uint64_t it_step = 0;
                hit_sink sink = { buf + cnt, cap - cnt, 0, NULL, NULL };
                propagate_step(T, &steps[i], &r, &sink, &it_step, NULL);
                if (sink.count > cap - cnt) { need = cnt + sink.count; continue; }
                cnt += sink.count;
                it += it_step;
                x[i] = r.x;
                break;
            }
        }
#pragma omp critical
        {
            for (uint32_t k = 0; k < cnt; ++k) {
                if (total < max_hits) out[total] = buf[k];
                ++total;
            }
            it_total += it;
            COUNT_FOLD();
        }
        free(buf);
    }
    if (iterations) *iterations = it_total;
    return total;
#else
    (void)threads;
    return oracle_propagate(T, steps, n, x, a, out, max_hits, iterations);
#endif
}

/* ---- sub-function probes for the unit tests ---- */
void oracle_eval_medium(const oracle_tables *T, int what, const float *in, int n, int layer, float *out)
{
    for (int i = 0; i < n; ++i) {
        switch (what) {
        case 0: out[i] = getAbsorptionLength(T, layer, in[i]); break;
        case 1: out[i] = getScatteringLength(T, layer, in[i]); break;
        case 2: out[i] = getPhaseRefIndex(T, in[i]); break;
        case 3: out[i] = getGroupVelocity(T, in[i]); break;
        case 4: out[i] = getWavelengthBias(T, in[i]); break;
        default: out[i] = 0.0f;
        }
    }
}
void oracle_eval_field(const oracle_tables *T, int what, const float *xyz, int n, float *out)
{
    for (int i = 0; i < n; ++i) {
        float v[4] = { xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.0f };
        switch (what) {
        case 0: out[i] = getTiltZShift(T, v); break;
        case 1: out[i] = getDirectionalAbsLenCorrFactor(T, v); break;
        case 2: transformDirection(T->has_pre, T->pre_renorm, T->pre, v);
                out[3 * i] = v[0]; out[3 * i + 1] = v[1]; out[3 * i + 2] = v[2]; break;
        case 3: transformDirection(T->has_post, T->post_renorm, T->post, v);
                out[3 * i] = v[0]; out[3 * i + 1] = v[1]; out[3 * i + 2] = v[2]; break;
        default: out[i] = 0.0f;
        }
    }
}
void oracle_eval_rng(uint64_t *x, uint32_t a, int n, float *out_co)
{
    rng_t r = { *x, a };
    for (int i = 0; i < n; ++i) out_co[i] = rand_co(&r);
    *x = r.x;
}
void oracle_eval_wlen(const oracle_tables *T, int kgen, uint64_t *x, uint32_t a, int n, float *out)
{
    rng_t r = { *x, a };
    for (int i = 0; i < n; ++i) out[i] = generateWavelength_k(T, kgen, &r);
    *x = r.x;
}
void oracle_eval_scatcos(const oracle_tables *T, uint64_t *x, uint32_t a, int n, float *out)
{
    rng_t r = { *x, a };
    for (int i = 0; i < n; ++i) out[i] = makeScatteringCosAngle(T, &r);
    *x = r.x;
}
/* math probes: what = 0 log,1 exp,2 sin,3 cos,4 powr(x,y),5 acos,6 atan2(x,y),7 rsqrt,8 sqrt,9 div(x,y) */
void oracle_eval_math(int what, const float *xs, const float *ys, int n, float *out)
{
    for (int i = 0; i < n; ++i) {
        const float x = xs[i], y = ys ? ys[i] : (float)0.0f;
        switch (what) {
        case 0: out[i] = om_log(x); break;
        case 1: out[i] = om_exp(x); break;
        case 2: out[i] = om_sin(x); break;
        case 3: out[i] = om_cos(x); break;
        case 4: out[i] = om_powr(x, y); break;
        case 5: out[i] = om_acos(x); break;
        case 6: out[i] = om_atan2(x, y); break;
        case 7: out[i] = om_rsqrt(x); break;
        case 8: out[i] = om_sqrt(x); break;
        case 9: out[i] = x / y; break;
        case 10: out[i] = om_acos_f(x); break;
        case 14: out[i] = om_powr_unit(x, y); break;
        case 15: out[i] = om_cbrt(x); break;
        default: out[i] = 0.0f;
        }
    }
}
size_t oracle_sizeof_tables(void) { return sizeof(oracle_tables); }

#ifdef ORACLE_TRACE
/* traces ONE step: returns the number of loop iterations written (8 bytes each) */
uint64_t oracle_trace_step(const oracle_tables *T, const oracle_step *step, uint64_t x, uint32_t a, uint8_t *buf, uint64_t cap)
{
    static oracle_photon scratch[4096];
    hit_sink sink = { scratch, 4096, 0 };
    rng_t r = { x, a };
    g_trace = buf; g_trace_cap = cap; g_trace_n = 0;
    propagate_step(T, step, &r, &sink, 0, NULL);
    g_trace = 0;
    return g_trace_n;
}
#endif

#ifdef __cplusplus
}
#endif
#ifdef ORACLE_COUNT_OPS
/* counting build: totals over all threads since the last reset (the multi-threaded driver above folds each thread's counters
 * in when its loop ends; the serial entry points fold the caller's) */
#include "count_ops_api.hpp"
#endif
