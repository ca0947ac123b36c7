/*
 * oracle_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Deterministic single-precision math used by the CPU restatement of clsim's
 * propagation kernel.  The reference kernel calls the OpenCL runtime builtins
 * (non-native branch of resources/kernels/propagation_kernel.c.cl:52-60:
 * a/b, 1.f/a, powr, sqrt, rsqrt, cos, sin, log, exp; plus acos/atan2 in
 * sphDirFromCar, c.cl:206-223).  Those builtins are a vendor library that is
 * not part of /root/reference and has no pinned bit pattern (OpenCL only bounds
 * them: <=3-4 ulp for log/exp/sin/cos, <=16 ulp for powr, <=2 ulp rsqrt).
 * This header fixes ONE bit pattern for each of them, built only from IEEE-754
 * correctly rounded operations (+ - * / sqrt fma, int<->float conversions), so
 * that an x86-64 build (gcc -ffp-contract=off -mfma) and a gfx950 build of the
 * same operation sequence agree bit for bit.  The HIP product carries its own
 * implementation of the same sequences (clsim_amd/csrc/detmath.hip.h);
 * tests/test_detmath_gpu.py checks the two against each other on the GPU.
 *
 * Polynomial coefficients: the classic Cephes single precision sets
 * (S. Moshier, logf/expf/sinf/cosf), accuracy measured by oracle/mathcheck.c.
 */
#ifndef CLSIM_ORACLE_MATH_H
#define CLSIM_ORACLE_MATH_H

#include <stdint.h>
#include <string.h>

#define OM_INLINE static inline __attribute__((always_inline))

OM_INLINE uint32_t om_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
OM_INLINE float om_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
OM_INLINE uint64_t om_d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
OM_INLINE double om_u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

OM_INLINE float om_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
OM_INLINE float om_sqrt(float a) { return __builtin_sqrtf(a); }   /* IEEE */
OM_INLINE float om_div(float a, float b) { return a / b; }         /* IEEE */
OM_INLINE float om_recip(float a) { return 1.0f / a; }             /* c.cl:53 */
/* OpenCL rsqrt (<=2 ulp): fixed here as two correctly rounded ops. */
OM_INLINE float om_rsqrt(float a) { return 1.0f / __builtin_sqrtf(a); }
OM_INLINE float om_fabs(float a) { return __builtin_fabsf(a); }
OM_INLINE float om_rint(float a) { return __builtin_rintf(a); }    /* nearest-even */

#define OM_LN2_HI 0.693359375f            /* 355/512, 9 significant bits   */
#define OM_LN2_LO (-2.12194440e-4f)
#define OM_LOG2E 1.44269504088896341f

/* log1p(r)-r+r*r/2 = r^3*P(r) on r in [sqrt(.5)-1, sqrt(2)-1] (Cephes logf). */
OM_INLINE float om_log_poly(float r)
{
    float p = 7.0376836292e-2f;
    p = om_fma(p, r, -1.1514610310e-1f);
    p = om_fma(p, r, 1.1676998740e-1f);
    p = om_fma(p, r, -1.2420140846e-1f);
    p = om_fma(p, r, 1.4249322787e-1f);
    p = om_fma(p, r, -1.6668057665e-1f);
    p = om_fma(p, r, 2.0000714765e-1f);
    p = om_fma(p, r, -2.4999993993e-1f);
    p = om_fma(p, r, 3.3333331174e-1f);
    return p;
}

/* Split a positive normal float into 2^e * m with m in [sqrt(.5), sqrt(2)). */
OM_INLINE float om_frexp_sqrt2(float x, int *e)
{
    const uint32_t ix = om_f2u(x);
    const int32_t d = (int32_t)(ix - 0x3f3504f3u);    /* bits(sqrt(.5)) */
    const int32_t ee = d >> 23;                        /* arithmetic shift */
    *e = ee;
    return om_u2f(ix - ((uint32_t)ee << 23));
}

/* Natural log for positive normal x.  x<=0 / subnormal are outside the
 * kernel's domain (arguments are (0,1] uniforms >= 2^-24). */
OM_INLINE float om_log(float x)
{
    int e;
    const float m = om_frexp_sqrt2(x, &e);
    const float r = m - 1.0f;                    /* exact */
    const float fe = (float)e;
    const float z = r * r;
    float y = (r * z) * om_log_poly(r);
    y = om_fma(fe, OM_LN2_LO, y);
    y = om_fma(-0.5f, z, y);
    return om_fma(fe, OM_LN2_HI, r + y);
}

/* exp(r)-1-r = r*r*Q(r), |r| <= ln2/2 (Cephes expf). */
OM_INLINE float om_exp_poly(float r)
{
    float p = 1.9875691500e-4f;
    p = om_fma(p, r, 1.3981999507e-3f);
    p = om_fma(p, r, 8.3334519073e-3f);
    p = om_fma(p, r, 4.1665795894e-2f);
    p = om_fma(p, r, 1.6666665459e-1f);
    p = om_fma(p, r, 5.0000001201e-1f);
    return p;
}

/* exp(hi+lo) with |lo| << |hi|; result flushed to 0 below 2^-126, inf above. */
OM_INLINE float om_exp_hl(float hi, float lo)
{
    if (hi < -86.0f) return 0.0f;
    if (hi > 88.0f) return om_u2f(0x7f800000u);
    const float k = om_rint(hi * OM_LOG2E);
    float r = om_fma(-k, OM_LN2_HI, hi);          /* exact */
    r = om_fma(-k, OM_LN2_LO, r);
    r = r + lo;
    const float z = r * r;
    float p = om_fma(z, om_exp_poly(r), r);
    p = p + 1.0f;
    const int32_t ik = (int32_t)k;
    return om_u2f(om_f2u(p) + ((uint32_t)ik << 23));
}

OM_INLINE float om_exp(float x) { return om_exp_hl(x, 0.0f); }

/* powr(x,y), x>=0 (OpenCL powr; used by the ice functions
 * I3CLSimHelperGenerateMediumPropertiesSource_Optimizers.cxx:180,240 and by
 * I3CLSimRandomValueSimplifiedLiu.cxx:84).  log(x) is kept as an unevaluated
 * sum hi+lo so that y*log(x) keeps ~2^-30 relative accuracy. */
OM_INLINE float om_powr(float x, float y)
{
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : om_u2f(0x7f800000u));
    int e;
    const float m = om_frexp_sqrt2(x, &e);
    const float r = m - 1.0f;
    const float fe = (float)e;
    const float z = r * r;
    float c = (r * z) * om_log_poly(r);
    c = om_fma(fe, OM_LN2_LO, c);
    c = om_fma(-0.5f, z, c);
    /* hi + err = fe*LN2_HI + r exactly (two-sum; fe*LN2_HI is exact) */
    const float t = fe * OM_LN2_HI;
    const float hi = t + r;
    const float bb = hi - t;
    const float err = (t - (hi - bb)) + (r - bb);
    const float lo = err + c;
    /* y*(hi+lo) as ph+pl */
    const float ph = y * hi;
    const float pe = om_fma(y, hi, -ph);
    const float pl = om_fma(y, lo, pe);
    return om_exp_hl(ph, pl);
}

/* pi/2 = C1+C2+C3 (+1e-23), each term a full binary32; the products k*Ci are
 * exact inside the fma. */
#define OM_PIO2_1 0x1.921fb6p+0f
#define OM_PIO2_2 (-0x1.777a5cp-25f)
#define OM_PIO2_3 (-0x1.ee59dap-50f)
#define OM_2OPI 0.636619772367581343f

OM_INLINE float om_sin_poly(float r, float z)
{
    float p = -1.9515295891e-4f;
    p = om_fma(p, z, 8.3321608736e-3f);
    p = om_fma(p, z, -1.6666654611e-1f);
    return om_fma(p * z, r, r);
}
OM_INLINE float om_cos_poly(float z)
{
    float p = 2.443315711809948e-5f;
    p = om_fma(p, z, -1.388731625493765e-3f);
    p = om_fma(p, z, 4.166664568298827e-2f);
    p = p * (z * z);
    p = om_fma(-0.5f, z, p);
    return p + 1.0f;
}

/* sin and cos of the same argument.  Reduction is accurate for |x| < ~1e4
 * (arguments in the kernel: 2*pi*u, step theta/phi). */
OM_INLINE void om_sincos(float x, float *s, float *c)
{
    const float k = om_rint(x * OM_2OPI);
    float r = om_fma(-k, OM_PIO2_1, x);
    r = om_fma(-k, OM_PIO2_2, r);
    r = om_fma(-k, OM_PIO2_3, r);
    const float z = r * r;
    const float ps = om_sin_poly(r, z);
    const float pc = om_cos_poly(z);
    const int32_t q = (int32_t)k;
    const float a = (q & 1) ? pc : ps;   /* |sin| source */
    const float b = (q & 1) ? ps : pc;   /* |cos| source */
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
}
OM_INLINE float om_sin(float x) { float s, c; om_sincos(x, &s, &c); return s; }
OM_INLINE float om_cos(float x) { float s, c; om_sincos(x, &s, &c); return c; }

/* ---- rare path (one call per recorded hit): evaluated in binary64 -------- */

#define OM_PI_D 3.14159265358979323846
/* atan on |t| <= tan(pi/8) : odd Taylor series to t^25 */
OM_INLINE double om_atan_small_d(double t)
{
    const double z = t * t;
    double p = 1.0 / 25.0;
    p = __builtin_fma(p, -z, 1.0 / 23.0);
    p = __builtin_fma(p, -z, 1.0 / 21.0);
    p = __builtin_fma(p, -z, 1.0 / 19.0);
    p = __builtin_fma(p, -z, 1.0 / 17.0);
    p = __builtin_fma(p, -z, 1.0 / 15.0);
    p = __builtin_fma(p, -z, 1.0 / 13.0);
    p = __builtin_fma(p, -z, 1.0 / 11.0);
    p = __builtin_fma(p, -z, 1.0 / 9.0);
    p = __builtin_fma(p, -z, 1.0 / 7.0);
    p = __builtin_fma(p, -z, 1.0 / 5.0);
    p = __builtin_fma(p, -z, 1.0 / 3.0);
    p = __builtin_fma(p, -z, 1.0);
    return t * p;
}
OM_INLINE double om_atan2_d(double y, double x)
{
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double mx = (ax > ay) ? ax : ay;
    const double mn = (ax > ay) ? ay : ax;
    double a;
    if (mx == 0.0) {
        a = 0.0;
    } else {
        double t = mn / mx;                     /* [0,1] */
        double off = 0.0;
        if (t > 0.41421356237309503) {          /* tan(pi/8) */
            t = (t - 1.0) / (t + 1.0);
            off = 0.78539816339744828;          /* pi/4 */
        }
        a = off + om_atan_small_d(t);
    }
    if (ay > ax) a = 1.57079632679489656 - a;   /* pi/2 - a */
    if (x < 0.0) a = OM_PI_D - a;
    if (y < 0.0) a = -a;
    return a;
}
OM_INLINE float om_atan2(float y, float x) { return (float)om_atan2_d((double)y, (double)x); }
/* acos for |v|<=1 */
OM_INLINE float om_acos(float v)
{
    const double d = (double)v;
    const double s = __builtin_sqrt((1.0 - d) * (1.0 + d));
    return (float)om_atan2_d(s, d);
}

/* single precision arccosine of the table maker's path samples (detmath.hip.h: acos_f): Cephes asinf polynomial */
OM_INLINE float om_acos_f(float x)
{
    const float ax = __builtin_fabsf(x);
    if (!(ax <= 1.0f)) return om_u2f(0x7fc00000u);
    const int big = ax > 0.5f;
    const float z = big ? 0.5f * (1.0f - ax) : ax * ax;
    const float s = big ? om_sqrt(z) : ax;
    const float p = (((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f;
    const float a = s + (s * z) * p;
    if (big) return (x < 0.0f) ? (3.14159265358979f - 2.0f * a) : (2.0f * a);
    return (x < 0.0f) ? (1.5707963267948966f + a) : (1.5707963267948966f - a);
}

/* powr(x, y) for x in [0, 1] and 0 < y with y * |log x| <= 2: the scattering angle of the simplified Liu function,
 * 2 u^beta - 1 with beta = (1 - g)/(1 + g) ~ 0.05 (I3CLSimRandomValueSimplifiedLiu.cxx:84) -- once per scatter, where the
 * hi+lo logarithm of om_powr is not needed: |y log x| <= 1.2 for u >= 2^-32, so the rounding of log x (0.82 ulp) and of
 * the product enter exp with an absolute error below 2^-23 and the result stays within 3 ulp (mathcheck.c: 1.94 ulp at
 * beta = 0.0526, 2.84 at 0.09; OpenCL allows powr 16).  x = 0 gives 0. */
OM_INLINE float om_powr_unit(float x, float y)
{
    if (x == 0.0f) return 0.0f;
    return om_exp(y * om_log(x));
}

/* cbrt(x) and pow(x, y) for the table maker's power axes (tabulator/Axis.cxx:150-171: inverse transform cbrt(x) for power
 * 3, pow(x, 1/power) above): cbrt through the hi+lo logarithm with the exponent RN(1/3), corrected for the exponent's
 * second word (1e-8 |ln x| relative otherwise): <= 1.5 ulp (OpenCL: cbrt 2 ulp), odd symmetry as the builtin; pow of
 * a negative base with a non-integer exponent is NaN as in OpenCL. */
OM_INLINE float om_cbrt(float x)
{
    const float ax = x < 0.0f ? -x : x;
    if (ax == 0.0f) return x;
    float r = om_powr(ax, 0.333333343f);
    r = om_fma(r, -9.934107e-09f * om_log(ax), r);          /* the exponent's second word: 1/3 - RN(1/3) */
    return x < 0.0f ? -r : r;
}
OM_INLINE float om_pow_frac(float x, float y) { return (x < 0.0f) ? om_u2f(0x7fc00000u) : om_powr(x, y); }

/* ---- ORACLE_LIBM (analysis build, tools/math_sensitivity.py): a DIFFERENT conforming math library -------------------
 * The deterministic functions above are this repository's definition of the OpenCL builtins; the reference kernel
 * runs on whatever the OpenCL runtime provides (a few ulp, unpinned).  With -DORACLE_LIBM the kernel-facing names are
 * mapped to glibc's single precision functions instead, so that the physics effect of "another conforming math
 * library" can be measured by comparing two runs of the same restatement.  Never used by the parity tests. */
#ifdef ORACLE_LIBM
#include <math.h>
OM_INLINE float om_libm_log(float x) { return logf(x); }
OM_INLINE float om_libm_exp(float x) { return expf(x); }
OM_INLINE float om_libm_powr(float x, float y) { return powf(x, y); }
OM_INLINE void om_libm_sincos(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
OM_INLINE float om_libm_sin(float x) { return sinf(x); }
OM_INLINE float om_libm_cos(float x) { return cosf(x); }
OM_INLINE float om_libm_atan2(float y, float x) { return atan2f(y, x); }
OM_INLINE float om_libm_acos(float v) { return acosf(v); }
#define om_log om_libm_log
#define om_exp om_libm_exp
#define om_powr om_libm_powr
#define om_powr_unit om_libm_powr
#define om_cbrt cbrtf
#define om_pow_frac powf
#define om_sincos om_libm_sincos
#define om_sin om_libm_sin
#define om_cos om_libm_cos
#define om_atan2 om_libm_atan2
#define om_acos om_libm_acos
#endif

#endif
