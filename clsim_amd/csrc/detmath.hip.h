// Device math for the propagator: single precision log/exp/powr/sin/cos (and
// acos/atan2 for the hit record) built only from IEEE-754 correctly rounded
// operations -- v_fma_f32, v_mul/add_f32, IEEE divide and sqrt (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt), v_rndne, integer ops -- never from
// the approximate v_log/v_exp/v_sin/v_cos units, whose results cannot be
// reproduced bit for bit off the GPU.  The reference calls the OpenCL runtime's
// builtins here (propagation_kernel.c.cl:52-60), which have no pinned bit
// pattern; this file defines one.  Compile with -ffp-contract=off: every fused
// multiply-add below is explicit.
//
// Polynomials: Cephes single precision (logf, expf, sinf, cosf), Horner form.
// powr keeps log(x) as an unevaluated hi+lo pair so that |y*log x| ~ 10 (the
// lambda^-kappa of the ice model) still rounds within ~1.2 ulp.
//
// Round 5: log, and sin / cos of arguments in [0, RN(2 pi)], are the table forms of oracle/oracle_math.h (om_log,
// om_sincos_2pi) -- constants in math_tables.h (tools/make_math_tables.py writes the same text for both sides): 16 and
// 17 vector instructions in place of 22 and 31.  The propagation kernels read the two tables from the front of their LDS
// image (prop_device.hip.h: lds_log / lds_sincos_2pi; kMathTableWords), every other kernel from the global copies below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "math_tables.h"

#define DM __device__ __forceinline__

namespace dm {

DM float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DM float sqrt_(float a) { return __builtin_sqrtf(a); }
DM float rsqrt_(float a) { return 1.0f / __builtin_sqrtf(a); }
DM float rint_(float a) { return __builtin_rintf(a); }

DM uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
DM float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// ---- correctly rounded 1/x and sqrt(x) for arguments in a known range ----
// hipcc's IEEE sequences (-fhip-fp32-correctly-rounded-divide-sqrt) carry range scaling (v_div_scale x2, or a scale /
// unscale pair around v_sqrt), a second quotient refinement and a special-case fix-up: 11 instructions per divide, 16 per
// square root.  Where the argument is known to be finite and far from the ends of the exponent range the same correctly
// rounded result takes fewer:
//   rcp_: v_rcp_f32 (<= 1 ulp) + one Newton step in fma arithmetic.  The step's exact value is (1/x)(1 - d^2) with
//       d <= 2^-23 the relative error of v_rcp_f32, so it rounds to RN(1/x) unless 1/x lies within 2^-46 of a rounding
//       boundary, which only a handful of significands can (the classic one: all ones).  Whether any of them is missed
//       depends on the hardware's v_rcp_f32, so it is not argued, it is TESTED: clsimhip_check_math_exhaustive runs all
//       2^23 significands x every exponent in [-100, 100] x both signs on the device against the IEEE divide
//       (tests/test_detmath_gpu.py): gfx950 misses none.  |x| in [2^-100, 2^100]; garbage (never a trap) outside.
//   sqrt_near_: v_sqrt_f32 (<= 1 ulp) + the compiler's own +-1 ulp residual selection, without the scaling (arguments
//       >= 2^-96 need none) and without the class fix-up for inf / NaN (finite arguments).  Zero stays zero.
//       x = 0 or x in [2^-96, 2^100]; same exhaustive test.
//   rsqrt_near_: their composition (the reference's rsqrt is 1/sqrt here: oracle_math.h).
// Results are RN(1/x) and RN(sqrt x): the x86 side computes them with the IEEE divide / sqrt.
DM float rcp_(float x)
{
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e0 = fma_(-x, r0, 1.0f);
    return fma_(e0, r0, r0);
}
// rcp_of_rcp_(b, x) = RN(1/b) for b = RN(1/x) -- the reciprocal of a reciprocal whose argument is still at hand (round 4).  The layer walk
// forms a length as RN(1/x) (x = b400 * lambda^-alpha, ...) and then divides by that length, i.e. needs RN(1/RN(1/x)) -- which is x itself
// or a neighbour.  With x as the seed one Newton step in fma arithmetic gives it exactly: for EVERY significand of x,
// fma(fma(-b, x, 1), x, x) == RN(1/b) (the residual 1 - b x is below 2^-23 and exact in the fma; the step's value is (1/b)(1 - r^2) with
// r^2 <= 2^-46, and no significand puts it within that of a rounding boundary).  Scaling x by a power of two scales every intermediate
// exactly, so one binade proves all -- away from the ends of the exponent range: |x| in [2^-100, 2^100].  Not argued, TESTED: all 2^23
// significands on the host when the idea was tried, and on the device against the IEEE divide, every exponent of the range, both signs
// (clsimhip_check_math_exhaustive(17), tests/test_detmath_gpu.py).  Two full-rate instructions in place of v_rcp_f32 (quarter rate) + two.
DM float rcp_of_rcp_(float b, float x)
{
    const float e = fma_(-b, x, 1.0f);
    return fma_(e, x, x);
}
// div_near_(a, b) = RN(a / b) for 2^-50 <= |b| <= 2^50 and 2^-40 <= |a| <= 2^60, in 8 instructions without the IEEE
// sequence's VCC-carried scaling (11, and stalls: measured +3.4 % for four call sites).  Markstein's scheme on the exact
// reciprocal: y = RN(1/b) (rcp_ above, tested exhaustively); q0 = RN(a y) is within 2^-23 |a/b|; one correction
// q1 = RN(q0 + RN(a - b q0) y) lands within 2^-46 |a/b| of the quotient before its rounding, so q1 is a faithful
// rounding of a/b; then r1 = a - b q1 is exact in an fma and q2 = RN(q1 + r1 y) = RN(a/b) by Markstein's theorem (IBM J.
// Res. Dev. 34 (1990); Muller et al., Handbook of Floating-Point Arithmetic, ch. 5: y within 2^-24 relative of 1/b, q1
// faithful, no underflow or overflow).  The operand ranges keep every intermediate normal: |q| in [2^-90, 2^110], and a
// non-zero residual is at least 2^-46 |a| >= 2^-86.  NOT valid for a = -0 (gives +0), for smaller |a| (residuals may
// underflow) or beyond: callers test div_near_ok_ -- wave-wide, the IEEE divide for everyone otherwise -- or hold a proof.
// clsimhip_check_math_exhaustive(16) runs all 2^23 divisor significands x divisor exponents x 40 numerators (random
// and adversarial) against the IEEE divide on the device (tests/test_detmath_gpu.py).
DM bool div_near_ok_(float a) { return __builtin_fabsf(a) >= 9.094947017729282e-13f; }      // 2^-40; false for NaN
DM float div_near_with_(float a, float b, float y)      // y = rcp_(b): shared by numerators over one divisor
{
    const float q0 = a * y;
    const float r0 = fma_(-b, q0, a);
    const float q1 = fma_(r0, y, q0);
    const float r1 = fma_(-b, q1, a);
    return fma_(r1, y, q1);
}
DM float div_near_(float a, float b) { return div_near_with_(a, b, rcp_(b)); }
DM float sqrt_near_(float x)
{
    // s is within an ulp of the root; the residuals of its two neighbours say which of the three is nearest: x - s_dn s <= 0
    // means s is too large (take s_dn), x - s_up s > 0 that it is too small (take s_up).  The two answers are taken from the
    // residuals' bit patterns -- positive and non-zero <=> positive as an integer -- and added to s_dn's pattern: two
    // v_med3_i32 and a three-operand add where the compare / select pairs went through VCC with wait states between them.
    // (x = 0: s = 0, the saturating subtraction keeps s_dn at 0, both residuals are +0 and the result is 0.)
    const float s = __builtin_amdgcn_sqrtf(x);
    const uint32_t dn = __builtin_elementwise_sub_sat(f2u(s), 1u);
    const float s_dn = u2f(dn), s_up = u2f(f2u(s) + 1u);
    const float r_dn = fma_(-s_dn, s, x);
    const float r_up = fma_(-s_up, s, x);
    uint32_t m_dn, m_up;                                             // 1 if the residual is > 0, else 0
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(m_dn) : "v"(r_dn));         // (asm: the compiler turns the clamp back into compare + carry)
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(m_up) : "v"(r_up));
    return u2f(dn + m_dn + m_up);
}
DM float rsqrt_near_(float x) { return rcp_(sqrt_near_(x)); }
// rsqrt_unit_(x) = RN(1 / RN(sqrt x)) for x within 1023 ulps of one (rsqrt_unit_ok_) -- the renormalisation of a vector that a rotation has
// just left within rounding errors of unit length (round 4).  So close to one both roundings are decided by the argument's distance from
// one in ulps: x = 1 + k 2^-23: sqrt x = 1 + (k/2) 2^-23 - k^2 2^-49 rounds to 1 + floor(k/2) 2^-23 =: 1 + m 2^-23, and 1 / that =
// 1 - m 2^-23 + m^2 2^-46 rounds to 1 - m 2^-23, which is 2m ulps below one (ulps below one are half as wide); x = 1 - j 2^-24: the root
// rounds to 1 - ceil(j/2) 2^-24 =: 1 - c 2^-24, its reciprocal 1 + (c/2) 2^-23 + ... to 1 + ceil(c/2) 2^-23.  Integer arithmetic on the bit
// pattern, full rate, in place of v_sqrt_f32 + 8 + v_rcp_f32 + 2.  The argument holds while the squared terms stay below a quarter ulp (k
// below 2896); not trusted, TESTED: every pattern of the window on the host, and on the device against the IEEE operations
// (clsimhip_check_math_exhaustive(18), tests/test_detmath_gpu.py).
DM bool rsqrt_unit_ok_(float x) { return (f2u(x) - (0x3f800000u - 1023u)) <= 2046u; }
DM float rsqrt_unit_(float x)
{
    const int d = (int)(f2u(x) - 0x3f800000u);
    const uint32_t above = 0x3f800000u - ((uint32_t)d & ~1u);          // d >= 0
    const uint32_t below = 0x3f800000u + ((uint32_t)(3 - d) >> 2);     // d < 0: ceil(ceil(j / 2) / 2) = (j + 3) >> 2, j = -d
    return u2f((d < 0) ? below : above);
}

// the tables: words [0, 128) of an LDS image = 32 log rows {INV, H, L, 0}; words [128, 194) = 33 sincos rows {S, C}; padded to 196
constexpr uint32_t kMathLogWords = 4u * MT_LOG_ROWS, kMathScWords = 2u * MT_SC_ROWS, kMathTableWords = (kMathLogWords + kMathScWords + 3u) & ~3u;
static __device__ const float kLogTable[kMathLogWords] = MT_LOG_TABLE;
static __device__ const float kScTable[kMathScWords] = MT_SC_TABLE;
static const float kLogTableHost[kMathLogWords] = MT_LOG_TABLE;       // (what tables.cpp puts at the front of the LDS image)
static const float kScTableHost[kMathScWords] = MT_SC_TABLE;

constexpr float LN2_HI = 0.693359375f;
constexpr float LN2_LO = -2.12194440e-4f;
constexpr float LOG2E = 1.44269504088896341f;

DM float log_poly(float r)
{
    float p = 7.0376836292e-2f;
    p = fma_(p, r, -1.1514610310e-1f);
    p = fma_(p, r, 1.1676998740e-1f);
    p = fma_(p, r, -1.2420140846e-1f);
    p = fma_(p, r, 1.4249322787e-1f);
    p = fma_(p, r, -1.6668057665e-1f);
    p = fma_(p, r, 2.0000714765e-1f);
    p = fma_(p, r, -2.4999993993e-1f);
    p = fma_(p, r, 3.3333331174e-1f);
    return p;
}

// x = 2^e * m, m in [sqrt(1/2), sqrt(2))
DM float frexp_sqrt2(float x, int &e)
{
    const uint32_t ix = f2u(x);
    const int32_t d = (int32_t)(ix - 0x3f3504f3u);
    e = d >> 23;
    return u2f(ix - ((uint32_t)e << 23));
}

// A table in LDS is named by its BYTE ADDRESS (LdsTable{address}): the row load is then a ds_read whose address register holds
// nothing but the row's offset -- through the symbol of the dynamic LDS array the compiler adds that symbol's address (0) with an
// instruction of its own.  (In LDS address 0 is an ordinary address; the null pointer of that address space is all ones.)
struct LdsTable { uint32_t address; };
template <class Row> DM Row table_row_(const float *table, uint32_t byte_offset)
{
    return *reinterpret_cast<const Row *>(reinterpret_cast<const char *>(table) + byte_offset);
}
template <class Row> DM Row table_row_(LdsTable table, uint32_t byte_offset)
{
    typedef const Row __attribute__((address_space(3))) *lds_row_ptr;
    return *reinterpret_cast<lds_row_ptr>(static_cast<uintptr_t>(table.address + byte_offset));
}

// oracle_math.h: om_log.  `table`: the 32 rows, in LDS (a ds_read_b128 at an immediate offset) or in global memory.
template <class Table>
DM float log_with_(float x, Table table)
{
    typedef float row_t __attribute__((ext_vector_type(4)));
    const uint32_t ix = f2u(x);
    const float m = u2f((ix & 0x007fffffu) | 0x3f800000u);
    const float fe = (float)__builtin_amdgcn_frexp_expf(x);             // e + 1 (x positive and normal)
    const row_t row = table_row_<row_t>(table, (ix >> 14) & 0x1f0u);
    const float r = fma_(m, row.x, -1.0f);
    const float z = r * r;
    float p = fma_(r, MT_LOG_P5, MT_LOG_P4);
    p = fma_(r, p, MT_LOG_P3);
    p = fma_(r, p, -0.5f);
    const float t = fma_(z, p, r);
    const float big = fma_(fe, MT_LN2_HI, row.y);
    const float small = t + fma_(fe, MT_LN2_LO, row.z);
    return big + small;
}
DM float log_(float x) { return log_with_<const float *>(x, kLogTable); }

DM float exp_poly(float r)
{
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    return p;
}

DM float exp_hl(float hi, float lo)
{
    if (hi < -86.0f) return 0.0f;
    if (hi > 88.0f) return u2f(0x7f800000u);
    const float k = rint_(hi * LOG2E);
    float r = fma_(-k, LN2_HI, hi);
    r = fma_(-k, LN2_LO, r);
    r = r + lo;
    const float z = r * r;
    float p = fma_(z, exp_poly(r), r);
    p = p + 1.0f;
    const int32_t ik = (int32_t)k;
    return u2f(f2u(p) + ((uint32_t)ik << 23));
}

DM float exp_(float x) { return exp_hl(x, 0.0f); }

DM float powr_(float x, float y)
{
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : u2f(0x7f800000u));
    int e;
    const float m = frexp_sqrt2(x, e);
    const float r = m - 1.0f;
    const float fe = (float)e;
    const float z = r * r;
    float c = (r * z) * log_poly(r);
    c = fma_(fe, LN2_LO, c);
    c = fma_(-0.5f, z, c);
    const float t = fe * LN2_HI;
    const float hi = t + r;
    const float bb = hi - t;
    const float err = (t - (hi - bb)) + (r - bb);
    const float lo = err + c;
    const float ph = y * hi;
    const float pe = fma_(y, hi, -ph);
    const float pl = fma_(y, lo, pe);
    return exp_hl(ph, pl);
}

// powr(x, y) for x in [0, 1], y > 0 with y |log x| <= 2 (oracle_math.h: om_powr_unit): single-word logarithm.  Without
// branches: the argument of exp lies in [-2, 0], where exp_hl's range tests never fire, and x = 0 (whose logarithm is
// finite garbage here) is put right by a select at the end.
DM float powr_unit_from_log_(float x, float y, float log_x)
{
    const float hi = y * log_x;
    const float k = rint_(hi * LOG2E);
    float r = fma_(-k, LN2_HI, hi);
    r = fma_(-k, LN2_LO, r);
    r = r + 0.0f;
    const float z = r * r;
    float p = fma_(z, exp_poly(r), r);
    p = p + 1.0f;
    const float v = u2f(f2u(p) + ((uint32_t)(int32_t)k << 23));
    return (x == 0.0f) ? 0.0f : v;
}
DM float powr_unit_(float x, float y) { return powr_unit_from_log_(x, y, log_(x)); }

// cbrt and pow with a fractional exponent for the table maker's power axes (oracle_math.h: om_cbrt, om_pow_frac)
DM float cbrt_(float x)
{
    const float ax = x < 0.0f ? -x : x;
    if (ax == 0.0f) return x;
    float r = powr_(ax, 0.333333343f);
    r = fma_(r, -9.934107e-09f * log_(ax), r);              // the exponent's second word: 1/3 - RN(1/3)
    return x < 0.0f ? -r : r;
}
DM float pow_frac_(float x, float y) { return (x < 0.0f) ? u2f(0x7fc00000u) : powr_(x, y); }

constexpr float PIO2_1 = 0x1.921fb6p+0f;
constexpr float PIO2_2 = -0x1.777a5cp-25f;
constexpr float PIO2_3 = -0x1.ee59dap-50f;
constexpr float TWO_O_PI = 0.636619772367581343f;

// oracle_math.h: om_sincos_2pi -- x in [0, RN(2 pi)].  `table`: the 33 rows {S, C}, in LDS or in global memory.
constexpr float SINCOS_2PI_MAX = 6.2831855f;
template <class Table>
DM void sincos_2pi_with_(float x, float &s, float &c, Table table)
{
    typedef float row_t __attribute__((ext_vector_type(2)));
    const float kf = fma_(x, MT_SC_16OPI, 12582912.0f);                 // 1.5 * 2^23 + rint(x * 16/pi): the integer is in the low bits
    const float k = kf - 12582912.0f;
    float r = fma_(-k, MT_SC_H1, x);
    r = fma_(-k, MT_SC_H2, r);
    const float z = r * r;
    const float sp = fma_(z, MT_SIN_S1, MT_SIN_S0) * z;
    const float sr = fma_(sp, r, r);
    const float cm = fma_(z, MT_COS_C1, -0.5f) * z;
    // the row's byte offset, 8 k: the pattern of kf is 0x4b400000 + k, and shifting it left by three drops the high bits mod 2^32
    const row_t row = table_row_<row_t>(table, (f2u(kf) << 3) - (0x4b400000u << 3));
    const float S = row.x, C = row.y;
    s = S + fma_(S, cm, C * sr);
    c = C + fma_(C, cm, -(S * sr));
}
DM void sincos_2pi_(float x, float &s, float &c) { sincos_2pi_with_<const float *>(x, s, c, kScTable); }

DM void sincos_cephes_(float x, float &s, float &c)
{
    const float k = rint_(x * TWO_O_PI);
    float r = fma_(-k, PIO2_1, x);
    r = fma_(-k, PIO2_2, r);
    r = fma_(-k, PIO2_3, r);
    const float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = fma_(ps, z, 8.3321608736e-3f);
    ps = fma_(ps, z, -1.6666654611e-1f);
    ps = fma_(ps * z, r, r);
    float pc = 2.443315711809948e-5f;
    pc = fma_(pc, z, -1.388731625493765e-3f);
    pc = fma_(pc, z, 4.166664568298827e-2f);
    pc = pc * (z * z);
    pc = fma_(-0.5f, z, pc);
    pc = pc + 1.0f;
    const int32_t q = (int32_t)k;
    const float a = (q & 1) ? pc : ps;
    const float b = (q & 1) ? ps : pc;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}
// oracle_math.h: om_sincos -- the table form on [0, RN(2 pi)], the Cephes form elsewhere
template <class Table>
DM void sincos_with_(float x, float &s, float &c, Table table)
{
    if (x >= 0.0f && x <= SINCOS_2PI_MAX) sincos_2pi_with_(x, s, c, table);
    else sincos_cephes_(x, s, c);
}
DM void sincos_(float x, float &s, float &c) { sincos_with_<const float *>(x, s, c, kScTable); }

// ---- once per recorded hit: binary64 (IEEE divide/sqrt/fma) ----
DM double atan_small_d(double t)
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
DM double atan2_d(double y, double x)
{
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const double mx = (ax > ay) ? ax : ay;
    const double mn = (ax > ay) ? ay : ax;
    double a;
    if (mx == 0.0) {
        a = 0.0;
    } else {
        double t = mn / mx;
        double off = 0.0;
        if (t > 0.41421356237309503) {
            t = (t - 1.0) / (t + 1.0);
            off = 0.78539816339744828;
        }
        a = off + atan_small_d(t);
    }
    if (ay > ax) a = 1.57079632679489656 - a;
    if (x < 0.0) a = 3.14159265358979323846 - a;
    if (y < 0.0) a = -a;
    return a;
}
DM float atan2_(float y, float x) { return (float)atan2_d((double)y, (double)x); }
DM float acos_(float v)
{
    const double d = (double)v;
    const double s = __builtin_sqrt((1.0 - d) * (1.0 + d));
    return (float)atan2_d(s, d);
}

// Single precision arccosine for the table maker, which takes one per path sample (the binary64 form above costs ~380
// instructions, this one ~45): |x| <= 0.5: pi/2 - asin(x); else 2 asin(sqrt((1 - |x|)/2)), reflected for x < 0, with the
// classic degree-4 minimax polynomial for asin on [0, 0.5] (Cephes asinf).  <= 2 ulp (tests/test_oracle.py); the
// OpenCL the reference runs on allows 4.  |x| > 1 or NaN gives NaN like acos().
DM float acos_f(float x)
{
    const float ax = __builtin_fabsf(x);
    if (!(ax <= 1.0f)) return u2f(0x7fc00000u);
    const bool big = ax > 0.5f;
    const float z = big ? 0.5f * (1.0f - ax) : ax * ax;
    const float s = big ? sqrt_near_(z) : ax;                 // (big: z is 0 or in [2^-26, 1/4], where sqrt_near_ is the IEEE root; else unused)
    const float p = (((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f;
    const float a = s + (s * z) * p;                         // asin(s)
    if (big) return (x < 0.0f) ? (3.14159265358979f - 2.0f * a) : (2.0f * a);
    return (x < 0.0f) ? (1.5707963267948966f + a) : (1.5707963267948966f - a);
}

} // namespace dm
