// Device functions of the photon propagator, shared by the kernels in prop_kernel.hip (classic register-resident
// scheduling, TABULATE variants) and prop_pool_kernel.hip (per-wave photon pools): the arithmetic of
//   resources/kernels/propagation_kernel.c.cl:73-404, 546-696, sparse_collision_kernel.c.cl:27-587,
//   mwcrng_kernel.cl:12-28 and the generated medium / spectrum / geometry functions,
// each function citing the lines it restates.  Scheduling lives in the kernels; nothing here depends on it.
#pragma once
#include <hip/hip_runtime.h>

#include "detmath.hip.h"
#include "../../include/clsimhip.h"
#include "kparams.h"
#include "instrument.hip.h"

namespace clsimhip {

#ifndef CLSIMHIP_BLOCK
#define CLSIMHIP_BLOCK 256                       // 4 waves per workgroup, up to 7 workgroups per CU (<= 72 VGPRs)
#define CLSIMHIP_MIN_WAVES 7
#endif
constexpr int kBlock = CLSIMHIP_BLOCK;
constexpr int kMinWavesPerSimd = CLSIMHIP_MIN_WAVES;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kStageRecords = 8;                 // hit stubs staged per wave and flush
constexpr int kStubWords = 16;
#ifndef CLSIMHIP_PRIO_SHIFT
#define CLSIMHIP_PRIO_SHIFT 1
#endif
constexpr int kPrioShift = CLSIMHIP_PRIO_SHIFT;  // a wave changes its issue priority every 2^kPrioShift loop trips
constexpr int kTabSlots = 512;                   // TABULATE with the impact-angle axis: path samples one wave pools per loop trip (half as many)
#ifndef CLSIMHIP_TAB_POOL
#define CLSIMHIP_TAB_POOL 448
#endif
constexpr int kTabPool = CLSIMHIP_TAB_POOL;                    // TABULATE, four axes: pooled path samples (d, tag) of a wave: two consecutive segments of every lane (save_path_wave_carry)
constexpr int kTabSegWords = 12;                 // ... and a segment's record: position + time, direction + 1/v, length, depth, depth step, weight
constexpr int kTabWaveWords = 2 * kTabPool + 2 * 64 * kTabSegWords;      // (two generations of segment records; 2432 words >= 2 * kTabSlots + 64)
static_assert(kTabWaveWords >= 2 * kTabSlots + 64 && kTabWaveWords % 4 == 0, "table maker: a wave's LDS region");
constexpr float kEpsilon = 0.00001f;             // propagation_kernel.c.cl:505
constexpr float kSpeedOfLight = 0.299792458f;    // propagation_kernel.h.cl:148
constexpr float kPi = 3.14159265359f;            // propagation_kernel.h.cl:150
constexpr uint32_t kNoStep = 0xffffffffu;
constexpr int kTiltScalarBins = 6;               // inner tilt bin edges kept as scalars (nd <= 8)

// kernel parameters, read with scalar loads from the constant address space
typedef const __attribute__((address_space(4))) KParams *KP;

// Makes the parameter pointer opaque to the optimiser at this point, so that the
// loads that follow stay here (phase-local SGPR live ranges) instead of being
// hoisted to the kernel entry and spilled.
DM KP fresh_params(KP p)
{
    asm volatile("" : "+s"(p));
    return p;
}

extern __shared__ __attribute__((aligned(16))) uint32_t lds_words[];

// Lane mask of a predicate.  (HIP's __ballot takes an int: the bool goes through v_cndmask 0/1 and a compare back into a
// mask, two vector instructions per vote that the mask the predicate already lives in does not need.)
DM uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

struct Rec4 { float a, b, c, d; };

// the divergent regions the census build counts visits and active lanes of (instrument.hip.h: CENSUS_REGION)
enum CensusRegion { kCensusCrossing = 0, kCensusFilter = 1, kCensusLiu = 2, kCensusHG = 3, kCensusSearchFull = 4, kCensusSearchNamed = 5,
                    kCensusCreation = 6, kCensusService = 7, kCensusScatter = 8, kCensusWalk = 9, kCensusAim = 10, kCensusRegions = 16 };

// The math tables of detmath.hip.h are the first dm::kMathTableWords words of every LDS image (tables.cpp puts them there): fixed
// addresses, so the row loads are ds_read_b128 / ds_read_b64 with the table's offset as an immediate.
// lds_words is the kernels' only LDS object and so sits at LDS address 0 (tests/test_codegen.py: no propagation kernel has a static
// group segment): the tables are named by their byte addresses.
DM float lds_log(float x) { return dm::log_with_(x, dm::LdsTable{0u}); }
DM void lds_sincos_2pi(float x, float &s, float &c) { dm::sincos_2pi_with_(x, s, c, dm::LdsTable{4u * dm::kMathLogWords}); }

DM float ldsf(uint32_t i) { return __builtin_bit_cast(float, lds_words[i]); }
DM uint32_t ldsu(uint32_t i) { return lds_words[i]; }
DM Rec4 lds_rec4(uint32_t i) { return *reinterpret_cast<const Rec4 *>(&lds_words[i]); }   // i % 4 == 0
DM uint32_t lds_u16(uint32_t off, uint32_t i)
{
    const uint32_t w = lds_words[off + (i >> 1)];
    return (i & 1) ? (w >> 16) : (w & 0xffffu);
}

// mwcrng_kernel.cl:12-20: x = lo32(x)*a + hi32(x); u = float_rtz(lo32(x)) / 2^32
// convert_float_rtz keeps the 24 leading bits of the word: v_cvt_f32_u32 does exactly that with the wave's single precision
// rounding mode set to "toward zero" (MODE.FP_ROUND bits 1:0 = 3) for that one instruction -- two scalar instructions in
// place of the seven vector ones that mask the low bits away by hand (count leading zeros, shift, compare, select, and).
DM float rng_co(uint64_t &x, uint32_t a)
{
    x = (x & 0xffffffffull) * (uint64_t)a + (x >> 32);
    const uint32_t lo = (uint32_t)x;
    float t;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_f32_u32_e32 %0, %1\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=v"(t) : "v"(lo));
    return t * 2.3283064365386963e-10f;              // exact: a power of two
}
DM float rng_oc(uint64_t &x, uint32_t a) { return 1.0f - rng_co(x, a); }

DM float sqr(float a) { return a * a; }
// a / b for an invariant divisor b with r = RN(1/b): exact when Compile() proved it (`ok`, wave-uniform)
DM float div_by(float a, float b, float r, bool ok)
{
    if (ok) {
        const float q = a * r;
        return dm::fma_(dm::fma_(-b, q, a), r, q);
    }
    return a / b;
}
// 1/x: dm::rcp_ (3 instructions, RN(1/x) for |x| in [2^-100, 2^100]) where Compile() bounded the argument (`fast`,
// wave-uniform), the IEEE divide otherwise
DM float rcp_sel(float x, bool fast) { return fast ? dm::rcp_(x) : 1.0f / x; }
constexpr uint32_t kFastLengths = 32u, kFastMatrices = 64u, kFastAniso = 128u;      // KParams::div_ok bits 5-7
// FAST (template parameter of the pooled kernel): Compile() found the standard configuration with every proof in hand
// (KVariant::fast) -- mixed Liu / Henyey-Greenstein scattering with beta <= 0.09, every invariant divisor proven, lengths
// and transforms bounded, tilt bins on scalars -- so the wave-uniform tests of those facts, two scalar branches each, are
// compiled out of the loop.  A scalar instruction costs the kernel as much as a vector one.
template <bool FAST>
DM float div_by_t(float a, float b, float r, bool ok)
{
    if (FAST) { const float q = a * r; return dm::fma_(dm::fma_(-b, q, a), r, q); }
    return div_by(a, b, r, ok);
}
template <bool FAST>
DM float rcp_t(float x, bool fast) { return FAST ? dm::rcp_(x) : rcp_sel(x, fast); }
DM float clampf(float v, float lo, float hi) { v = (v > lo) ? v : lo; return (v < hi) ? v : hi; }
DM int clampi(int v, int lo, int hi) { v = (v > lo) ? v : lo; return (v < hi) ? v : hi; }
// The same for bounds known to be ordered, as one median instruction (the compiler forms it only for constant bounds, and
// canonicalises a float before max/min).  A NaN comes out as the lower bound either way.
DM float clampf_ordered(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
// clampi(v, 0, hi) for a wave-uniform hi >= 0 (a table dimension minus one, in a scalar register)
DM int clamp_index(int v, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
    return r;
}

struct Vec3 { float x, y, z; };

// FunctionFromTable.cxx:213-232: interpolation bin and fraction of an equally spaced table
DM void table_bin_fraction(float start, float step, int n, float wlen, int &bin, float &fraction)
{
    const float q = (wlen - start) / step;
    const float fbin = __builtin_truncf(q);
    fraction = q - fbin;                                // modf
    bin = (int)fbin;
    if ((bin < 0) || ((bin == 0) && (fraction < 0.0f))) { bin = 0; fraction = 0.0f; }
    else if (bin >= n - 1) { bin = n - 2; fraction = 1.0f; }
}
// FunctionFromTable.cxx:279-291 (float data in the LDS image)
DM float table_value(uint32_t off, float start, float step, int n, float wlen)
{
    int bin; float fraction;
    table_bin_fraction(start, step, n, wlen, bin, fraction);
    const float a = ldsf(off + (uint32_t)bin), b = ldsf(off + (uint32_t)bin + 1u);
    return a + (b - a) * fraction;                      // mix
}
// RefIndexIceCube.cxx:128-180, or one FromTable function for all layers
DM float phase_ref_index(KP P, float wlen)
{
    if (P->phase_kind == CLSIMHIP_REFINDEX_TABLE) return table_value(P->off_phase, P->phase_start, P->phase_step, P->phase_n, wlen);
    const float x = wlen / P->micrometer;
    return P->n[0] + x * (P->n[1] + x * (P->n[2] + x * (P->n[3] + x * P->n[4])));
}
// MediumPropertiesSource.cxx:255-272 with the group index of RefIndexIceCube.cxx:158-163 or a FromTable override; without an override
// (CLSIMHIP_REFINDEX_DISPERSION) from the phase index and its derivative, MediumPropertiesSource.cxx:274-300 with getDispersion =
// RefIndexIceCube.cxx:205-215 (`x*4.f*n4` is (x*4.f)*n4).  FAST: Compile() has seen one of the first two (tables.cpp).
template <bool FAST = false>
DM float group_velocity(KP P, float wlen)
{
    if (P->group_kind == CLSIMHIP_REFINDEX_TABLE)
        return P->c_light / table_value(P->off_group, P->group_start, P->group_step, P->group_n, wlen);
    const float x = wlen / P->micrometer;
    const float np = P->n[0] + x * (P->n[1] + x * (P->n[2] + x * (P->n[3] + x * P->n[4])));
    if (!FAST && P->group_kind == CLSIMHIP_REFINDEX_DISPERSION) {
        const float n_inv = 1.0f / np;
        const float y = (P->n[1] + x * (2.0f * P->n[2] + x * (3.0f * P->n[3] + x * 4.0f * P->n[4]))) / P->micrometer;
        return P->c_light * (1.0f + y * wlen * n_inv) * n_inv;
    }
    const float np_corr = P->g[0] + x * (P->g[1] + x * (P->g[2] + x * (P->g[3] + x * P->g[4])));
    return P->c_light / (np * np_corr);
}

// Per-photon wavelength factors of the medium functions.  ICECUBE: the three transcendental terms;
// TABLE: interpolation fraction (sca_pow) and the record index of (bin, layer 0) (abs_pow, as bits).
struct IceFactors { float sca_pow, abs_pow, abs_exp; };

template <int MED>
DM IceFactors ice_factors(KP P, float wlen)
{
    IceFactors f = {0.0f, 0.0f, 0.0f};
    if (MED == CLSIMHIP_LENGTHS_ICECUBE) {
        // _Optimizers.cxx:237-240: powr(wlen*(1/400nm), -alpha)
        f.sca_pow = dm::powr_(wlen * P->ref_wlen_recip, P->neg_alpha);
        // _Optimizers.cxx:170-180: powr(x,-kappa), A*exp(-B/x), x = wlen/nm
        const float x = wlen / P->nanometer;
        f.abs_pow = dm::powr_(x, P->neg_kappa);
        f.abs_exp = P->abs_A * dm::exp_(P->neg_B / x);
    } else if (MED == CLSIMHIP_LENGTHS_TABLE) {
        int bin;
        table_bin_fraction(P->len_tab_start, P->len_tab_step, P->len_tab_n, wlen, bin, f.sca_pow);
        f.abs_pow = __builtin_bit_cast(float, (uint32_t)(bin * P->num_layers));
    }
    return f;
}
// scattering and absorption length of one layer (_Optimizers.cxx:123-250, FunctionConstant.cxx:81-100,
// FunctionFromTable.cxx:262-291 behind the switch(layer) of MediumPropertiesSource.cxx:89-123)
// rcp_sca / rcp_abs: RN(1 / length), which the layer walk needs next to a length (the divisions by it, the crossing updates).
// ICECUBE lengths ARE reciprocals -- 1 / (b400 x^-alpha) ... -- and the reciprocal of a reciprocal whose argument is at hand is two fma
// (dm::rcp_of_rcp_: exact for every argument, tested exhaustively) instead of v_rcp_f32 + two: formed here, together with the length,
// when Compile() has bounded the lengths (`fast`).  Every other case -- constant or tabulated lengths, unbounded ones -- forms a reciprocal
// where it needs one, as before (two more live registers through the walk cost the classic kernel's 72-register instantiations a spill).
template <int MED, bool FAST = false>
DM void layer_lengths(uint32_t off_layers, const float *len_table, const IceFactors &f, int layer, float &sca_len, float &abs_len,
                      float &rcp_sca, float &rcp_abs, bool fast)
{
    if (MED == CLSIMHIP_LENGTHS_TABLE) {
        const float4 r = *reinterpret_cast<const float4 *>(len_table + 4u * (__builtin_bit_cast(uint32_t, f.abs_pow) + (uint32_t)layer));
        abs_len = r.x + (r.y - r.x) * f.sca_pow;
        sca_len = r.z + (r.w - r.z) * f.sca_pow;
        return;
    }
    const Rec4 r = lds_rec4(off_layers + 4u * (uint32_t)layer);
    if (MED == CLSIMHIP_LENGTHS_ICECUBE) {
        const float x_sca = r.c * f.sca_pow, x_abs = r.a * f.abs_pow + f.abs_exp * r.b;
        sca_len = rcp_t<FAST>(x_sca, fast);
        abs_len = rcp_t<FAST>(x_abs, fast);
        if (FAST || fast) {                             // lengths within [1e-15, 1e15] (Compile()): inside rcp_of_rcp_'s range
            rcp_sca = dm::rcp_of_rcp_(sca_len, x_sca);
            rcp_abs = dm::rcp_of_rcp_(abs_len, x_abs);
        }
    } else {
        sca_len = r.c;
        abs_len = r.a;
    }
}
// The scattering angle's ten wave-uniform constants in ONE load group (round 6): read field by field the compiler sinks each scalar load
// to its use, and the Liu / Henyey-Greenstein choice became four groups each with its own wait in every loop trip.  The fields sit next to
// each other in KParams (static_asserts below), so two eight-dword loads cover them.
struct ScatterK {
    float mix_frac, mix_frac_rest, liu_beta, hg_g, hg_one_minus_g2, hg_one_plus_g2, hg_two_g;
    float rcp_mix_frac, rcp_mix_frac_rest, rcp_hg_two_g;
    uint32_t div_ok;
};
static_assert(offsetof(KParams, hg_two_g) - offsetof(KParams, mix_frac) == 24 && offsetof(KParams, liu_beta) - offsetof(KParams, mix_frac) == 8,
              "mix_frac, mix_frac_rest, liu_beta, hg_g, hg_one_minus_g2, hg_one_plus_g2, hg_two_g are seven consecutive floats");
static_assert(offsetof(KParams, div_ok) - offsetof(KParams, rcp_tilt_dz) == 20 && offsetof(KParams, rcp_mix_frac) - offsetof(KParams, rcp_tilt_dz) == 8,
              "rcp_tilt_dz, rcp_layer_thickness, rcp_mix_frac, rcp_mix_frac_rest, rcp_hg_two_g, div_ok are six consecutive words");
static_assert(sizeof(KParams) >= offsetof(KParams, rcp_tilt_dz) + 32 && sizeof(KParams) >= offsetof(KParams, mix_frac) + 32, "the two groups are read as eight words each");
DM ScatterK scatter_constants(KP P)
{
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    typedef const __attribute__((address_space(4))) u32x8 *group_ptr;
    const u32x8 a = *reinterpret_cast<group_ptr>(&P->mix_frac);
    const u32x8 b = *reinterpret_cast<group_ptr>(&P->rcp_tilt_dz);
    ScatterK K;
    K.mix_frac = dm::u2f(a[0]); K.mix_frac_rest = dm::u2f(a[1]); K.liu_beta = dm::u2f(a[2]); K.hg_g = dm::u2f(a[3]);
    K.hg_one_minus_g2 = dm::u2f(a[4]); K.hg_one_plus_g2 = dm::u2f(a[5]); K.hg_two_g = dm::u2f(a[6]);
    K.rcp_mix_frac = dm::u2f(b[2]); K.rcp_mix_frac_rest = dm::u2f(b[3]); K.rcp_hg_two_g = dm::u2f(b[4]);
    K.div_ok = b[5];
    return K;
}
// HenyeyGreenstein.cxx:69-92
template <bool FAST = false>
DM float hg_cos(const ScatterK &K, float u)
{
    const float s = 2.0f * u - 1.0f;
    // FAST: 1 - g^2 >= 2^-40 and 1 - g >= 2^-50 were checked at Compile() (tables.cpp), so the numerator and every divisor
    // 1 + g s, |s| <= 1, are inside div_near_'s range
    const float ii = FAST ? dm::div_near_(K.hg_one_minus_g2, 1.0f + K.hg_g * s) : K.hg_one_minus_g2 / (1.0f + K.hg_g * s);
    return clampf_ordered(div_by_t<FAST>(K.hg_one_plus_g2 - ii * ii, K.hg_two_g, K.rcp_hg_two_g, FAST || (K.div_ok & 16u) != 0), -1.0f, 1.0f);
}
// SimplifiedLiu.cxx:64-88
template <bool FAST = false>
DM float liu_cos(const ScatterK &K, float u)
{
    const float beta = K.liu_beta;
    // beta <= 0.09 (mean cosine >= 0.835, wave-uniform): beta |log u| <= 2 for u >= 2^-32, the single-word logarithm form
    const float p = (FAST || beta <= 0.09f) ? dm::powr_unit_from_log_(u, beta, lds_log(u)) : dm::powr_(u, beta);
    return clampf_ordered(2.0f * p - 1.0f, -1.0f, 1.0f);
}
// Mixed.cxx:115-157, single random number form
template <bool FAST = false>
DM float scattering_cos(KP P, uint64_t &x, uint32_t a)
{
    const ScatterK K = scatter_constants(P);
    const float rr = rng_co(x, a);
    if (!FAST) {
        const int kind = P->scatter_kind;
        if (kind == 0) return hg_cos(K, rr);
        if (kind == 1) return liu_cos(K, rr);
    }
    const uint32_t ok = FAST ? 0xffu : K.div_ok;
    CENSUS_REGION(P, kCensusScatter);
    if (rr < K.mix_frac) {
        CENSUS_REGION(P, kCensusLiu);
        return liu_cos<FAST>(K, div_by_t<FAST>(rr, K.mix_frac, K.rcp_mix_frac, (ok & 4u) != 0));
    }
    CENSUS_REGION(P, kCensusHG);
    return hg_cos<FAST>(K, div_by_t<FAST>(1.0f - rr, K.mix_frac_rest, K.rcp_mix_frac_rest, (ok & 8u) != 0));
}

// ScalarFieldAnisotropyAbsLenScaling.cxx:92-140
template <bool FAST = false>
DM float abs_len_corr(KP P, const Vec3 &d)
{
    const float n0 = (P->an_azx * d.x) + (P->an_azy * d.y);
    const float n1 = (P->an_mazy * d.x) + (P->an_azx * d.y);
    const float s0 = n0 * n0, s1 = n1 * n1, s2 = d.z * d.z;
    // dot(float4,float4) with a zero 4th component: the +0 term cannot change a sum of squares
    const float nB = (s0 * P->an_rl[0] + s1 * P->an_rl[1]) + s2 * P->an_rl[2];
    const float An = (s0 * P->an_l[0] + s1 * P->an_l[1]) + s2 * P->an_l[2];
    // RN(2/x) = 2 RN(1/x): a scaling by two is exact
    const float x = (P->an_B2 - nB) * An;
    return (FAST || (P->div_ok & kFastAniso) != 0u) ? 2.0f * dm::rcp_(x) : 2.0f / x;
}
// VectorTransformMatrix.cxx:101-135
DM void apply_matrix(const __attribute__((address_space(4))) float *m, int renorm, Vec3 &d, bool fast)
{
    const float x = (m[0] * d.x) + (m[1] * d.y) + (m[2] * d.z);
    const float y = (m[3] * d.x) + (m[4] * d.y) + (m[5] * d.z);
    const float z = (m[6] * d.x) + (m[7] * d.y) + (m[8] * d.z);
    d.x = x; d.y = y; d.z = z;
    if (renorm) {
        const float n2 = d.x * d.x + d.y * d.y + d.z * d.z;
        const float norm = fast ? dm::rsqrt_near_(n2) : dm::rsqrt_(n2);
        d.x = d.x * norm; d.y = d.y * norm; d.z = d.z * norm;
    }
}

// ScalarFieldIceTiltZShift.cxx:145-213.  The distance bin is the first j with
// nr < dist[j] (last bin otherwise); dist is ascending, so it is counted.
template <bool FAST = false>
DM float tilt_z_shift(KP P, float px, float py, float pz)
{
    const float z_rescaled = div_by_t<FAST>(pz - P->tilt_first_z, P->tilt_dz, P->rcp_tilt_dz, (P->div_ok & 1u) != 0);
    const int nz = P->tilt_nz, nd = P->tilt_nd;
    const uint32_t off_dist = P->off_tilt_dist;
    const int k = clamp_index((int)__builtin_floorf(z_rescaled), nz - 2);           // nz >= 2 (tables.cpp)
    const float fraction_z_above = z_rescaled - (float)k;
    const float fraction_z_below = 1.0f - fraction_z_above;
    const float nr = P->tilt_lnx * px + P->tilt_lny * py;
    int j = 1;
    if (FAST || nd <= kTiltScalarBins + 2) {
        // inner bin edges live in the parameter block (padded with +inf): compares against SGPRs, no LDS
#pragma unroll
        for (int t = 0; t < kTiltScalarBins; ++t) j += (nr >= P->tilt_inner_dist[t]) ? 1 : 0;
    } else {
        for (int t = 1; t < nd - 1; ++t) j += (nr >= ldsf(off_dist + t)) ? 1 : 0;
    }
    const Rec4 bin = lds_rec4(P->off_tilt_bins + 4u * (uint32_t)j);    // dist[j], dist[j]-dist[j-1], 1/width, proven
    const float thisDist = bin.a;
    // the proof bit differs per bin: select, the divide is only executed if some lane's bin lacks the proof
    const float q = (thisDist - nr) * bin.c;
    float frac_at_lower = dm::fma_(dm::fma_(-bin.b, q, thisDist - nr), bin.c, q);
    if (!FAST && __builtin_bit_cast(uint32_t, bin.d) == 0u) frac_at_lower = (thisDist - nr) / bin.b;
    const float frac_at_upper = 1.0f - frac_at_lower;
    const uint32_t lo = P->off_tilt_zcorr + (uint32_t)((j - 1) * nz + k);
    const uint32_t hi = lo + (uint32_t)nz;
    const float val_at_lower = (ldsf(lo + 1) * fraction_z_above + ldsf(lo) * fraction_z_below);
    const float val_at_upper = (ldsf(hi + 1) * fraction_z_above + ldsf(hi) * fraction_z_below);
    return (val_at_upper * frac_at_upper + val_at_lower * frac_at_lower);
}

// InterpolatedDistribution.cxx:236-336 (constant spacing, or kind 3: its own x values, :292-297).  The reference scans
// the cumulative table linearly for the first entry >= r; the table is
// non-decreasing, so a bisection lands on the same bin.
DM float generate_wavelength(KP P, int gen, uint64_t &x, uint32_t a)
{
    if (P->gen_kind[gen] == 1) return P->gen_value[gen];      // RandomValueConstant
    if (P->gen_kind[gen] == 2) {                               // WlenCherenkovNoDispersion.cxx:72-92
        const float u = rng_oc(x, a);
        return 1.0f / (P->gen_first[gen] + u * P->gen_spacing[gen]);
    }
    const float r = rng_oc(x, a);
    const uint32_t cum = P->off_gen_ycum[gen], yv = P->off_gen_yv[gen];
    int lo = 1, hi = P->gen_n[gen] - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ldsf(cum + mid) >= r) hi = mid; else lo = mid + 1;
    }
    const int k = lo - 1;
    const float this_acu = (k == 0) ? 0.0f : ldsf(cum + k);
    const float b = ldsf(yv + k);
    const bool own_x = (P->gen_kind[gen] == 3);                // wave-uniform where the generator is (Cherenkov steps: always)
    const float x0 = own_x ? ldsf(P->off_gen_xv[gen] + (uint32_t)k) : (float)k * P->gen_spacing[gen] + P->gen_first[gen];
    const float sp = own_x ? (ldsf(P->off_gen_xv[gen] + (uint32_t)k + 1u) - x0) : P->gen_spacing[gen];
    const float slope = (ldsf(yv + k + 1) - b) / sp;
    const float dy = r - this_acu;
    if ((b == 0.0f) && (slope == 0.0f)) return x0;
    else if (b == 0.0f) return x0 + dm::sqrt_(2.0f * dy / slope);
    else if (slope == 0.0f) return x0 + dy / b;
    else return x0 + (dm::sqrt_(dy * (2.0f * slope) / (b * b) + 1.0f) - 1.0f) * b / slope;
}

// FunctionFromTable.cxx:167-300
DM float wavelength_bias(KP P, float wavelength)
{
    if (P->bias_kind == 1) return P->bias_value;
    const float q = (wavelength - P->bias_start) / P->bias_step;
    const float fbin = __builtin_truncf(q);
    float fraction = q - fbin;
    int ibin = (int)fbin;
    const int n = P->bias_n;
    if ((ibin < 0) || ((ibin == 0) && (fraction < 0.0f))) { ibin = 0; fraction = 0.0f; }
    else if (ibin >= n - 1) { ibin = n - 2; fraction = 1.0f; }
    const uint32_t off = P->off_bias;
    const float v0 = ldsf(off + ibin), v1 = ldsf(off + ibin + 1);
    return v0 + (v1 - v0) * fraction;
}

// GeometrySource.cxx:685-700
DM void dom_position(KP P, uint32_t s, uint32_t d, float &x, float &y, float &z)
{
    const uint32_t rec = P->off_strings + 8u * s;
    const uint32_t index = (ldsu(rec + 4) >> 8) + d;
    int tx, ty;
    if (P->dom_in_lds) {                // wave-uniform: templates staged in LDS when they fit
        const uint32_t w = ldsu(P->off_dom_xy + index);
        tx = (int)(int16_t)(w & 0xffffu);
        ty = (int)(int16_t)(w >> 16);
        z = ldsf(P->off_dom_z + index);
    } else {
        tx = P->dom_tx[index];
        ty = P->dom_ty[index];
        z = P->dom_tz[index];
    }
    x = (float)tx * P->dom_mul_x + ldsf(rec + 5);
    y = (float)ty * P->dom_mul_y + ldsf(rec + 6);
}

// propagation_kernel.c.cl:83-129.  The reference branches on sinth > 0; a direction exactly along z is so rare that the
// wave takes the general formulas for all lanes (a division by zero there yields a value that is thrown away) and visits
// the special case only when a ballot says some lane needs it.
DM void scatter_direction(float cosa, float sina, Vec3 &d, float u)
{
    const float b = 2.0f * kPi * u;                                  // u in [0, 1): b in [0, RN(2 pi)]
    float sinb, cosb;
    lds_sincos_2pi(b, sinb, cosb);
    const float t = 1.0f - d.z * d.z;
    const float sinth = dm::sqrt_near_((t > 0.0f) ? t : 0.0f);     // 0 or >= 2^-24: |d.z| <= 1 is a float
    const float ox = d.x, oy = d.y, oz = d.z;
    // two numerators (at most 2 in magnitude) over one divisor in [2^-12, 1]: the exact divide of detmath.hip.h on a shared
    // reciprocal.  A numerator that is zero (sina = 0: a scattering cosine rounded to +-1), signed zero or below 2^-40 is
    // outside its range: such a lane -- one wave trip in 1e5 -- takes the IEEE divide in the block below.
    const float num_x = (oy * cosb + oz * ox * sinb) * sina, num_y = (ox * cosb - oz * oy * sinb) * sina;
    const float recip_sinth = dm::rcp_(sinth);             // (sinth = 0: infinite, and what follows from it is discarded below)
    d.x = ox * cosa - dm::div_near_with_(num_x, sinth, recip_sinth);
    d.y = oy * cosa + dm::div_near_with_(num_y, sinth, recip_sinth);
    d.z = oz * cosa + sina * sinb * sinth;
    const bool along_z = !(sinth > 0.0f);
    const bool ieee = !(dm::div_near_ok_(num_x) && dm::div_near_ok_(num_y));
    if (__builtin_expect(ballot(along_z || ieee) != 0ull, 0)) {
        if (ieee) {
            d.x = ox * cosa - num_x / sinth;
            d.y = oy * cosa + num_y / sinth;
        }
        const float sgn = (oz > 0.0f) ? 1.0f : ((oz < 0.0f) ? -1.0f : oz);
        d.x = along_z ? sina * cosb : d.x;
        d.y = along_z ? sina * sinb : d.y;
        d.z = along_z ? cosa * sgn : d.z;
    }
    // a rotated unit vector: its squared length is one to within a few ulps -- the integer form of the reciprocal root (detmath.hip.h:
    // rsqrt_unit_) when every lane's is within 1023, the general one for the whole wave otherwise
    const float len2 = sqr(d.x) + sqr(d.y) + sqr(d.z);
    const float recip_length = __builtin_expect(ballot(!dm::rsqrt_unit_ok_(len2)) == 0ull, 1) ? dm::rsqrt_unit_(len2) : dm::rsqrt_near_(len2);
    d.x *= recip_length; d.y *= recip_length; d.z *= recip_length;
}

// propagation_kernel.c.cl:206-223
DM void sph_dir_from_car(const Vec3 &d, float &theta, float &phi)
{
    const float r_inv = dm::rsqrt_(d.x * d.x + d.y * d.y + d.z * d.z);
    theta = 0.0f;
    if (__builtin_fabsf(d.z * r_inv) <= 1.0f) theta = dm::acos_(d.z * r_inv);
    else if (d.z < 0.0f) theta = kPi;
    if (theta < 0.0f) theta += 2.0f * kPi;
    phi = dm::atan2_(d.y, d.x);
    if (phi < 0.0f) phi += 2.0f * kPi;
}

// Per-lane photon state that the scatter loop touches every iteration.  What only a
// hit record needs (start position / direction, wavelength, initial absorption budget)
// is NOT kept: it is a pure function of the step and of the RNG state at the photon's
// birth, so the rare hit re-derives it from `rx_start` with the very code that created
// the photon (photon_birth).  7 VGPRs per lane buy an extra wave per SIMD.
struct Photon {
    float px, py, pz, pt;       // position, time
    Vec3 d;                     // direction
    float inv_groupvel, total_path;
    float abs_lens_left;
    uint32_t num_scatters;
    IceFactors ice;
    uint64_t rx_start;          // RNG state word when the photon was created
    int layer;                  // carried layer index (getTiltZShift_IS_CONSTANT, c.cl:521-523)
    float tab_remainder, tab_depth;     // TABULATE only: prevStepRemainder, depthPropagated (c.cl:530-534, 563)
    float tab_wlen;                     // TABULATE only: photonDirAndWlen.w (enters the impact-angle dot product)
};

struct Birth {                  // propagation_kernel.c.cl:132-184 + :587: what createPhotonFromTrack yields
    float x, y, z, t;
    Vec3 d;
    float wlen;
    float abs_lens_initial;
};

// c.cl:482-489: direction of a step from its (theta, phi); evaluated once when a lane takes the step
DM Vec3 step_direction(const DevStep *step_ptr)
{
    // (the global-memory copy of the sincos table: scan_steps_kernel, one of the two callers, stages no LDS image; once per step)
    float sin_t, cos_t, sin_p, cos_p;
    dm::sincos_(step_ptr->theta, sin_t, cos_t);
    dm::sincos_(step_ptr->phi, sin_p, cos_p);
    Vec3 d;
    d.x = sin_t * cos_p; d.y = sin_t * sin_p; d.z = cos_t;
    return d;
}

// the direction scan_steps_kernel left in a work record's step (in the places of theta, phi and weight)
DM Vec3 work_direction(const DevStep *work_step)
{
    Vec3 d;
    d.x = work_step->theta; d.y = work_step->phi; d.z = work_step->weight;
    return d;
}

// propagation_kernel.c.cl:132-184 + :587.  The step record is re-read from HBM/L2 here
// (48 B every ~30 loop iterations) instead of living in registers.  Consumes the RNG draws
// of a photon's birth in the reference's order: position, wavelength, azimuth, absorption budget.
template <bool FLASHER>
DM Birth photon_birth(KP P, const DevStep *step_ptr, const Vec3 &step_dir, uint64_t &rx, uint32_t ra)
{
    const DevStep st = *step_ptr;
    Birth b;
    const float shift = st.length * rng_co(rx, ra);
    const float inv_speed = 1.0f / (kSpeedOfLight * st.beta);
    b.x = st.x + step_dir.x * shift;
    b.y = st.y + step_dir.y * shift;
    b.z = st.z + step_dir.z * shift;
    b.t = st.t + inv_speed * shift;
    const uint32_t source_type = st.source_type_and_pad & 0xffu;
    b.d = step_dir;
    if (!FLASHER || source_type == 0) {
        const float wavelength = generate_wavelength(P, 0, rx, ra);
        const float rcp = 1.0f / (st.beta * phase_ref_index(P, wavelength));
        const float cos_c = (rcp < 1.0f) ? rcp : 1.0f;
        const float sin_c = dm::sqrt_near_(1.0f - cos_c * cos_c);
        b.wlen = wavelength;
        scatter_direction(cos_c, sin_c, b.d, rng_co(rx, ra));
    } else {
        // generateWavelength(number): 0 for an out-of-range generator (MediumPropertiesSource.cxx:392-432)
        b.wlen = (source_type < (uint32_t)P->num_gen) ? generate_wavelength(P, (int)source_type, rx, ra) : 0.0f;
    }
    // c.cl:582-588: a fixed budget draws no random number
    b.abs_lens_initial = P->has_fixed_abs ? P->fixed_abs : -lds_log(rng_oc(rx, ra));
    return b;
}

// c.cl:546-596
template <int MED, bool TILT, bool FLASHER, bool TAB, bool FAST = false>
DM void create_photon(KP P, const DevStep *step_ptr, const Vec3 &step_dir, uint64_t &rx, uint32_t ra, Photon &ph)
{
    CENSUS_REGION(P, kCensusCreation);
    ph.rx_start = rx;
    // TABULATE: the first sub-step is drawn between the photon's creation and its (fixed) absorption budget,
    // which draws nothing (c.cl:559-563, 582-588)
    const Birth b = photon_birth<FLASHER>(P, step_ptr, step_dir, rx, ra);
    if (TAB) { ph.tab_remainder = P->tab_volume_step * rng_oc(rx, ra); ph.tab_depth = 0.0f; ph.tab_wlen = b.wlen; }
    ph.px = b.x; ph.py = b.y; ph.pz = b.z; ph.pt = b.t;
    ph.d = b.d;
    ph.num_scatters = 0;
    ph.total_path = 0.0f;
    if (!TILT) ph.layer = clamp_index((int)div_by_t<FAST>(ph.pz - P->layer_bottom, P->layer_thickness, P->rcp_layer_thickness, (P->div_ok & 2u) != 0), P->num_layers - 1);
    ph.inv_groupvel = 1.0f / group_velocity<FAST>(P, b.wlen);
    ph.abs_lens_left = b.abs_lens_initial;
    ph.ice = ice_factors<MED>(P, b.wlen);
}

// propagation_kernel.c.cl:598-696: distance to the next scatter / absorption through the layers
template <int MED, bool TILT, bool ANISO, bool FAST = false>
DM float propagate_through_layers(KP P, Photon &ph, uint64_t &rx, uint32_t ra)
{
    const float thickness = P->layer_thickness, bottom = P->layer_bottom;
    const int num_layers = P->num_layers;
    const uint32_t off_layers = P->off_layers;
    const float *len_table = (MED == CLSIMHIP_LENGTHS_TABLE) ? P->len_table : nullptr;
    const bool fast = FAST || (P->div_ok & kFastLengths) != 0u;
    CENSUS_REGION(P, kCensusWalk);
    float effective_z;
    int current_layer;
    if (TILT) {
        effective_z = ph.pz - tilt_z_shift<FAST>(P, ph.px, ph.py, ph.pz);
        current_layer = clamp_index((int)div_by_t<FAST>(effective_z - bottom, thickness, P->rcp_layer_thickness, (P->div_ok & 2u) != 0), num_layers - 1);
    } else {
        effective_z = ph.pz - P->tilt_const;
        current_layer = ph.layer;
    }
    const float dz = ph.d.z;
    // without anisotropy the factor is the literal 1.f: x*1 and x/1 are exact, so both are skipped
    const float corr = (ANISO && P->has_abs_corr) ? abs_len_corr<FAST>(P, ph.d) : 1.0f;
    if (ANISO) ph.abs_lens_left *= corr;
    const float lower = ((float)current_layer * thickness) + bottom;
    float boundary = (dz < 0.0f) ? lower : (lower + thickness);
    const float sca_step_left = -lds_log(rng_oc(rx, ra));
    float sca_len, abs_len, rcp_sca = 0.0f, rcp_abs = 0.0f;     // the current layer's lengths and, when `seeded`, RN(1 / length)
    const bool seeded = (MED == CLSIMHIP_LENGTHS_ICECUBE) && fast;
    layer_lengths<MED, FAST>(off_layers, len_table, ph.ice, current_layer, sca_len, abs_len, rcp_sca, rcp_abs, fast);
    const float recip_thickness = P->recip_thickness;
    // Two divides of one numerator by lengths that Compile() has bounded to (2^-50, 2^50) (`fast`): the 8-instruction exact
    // divide when every lane's height above the boundary is inside its range (a photon ON a boundary, or within 1e-12 m of
    // one at z = 0, sends its wave through the IEEE sequence instead; positions stay below 2^55 m with those lengths)
    const float to_boundary = boundary - effective_z;
    float over_sca, over_abs;
    if (__builtin_expect(fast && (ballot(!dm::div_near_ok_(to_boundary)) == 0ull), 1)) {
        over_sca = seeded ? dm::div_near_with_(to_boundary, sca_len, rcp_sca) : dm::div_near_(to_boundary, sca_len);
        over_abs = seeded ? dm::div_near_with_(to_boundary, abs_len, rcp_abs) : dm::div_near_(to_boundary, abs_len);
    } else {
        over_sca = to_boundary / sca_len;
        over_abs = to_boundary / abs_len;
    }
    float ais = (dz * sca_step_left - over_sca) * recip_thickness;
    float aia = (dz * ph.abs_lens_left - over_abs) * recip_thickness;
    int j = current_layer;
    {
        // c.cl:643-668 has one loop for photons going down and one for photons going up; a wave holds both kinds, so the
        // two loops cost it the sum of their longest walks.  One loop with a sign does the same arithmetic (x - y is
        // x + (-y), a product with +-1 is exact, and (-ais < 0) is (ais > 0) also for signed zeros) in the longer walk only.
        const bool down = (dz < 0.0f);
        const float sgn = down ? -1.0f : 1.0f;
        const int step = down ? -1 : 1;
        const int last = down ? 0 : (num_layers - 1);
        const float signed_thickness = sgn * thickness;
        while ((j != last) && (sgn * ais > 0.0f) && (sgn * aia > 0.0f)) {
            CENSUS_REGION(P, kCensusCrossing);
            j += step;
            boundary += signed_thickness;
            layer_lengths<MED, FAST>(off_layers, len_table, ph.ice, j, sca_len, abs_len, rcp_sca, rcp_abs, fast);
            ais -= sgn * (seeded ? rcp_sca : rcp_t<FAST>(sca_len, fast));
            aia -= sgn * (seeded ? rcp_abs : rcp_t<FAST>(abs_len, fast));
        }
    }
    float distance, to_absorption;
    if ((current_layer == j) || (__builtin_fabsf(dz) < kEpsilon)) {
        distance = sca_step_left * sca_len;
        to_absorption = ph.abs_lens_left * abs_len;
    } else {
        const float recip_dz = dm::rcp_(dz);            // 1e-5 <= |dz| <= 1 on this branch
        distance = (ais * thickness * sca_len + boundary - effective_z) * recip_dz;
        to_absorption = (aia * thickness * abs_len + boundary - effective_z) * recip_dz;
    }
    if (!TILT) ph.layer = j;
    if (to_absorption < distance) {
        distance = to_absorption;
        ph.abs_lens_left = 0.0f;
    } else {
        const float left = to_absorption - distance;       // >= +0, below 2^55
        ph.abs_lens_left = __builtin_expect(fast && (ballot(!dm::div_near_ok_(left)) == 0ull), 1) ? (seeded ? dm::div_near_with_(left, abs_len, rcp_abs) : dm::div_near_(left, abs_len)) : left / abs_len;
    }
    if (ANISO) ph.abs_lens_left = ph.abs_lens_left / corr;       // (the exact divide with its range test gains nothing here: measured)
    return distance;
}

// ---- DOM search: sparse_collision_kernel.c.cl:27-303, 462-547 (STOP_PHOTONS_ON_DETECTION) ----
struct Detector {               // wave-uniform values of the search, fetched once per call
    uint32_t off_strings, off_sets, off_layer_to_om;
    int max_layers;
    float string_max_radius_sq, string_max_radius, om_radius_sq, pancake;
    int has_pancake;
};

// collision c.cl:27-192
DM void collide_with_string(KP P, const Detector &D, uint32_t s, float dir_len_xy_sqr, const Photon &ph, float &step_len,
                            bool &hit, uint32_t &hit_string, uint32_t &hit_dom)
{
    const Rec4 str = lds_rec4(D.off_strings + 8u * s);      // x, y, maxZ+R, minZ-R
    const float wx = str.a - ph.px, wy = str.b - ph.py;
    {
        const float smin = sqr((ph.px - str.a) * ph.d.y - (ph.py - str.b) * ph.d.x) / dir_len_xy_sqr;
        if (smin > D.string_max_radius_sq) return;
    }
    {
        // Not in the reference: a conservative early-out.  The test above is about the INFINITE line.
        // A DOM of this string can only be hit at a point of the segment [0, step_len] that lies within
        // GEO_STRING_MAX_RADIUS of the string axis in xy (the hit point is inside the oversized sphere,
        // whose centre is within maxR - OM_RADIUS of the axis).  If the point of the segment closest to
        // the axis is an END point and that end point is farther away -- with a margin that exceeds the
        // float error of these few operations by more than an order of magnitude -- every sphere test
        // of this string fails, so skipping them cannot change the result.
        const float along = wx * ph.d.x + wy * ph.d.y;          // (axis - start) . dir_xy
        const float ex = wx - step_len * ph.d.x, ey = wy - step_len * ph.d.y;
        const float reach = D.string_max_radius + (1e-3f + 9.5367431640625e-7f * (__builtin_fabsf(ph.px) + __builtin_fabsf(ph.py) + __builtin_fabsf(str.a) + __builtin_fabsf(str.b)));
        const float reach_sq = reach * reach;
        if ((along <= 0.0f) && (wx * wx + wy * wy > reach_sq)) return;
        if ((along >= step_len * dir_len_xy_sqr) && (ex * ex + ey * ey > reach_sq)) return;
    }
    if ((ph.d.z > 0.0f) && (ph.pz > str.c)) return;
    if ((ph.d.z < 0.0f) && (ph.pz < str.d)) return;
    const uint32_t set = ldsu(D.off_strings + 8u * s + 4) & 0xffu;
    const Rec4 lay = lds_rec4(D.off_sets + 4u * set);        // nlayers (bits), start z, height
    const float start_z = lay.b, height = lay.c;
    const int nl = (int)__builtin_bit_cast(uint32_t, lay.a);
    int low = (int)((ph.pz - start_z) / height);
    int high = (int)((ph.pz + ph.d.z * step_len - start_z) / height);
    if (high < low) { const int tmp = low; low = high; high = tmp; }
    low = clampi(low, 0, nl - 1);
    high = clampi(high, 0, nl - 1);
    const uint32_t base = set * (uint32_t)D.max_layers;
    for (int layer = low; layer <= high; ++layer) {
        const uint32_t dom = lds_u16(D.off_layer_to_om, base + (uint32_t)layer);
        if (dom == 0xFFFFu) continue;
        float dom_x, dom_y, dom_z;
        dom_position(P, s, dom, dom_x, dom_y, dom_z);
        const float dx = dom_x - ph.px, dy = dom_y - ph.py, dzz = dom_z - ph.pz;
        // dot() of float4s whose 4th component is 0: ((x+y)+z); the trailing +0 only matters for -0
        const float dr2 = (dx * dx + dy * dy) + dzz * dzz;
        const float urdot = (dx * ph.d.x + dy * ph.d.y) + dzz * ph.d.z;
        float discr = sqr(urdot) - dr2 + D.om_radius_sq;
        if (discr < 0.0f) continue;
        discr = D.has_pancake ? (dm::sqrt_(discr) / D.pancake) : dm::sqrt_(discr);
        if (urdot + discr < 0.0f) continue;
        const float smin1 = urdot - discr;
        if (smin1 < 0.0f) continue;
        if (smin1 < step_len) {
            step_len = smin1;
            hit_string = s;
            hit_dom = dom;
            hit = true;
        }
    }
}

// String proximity map (kparams.h): xy distance [m] that a photon at (x, y) can travel before it could touch a DOM
DM uint32_t free_flight_bound(KP P, float x, float y)
{
    const int n = P->prox_n;
    const int ix = clamp_index((int)((x - P->prox_x0) * P->prox_inv_cell), n - 1);
    const int iy = clamp_index((int)((y - P->prox_y0) * P->prox_inv_cell), n - 1);
    // (24-bit multiplies -- full rate where the 32-bit one is a quarter -- in this and the two other index computations of the loop: measured
    // 0.4 % SLOWER, profiles/r04/ab_quarter_rate.txt: v_mad_u64_u32 forms the product and the sum in one instruction)
    return P->prox_map[(uint32_t)iy * (uint32_t)n + (uint32_t)ix];
}
DM float free_flight_of(uint32_t word) { return (float)(word & 0xffu) * 0.25f; }

// Between the first level (the step reaches the nearest string's cylinder) and the second (the DOM proximity map): when the step
// is too short to reach any OTHER string (bits 8-15 of the map word), a DOM can only be hit on the string the word names, and
// only at a point of the segment whose xy distance from that string's axis is at most prox_reach (the hit point lies on a DOM
// sphere, whose centre is within the largest DOM offset of the axis; a pancaked DOM keeps its lateral extent).  The xy
// projection of the segment comes that close unless
//   * the string lies behind: (axis - start) . d_xy <= 0 and the start is farther away than prox_reach, or
//   * the infinite line misses the cylinder: cross(axis - start, d_xy)^2 > |d_xy|^2 reach^2
// (the case of a segment that ends before it gets there is what the first level tested).  Both with the margin of
// collide_with_string's early-out (1 mm + 4 ulp of the coordinates).  A photon that passes a string at 10 m has a chance of
// reach / (pi * 10 m) = 6 % to be aimed at it: the rest needs neither the DOM proximity map nor a search.
DM bool segment_misses_string(KP P, const Photon &ph, float len, uint32_t word)
{
    CENSUS_REGION(P, kCensusAim);
    const uint32_t s = word >> 16;
    if (s == 0xffffu) return false;
    if (!(len < (float)((word >> 8) & 0xffu) * 0.25f)) return false;         // another string is within reach
    const Rec4 str = lds_rec4(P->off_strings + 8u * s);        // x, y, ...
    const float wx = str.a - ph.px, wy = str.b - ph.py;
    const float reach = P->prox_reach + (1e-3f + 9.5367431640625e-7f * (__builtin_fabsf(ph.px) + __builtin_fabsf(ph.py) + __builtin_fabsf(str.a) + __builtin_fabsf(str.b)));
    const float reach_sq = reach * reach;
    const float along = wx * ph.d.x + wy * ph.d.y;
    const float cross = wx * ph.d.y - wy * ph.d.x;
    const float m = ph.d.x * ph.d.x + ph.d.y * ph.d.y;
    const bool behind = (along <= 0.0f) && (wx * wx + wy * wy > reach_sq);
    const bool beside = cross * cross > (m * reach_sq) * 1.0001f;
    return behind || beside;
}

// Second and third level of the search filter (kparams.h: DOM proximity map), for a lane whose step of length `len` reaches a
// string cylinder.  The cell names the nearest DOM and bounds the distance to every other one: a step at least that long
// goes to the full search.  Otherwise only the named DOM is within reach, and the full search -- whatever its own pruning
// does -- can report nothing but a hit on that DOM, which requires the segment to touch its sphere (a pancaked DOM lies
// inside it).  So the segment's closest approach to the DOM centre is taken here, against the radius plus 6 cm (5 cm of the
// map's safety, 1 cm for this arithmetic in single precision at coordinates of a few hundred metres): farther away, the
// search would find nothing and is skipped; closer, the lane goes to the full search, which decides.  Conservative in one
// direction only, so no bit of the result depends on it.  Photons born at a DOM (flashers) spend their lives within metres
// of it: nearly all of their steps pass this sphere by.
// Returns 0: no search (it would find nothing); 1: the full search; 2 + id: only DOM `id` is in reach and the segment comes
// close to it -- the search may then be confined to what the full search would do for that one DOM (find_collision_named).
constexpr uint32_t kSearchNone = 0u, kSearchFull = 1u, kSearchNamed = 2u;
// INSIDE (the flasher instantiations): also the test for a photon that starts inside the named DOM, below.
template <bool INSIDE = false>
DM uint32_t dom_search_needed(KP P, const Photon &ph, float len)
{
    CENSUS_REGION(P, kCensusFilter);
    const float inv = P->dprox_inv_cell;
    const int ny = P->dprox_ny, nz = P->dprox_nz;
    const int ix = clamp_index((int)((ph.px - P->dprox_x0) * inv), P->dprox_nx - 1);
    const int iy = clamp_index((int)((ph.py - P->dprox_y0) * inv), ny - 1);
    const int iz = clamp_index((int)((ph.pz - P->dprox_z0) * inv), nz - 1);
    // the cell: {word, centre of the DOM it names} in one 16-byte load (z runs fastest; at most 2^24 cells)
    const uint4 cell = P->dom_cells[((uint32_t)ix * (uint32_t)ny + (uint32_t)iy) * (uint32_t)nz + (uint32_t)iz];
    const uint32_t w = cell.x;
    const float others = (float)((w >> 16) & 0xffu) * 0.25f;
    if (!(len < others)) return kSearchFull;
    const uint32_t id = w & 0xffffu;
    if (id == 0xffffu) return kSearchNone;
    const float wx = dm::u2f(cell.y) - ph.px, wy = dm::u2f(cell.z) - ph.py, wz = dm::u2f(cell.w) - ph.pz;        // (== dom_centres[id])
    // The reference's own sphere test for this DOM (c.cl:133-163) begins with urdot = (centre - photon) . direction -- these
    // operands, this order -- and discards the DOM when smin1 = urdot - discr < 0 with some discr >= 0: "starting inside the
    // DOM", which lets a flasher's photons leave the sphere they are born in (:157-159).
    //  * urdot < 0, the centre lies behind the photon: smin1 <= urdot < 0 whatever discr is.  No hit, no search.
    //  * INSIDE: urdot >= 0 and smin1 < 0, i.e. urdot < sqrt(urdot^2 - dr^2 + R^2) / pancake, i.e. (pancake^2 - 1) urdot^2 < R^2 - dr^2.
    //    Decided here only when it holds by a margin (0.1 % and 1e-5 m^2 against rounding errors of 3e-7 m^2 in the reference's
    //    discr, R^2 = 0.68 m^2): no hit, no search; anything closer to the boundary goes to the search, which decides.
    // A photon born at a DOM spends its first trips inside that sphere, one search each (profiles/r03/census_regions.txt: 0.27
    // searches per trip in C5, the parked lanes 5.8 % of all): with this they never park.
    const float urdot = (wx * ph.d.x + wy * ph.d.y) + wz * ph.d.z;
    if (urdot < 0.0f) return kSearchNone;
    if (INSIDE) {
        const float room = P->om_radius_sq - ((wx * wx + wy * wy) + wz * wz);
        const float p = P->has_pancake ? P->pancake : 1.0f;
        if ((p * p - 1.0f) * (urdot * urdot) < room * 0.999f - 1e-5f) return kSearchNone;
    }
    const float along = clampf_ordered(urdot, 0.0f, len);      // len > 0
    const float qx = wx - along * ph.d.x, qy = wy - along * ph.d.y, qz = wz - along * ph.d.z;
    const float reach = P->dprox_radius + 0.01f;
    return ((qx * qx + qy * qy) + qz * qz > reach * reach) ? kSearchNone : (kSearchNamed + id);
}

// The search for a lane that dom_search_needed() sent here with a name: every DOM but `id` is farther from the photon than
// the step is long, so the reference's search (find_collision below: sparse_collision_kernel.c.cl:462-547 -> :305-460 ->
// :194-303 -> :27-192) can report nothing but a hit on that DOM -- and reports it only if its own pruning leads it there:
//   (1) one of the string's cells lies in the range of cells the segment's end points span in the string's subdetector
//       (c.cl:478-540),
//   (2) the string passes the tests of checkForCollision_OnString (:45-75): infinite line within GEO_STRING_MAX_RADIUS of the
//       axis, photon not already above / below the string and moving away,
//   (3) one of the z layers between the end points' layers names the DOM (:96-118); a DOM may lie in several layers and is
//       then tested several times, which changes nothing after the first (smin1 < step_len is strict),
//   (4) the sphere test itself (:133-190).
// These are evaluated here for the one string and the one DOM, with the arithmetic of the full search, and nothing else is:
// the result is the full search's for every input (tests/test_named_search_gpu.py runs both on whole production bunches).
// A hit on no other DOM can precede it, so the step length the ranges are computed from is the one passed in.
template <bool FAST = false>
DM bool find_collision_named(KP P, const Photon &ph, float &step_len, uint32_t id, const uint4 named, uint32_t &hit_string, uint32_t &hit_dom)
{
    CENSUS_REGION(P, kCensusSearchNamed);
    const float dir_len_xy_sqr = sqr(ph.d.x) + sqr(ph.d.y);
    if (dir_len_xy_sqr <= 0.0f) return false;
    const uint32_t s = named.x & 0xffffu, dom = named.x >> 16;
    const int cell_x0 = (int)(named.y & 0xfffu), cell_y0 = (int)((named.y >> 12) & 0xfffu);
    const int cell_x1 = (int)(named.w & 0xfffu), cell_y1 = (int)((named.w >> 12) & 0xfffu);
    const uint32_t sd = named.y >> 24;
    {   // (1)
        const uint32_t off_subdet = P->off_subdet;
        const Rec4 g0 = lds_rec4(off_subdet + 12u * sd);
        const Rec4 g1 = lds_rec4(off_subdet + 12u * sd + 4u);
        const Rec4 g2 = lds_rec4(off_subdet + 12u * sd + 8u);
        const int nx = (int)__builtin_bit_cast(uint32_t, g0.a), ny = (int)__builtin_bit_cast(uint32_t, g0.b);
        const float wx = g0.c, wy = g0.d, sx = g1.a, sy = g1.b;
        const uint32_t proven = FAST ? 3u : __builtin_bit_cast(uint32_t, g1.d);       // (lanes may be in different subdetectors here)
        const bool okx = (proven & 1u) != 0, oky = (proven & 2u) != 0;
        int low_x, low_y, high_x, high_y;
        if (FAST) {
            low_x = (int)div_by_t<true>(ph.px - sx, wx, g2.a, true);
            low_y = (int)div_by_t<true>(ph.py - sy, wy, g2.b, true);
            high_x = (int)div_by_t<true>(ph.px + ph.d.x * step_len - sx, wx, g2.a, true);
            high_y = (int)div_by_t<true>(ph.py + ph.d.y * step_len - sy, wy, g2.b, true);
        } else {
            // the proof bits are per subdetector and the lanes here may be in different ones: select per lane
            const float ax = ph.px - sx, ay = ph.py - sy, bx = ph.px + ph.d.x * step_len - sx, by = ph.py + ph.d.y * step_len - sy;
            low_x = (int)(okx ? div_by_t<true>(ax, wx, g2.a, true) : ax / wx);
            low_y = (int)(oky ? div_by_t<true>(ay, wy, g2.b, true) : ay / wy);
            high_x = (int)(okx ? div_by_t<true>(bx, wx, g2.a, true) : bx / wx);
            high_y = (int)(oky ? div_by_t<true>(by, wy, g2.b, true) : by / wy);
        }
        if (high_x < low_x) { const int tmp = low_x; low_x = high_x; high_x = tmp; }
        if (high_y < low_y) { const int tmp = low_y; low_y = high_y; high_y = tmp; }
        low_x = clampi(low_x, 0, nx - 1); low_y = clampi(low_y, 0, ny - 1);
        high_x = clampi(high_x, 0, nx - 1); high_y = clampi(high_y, 0, ny - 1);
        // the string is met in any of its cells (and then possibly several times: the repeats find the same DOM at the same
        // distance, which is not closer than itself)
        if ((cell_x1 < low_x) || (cell_x0 > high_x) || (cell_y1 < low_y) || (cell_y0 > high_y)) return false;
    }
    // (2)
    const uint32_t off_strings = P->off_strings;
    const Rec4 str = lds_rec4(off_strings + 8u * s);          // x, y, maxZ+R, minZ-R
    {
        const float smin = sqr((ph.px - str.a) * ph.d.y - (ph.py - str.b) * ph.d.x) / dir_len_xy_sqr;
        if (smin > P->string_max_radius_sq) return false;
    }
    if ((ph.d.z > 0.0f) && (ph.pz > str.c)) return false;
    if ((ph.d.z < 0.0f) && (ph.pz < str.d)) return false;
    {   // (3)
        const uint32_t set = ldsu(off_strings + 8u * s + 4) & 0xffu;
        const Rec4 lay = lds_rec4(P->off_sets + 4u * set);        // nlayers (bits), start z, height
        const float start_z = lay.b, height = lay.c;
        const int nl = (int)__builtin_bit_cast(uint32_t, lay.a);
        int low = (int)((ph.pz - start_z) / height);
        int high = (int)((ph.pz + ph.d.z * step_len - start_z) / height);
        if (high < low) { const int tmp = low; low = high; high = tmp; }
        low = clampi(low, 0, nl - 1);
        high = clampi(high, 0, nl - 1);
        const int first = (int)(named.z & 0xffffu), last = (int)(named.z >> 16);
        if ((high < first) || (low > last)) return false;
    }
    // (4) the DOM's position: dom_centres holds what dom_position() reconstructs, made with the same operations
    const float4 c = P->dom_centres[id];
    const float dx = c.x - ph.px, dy = c.y - ph.py, dzz = c.z - ph.pz;
    const float dr2 = (dx * dx + dy * dy) + dzz * dzz;
    const float urdot = (dx * ph.d.x + dy * ph.d.y) + dzz * ph.d.z;
    float discr = sqr(urdot) - dr2 + P->om_radius_sq;
    if (discr < 0.0f) return false;
    discr = P->has_pancake ? (dm::sqrt_(discr) / P->pancake) : dm::sqrt_(discr);
    if (urdot + discr < 0.0f) return false;
    const float smin1 = urdot - discr;
    if (smin1 < 0.0f) return false;
    if (!(smin1 < step_len)) return false;
    step_len = smin1;
    hit_string = s;
    hit_dom = dom;
    return true;
}

// collision c.cl:194-303 + :462-547
template <bool FAST = false>
DM bool find_collision(KP P, const Photon &ph, float &step_len, uint32_t &hit_string, uint32_t &hit_dom)
{
    CENSUS_REGION(P, kCensusSearchFull);
    const float dir_len_xy_sqr = sqr(ph.d.x) + sqr(ph.d.y);
    if (dir_len_xy_sqr <= 0.0f) return false;
    Detector D;
    D.off_strings = P->off_strings; D.off_sets = P->off_sets; D.off_layer_to_om = P->off_layer_to_om;
    D.max_layers = P->max_layers;
    D.string_max_radius_sq = P->string_max_radius_sq; D.string_max_radius = P->string_max_radius; D.om_radius_sq = P->om_radius_sq;
    D.pancake = P->pancake; D.has_pancake = P->has_pancake;
    const int num_subdet = P->num_subdet;
    const uint32_t off_subdet = P->off_subdet;
    bool hit = false;
    for (int sd = 0; sd < num_subdet; ++sd) {
        const Rec4 g0 = lds_rec4(off_subdet + 12u * (uint32_t)sd);        // nx, ny (bits), width x, width y
        const Rec4 g1 = lds_rec4(off_subdet + 12u * (uint32_t)sd + 4u);   // start x, start y, cell offset, proof bits
        const Rec4 g2 = lds_rec4(off_subdet + 12u * (uint32_t)sd + 8u);   // 1/width x, 1/width y
        const int nx = (int)__builtin_bit_cast(uint32_t, g0.a), ny = (int)__builtin_bit_cast(uint32_t, g0.b);
        const float wx = g0.c, wy = g0.d, sx = g1.a, sy = g1.b;
        const uint32_t cells = __builtin_bit_cast(uint32_t, g1.c);
        // all lanes are in the same subdetector here, so the proof bits are wave-uniform
        const uint32_t proven = FAST ? 3u : (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint32_t, g1.d));
        const bool okx = (proven & 1u) != 0, oky = (proven & 2u) != 0;
        int low_x = (int)div_by_t<FAST>(ph.px - sx, wx, g2.a, okx);
        int low_y = (int)div_by_t<FAST>(ph.py - sy, wy, g2.b, oky);
        int high_x = (int)div_by_t<FAST>(ph.px + ph.d.x * step_len - sx, wx, g2.a, okx);
        int high_y = (int)div_by_t<FAST>(ph.py + ph.d.y * step_len - sy, wy, g2.b, oky);
        if (high_x < low_x) { const int tmp = low_x; low_x = high_x; high_x = tmp; }
        if (high_y < low_y) { const int tmp = low_y; low_y = high_y; high_y = tmp; }
        low_x = clampi(low_x, 0, nx - 1); low_y = clampi(low_y, 0, ny - 1);
        high_x = clampi(high_x, 0, nx - 1); high_y = clampi(high_y, 0, ny - 1);
        for (int cy = low_y; cy <= high_y; ++cy)
            for (int cx = low_x; cx <= high_x; ++cx) {
                const uint32_t s = lds_u16(cells, (uint32_t)(cy * nx + cx));
                if (s == 0xFFFFu) continue;
                collide_with_string(P, D, s, dir_len_xy_sqr, ph, step_len, hit, hit_string, hit_dom);
            }
    }
    return hit;
}

// ---- the search without STOP_PHOTONS_ON_DETECTION (SetStopDetectedPhotons(false), OpenCL.cxx:395-397): the #else / #ifndef
// branches of sparse_collision_kernel.c.cl (:85-104, :165-186, :245-253, :580-584).  Every DOM the segment enters is saved on
// the spot with the distance to it, the step is not shortened and the photon travels on.  The reference keeps a string from
// being tested twice (it lies in every cell its bounding square overlaps) and a DOM from being tested twice (several z layers
// name it) with bit masks whose bit is `1 << convert_ulong(n % 64)`: the literal is an int, so OpenCL shifts by n % 32 and
// widens the result with its sign -- strings (DOMs) n and n + 32 share a bit, and bit 31 drags bits 32..63 along, which makes
// the upper half of each 64-bit word a copy of bit 31: a word's state is its lower 32 bits.  Restated as such; the string
// words live in LDS (one per 64 strings and lane), the DOM word of a string call in a register (:86-90 index that array with
// the STRING number and size it by the DOM number: one word is in use per call, and where stringNum/64 lies beyond it the
// reference is undefined -- here, as in oracle/clsim_oracle.c, the word is there).
struct KeepSink {
    uint32_t step_index;
    const float4 *ring;                 // SAVE_PHOTON_HISTORY: the lane's ring, copied beside every hit (c.cl:387-392)
    uint32_t history_n;
    uint32_t *string_mask;              // LDS: word w of this lane at string_mask[w * mask_stride]
    uint32_t mask_stride, mask_words;
};

// c.cl:307-404 from inside the search: one atomic per hit (this is not the fast path), the stub into its slot
DM void save_hit_now(KP P, const Photon &ph, float smin1, const KeepSink &K, uint32_t s, uint32_t dom)
{
    const uint32_t index = atomicAdd(P->hit_count, 1u);                     // keeps counting past the buffer (c.cl:329-334)
    if (index >= P->max_hits) return;
    uint32_t *st = reinterpret_cast<uint32_t *>(P->out) + (size_t)index * 20u;
    st[0] = dm::f2u(ph.px); st[1] = dm::f2u(ph.py); st[2] = dm::f2u(ph.pz); st[3] = dm::f2u(ph.pt);
    st[4] = dm::f2u(ph.d.x); st[5] = dm::f2u(ph.d.y); st[6] = dm::f2u(ph.d.z); st[7] = dm::f2u(smin1);
    st[8] = dm::f2u(ph.total_path); st[9] = dm::f2u(ph.abs_lens_left); st[10] = dm::f2u(ph.inv_groupvel);
    st[11] = ph.num_scatters; st[12] = K.step_index;
    st[13] = (uint32_t)ph.rx_start; st[14] = (uint32_t)(ph.rx_start >> 32);
    st[15] = (s & 0xffffu) | (dom << 16);
    if (K.history_n != 0u) {
        float4 *dst = reinterpret_cast<float4 *>(P->hist_out) + (size_t)index * K.history_n;
        for (uint32_t k = 0; k < K.history_n; ++k) dst[k] = K.ring[k];
    }
}

// c.cl:27-192 without STOP_PHOTONS_ON_DETECTION
DM void collide_with_string_keep(KP P, const Detector &D, uint32_t s, float dir_len_xy_sqr, const Photon &ph, float step_len, const KeepSink &K)
{
    const Rec4 str = lds_rec4(D.off_strings + 8u * s);      // x, y, maxZ+R, minZ-R
    {
        const float smin = sqr((ph.px - str.a) * ph.d.y - (ph.py - str.b) * ph.d.x) / dir_len_xy_sqr;
        if (smin > D.string_max_radius_sq) return;
    }
    if ((ph.d.z > 0.0f) && (ph.pz > str.c)) return;
    if ((ph.d.z < 0.0f) && (ph.pz < str.d)) return;
    const uint32_t set = ldsu(D.off_strings + 8u * s + 4) & 0xffu;
    const Rec4 lay = lds_rec4(D.off_sets + 4u * set);        // nlayers (bits), start z, height
    const float start_z = lay.b, height = lay.c;
    const int nl = (int)__builtin_bit_cast(uint32_t, lay.a);
    int low = (int)((ph.pz - start_z) / height);
    int high = (int)((ph.pz + ph.d.z * step_len - start_z) / height);
    if (high < low) { const int tmp = low; low = high; high = tmp; }
    low = clampi(low, 0, nl - 1);
    high = clampi(high, 0, nl - 1);
    const uint32_t base = set * (uint32_t)D.max_layers;
    uint32_t dom_mask = 0u;                                    // :85-90
    for (int layer = low; layer <= high; ++layer) {
        const uint32_t dom = lds_u16(D.off_layer_to_om, base + (uint32_t)layer);
        if (dom == 0xFFFFu) continue;
        const uint32_t bit = 1u << (dom & 31u);
        if ((dom_mask & bit) != 0u) continue;                   // :103
        dom_mask |= bit;                                        // :104
        float dom_x, dom_y, dom_z;
        dom_position(P, s, dom, dom_x, dom_y, dom_z);
        const float dx = dom_x - ph.px, dy = dom_y - ph.py, dzz = dom_z - ph.pz;
        const float dr2 = (dx * dx + dy * dy) + dzz * dzz;
        const float urdot = (dx * ph.d.x + dy * ph.d.y) + dzz * ph.d.z;
        float discr = sqr(urdot) - dr2 + D.om_radius_sq;
        if (discr < 0.0f) continue;
        discr = D.has_pancake ? (dm::sqrt_(discr) / D.pancake) : dm::sqrt_(discr);
        if (urdot + discr < 0.0f) continue;
        const float smin1 = urdot - discr;
        if (smin1 < 0.0f) continue;
        if (smin1 < step_len) save_hit_now(P, ph, smin1, K, s, dom);          // :165-186
    }
}

// c.cl:194-303 + :462-587 without STOP_PHOTONS_ON_DETECTION
DM void find_collisions_keep(KP P, const Photon &ph, float step_len, const KeepSink &K)
{
    const float dir_len_xy_sqr = sqr(ph.d.x) + sqr(ph.d.y);
    if (dir_len_xy_sqr <= 0.0f) return;
    Detector D;
    D.off_strings = P->off_strings; D.off_sets = P->off_sets; D.off_layer_to_om = P->off_layer_to_om;
    D.max_layers = P->max_layers;
    D.string_max_radius_sq = P->string_max_radius_sq; D.string_max_radius = P->string_max_radius; D.om_radius_sq = P->om_radius_sq;
    D.pancake = P->pancake; D.has_pancake = P->has_pancake;
    const int num_subdet = P->num_subdet;
    const uint32_t off_subdet = P->off_subdet;
    for (int sd = 0; sd < num_subdet; ++sd) {
        const Rec4 g0 = lds_rec4(off_subdet + 12u * (uint32_t)sd);        // nx, ny (bits), width x, width y
        const Rec4 g1 = lds_rec4(off_subdet + 12u * (uint32_t)sd + 4u);   // start x, start y, cell offset, proof bits
        const Rec4 g2 = lds_rec4(off_subdet + 12u * (uint32_t)sd + 8u);   // 1/width x, 1/width y
        const int nx = (int)__builtin_bit_cast(uint32_t, g0.a), ny = (int)__builtin_bit_cast(uint32_t, g0.b);
        const float wx = g0.c, wy = g0.d, sx = g1.a, sy = g1.b;
        const uint32_t cells = __builtin_bit_cast(uint32_t, g1.c);
        const uint32_t proven = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint32_t, g1.d));
        const bool okx = (proven & 1u) != 0, oky = (proven & 2u) != 0;
        int low_x = (int)div_by_t<false>(ph.px - sx, wx, g2.a, okx);
        int low_y = (int)div_by_t<false>(ph.py - sy, wy, g2.b, oky);
        int high_x = (int)div_by_t<false>(ph.px + ph.d.x * step_len - sx, wx, g2.a, okx);
        int high_y = (int)div_by_t<false>(ph.py + ph.d.y * step_len - sy, wy, g2.b, oky);
        if (high_x < low_x) { const int tmp = low_x; low_x = high_x; high_x = tmp; }
        if (high_y < low_y) { const int tmp = low_y; low_y = high_y; high_y = tmp; }
        low_x = clampi(low_x, 0, nx - 1); low_y = clampi(low_y, 0, ny - 1);
        high_x = clampi(high_x, 0, nx - 1); high_y = clampi(high_y, 0, ny - 1);
        for (uint32_t w = 0; w < K.mask_words; ++w) K.string_mask[w * K.mask_stride] = 0u;      // :245-249
        for (int cy = low_y; cy <= high_y; ++cy)
            for (int cx = low_x; cx <= high_x; ++cx) {
                const uint32_t s = lds_u16(cells, (uint32_t)(cy * nx + cx));
                if (s == 0xFFFFu) continue;
                uint32_t *word = K.string_mask + (s >> 6) * K.mask_stride;
                const uint32_t bit = 1u << (s & 31u);
                const uint32_t seen = *word;
                if ((seen & bit) != 0u) continue;               // :252
                *word = seen | bit;                             // :253
                collide_with_string_keep(P, D, s, dir_len_xy_sqr, ph, step_len, K);
            }
    }
}

// A detected photon leaves the propagation kernel as a 16-word stub written into its 80-byte output
// slot; assemble_hits_kernel expands it in place into the I3CLSimPhoton record.  Everything saveHit
// (propagation_kernel.c.cl:307-404) stores is a function of the stub: the birth of the photon is
// re-derived from the step and the RNG state at its creation (photon_birth), the rest is arithmetic.
struct HitStub {
    float px, py, pz, pt;               // photon at the start of its last segment
    float dx, dy, dz;
    float step_len;                     // distance to the DOM along the direction (shortened step)
    float total_path, abs_lens_left, inv_groupvel;
    uint32_t num_scatters;
    uint32_t step_index;
    uint32_t rx_lo, rx_hi;              // RNG state word at the photon's creation
    uint32_t string_and_dom;            // string index | DOM index << 16
};
static_assert(sizeof(HitStub) == 64, "hit stub");

// propagation_kernel.c.cl:307-404: the 20 words of an I3CLSimPhoton
template <bool FLASHER>
DM float make_hit_record(KP P, const HitStub &h, uint32_t *rec)
{
    const DevStep *step_ptr = P->steps + h.step_index;
    const Vec3 step_dir = step_direction(step_ptr);
    uint64_t rx = ((uint64_t)h.rx_hi << 32) | (uint64_t)h.rx_lo;
    const Birth born = photon_birth<FLASHER>(P, step_ptr, step_dir, rx, P->rng_a[h.step_index]);
    const uint32_t hit_string = h.string_and_dom & 0xffffu, hit_dom = h.string_and_dom >> 16;
    const Vec3 d = {h.dx, h.dy, h.dz};
    float dom_x, dom_y, dom_z;
    dom_position(P, hit_string, hit_dom, dom_x, dom_y, dom_z);
    if (P->has_pancake) {
        const float unpancake = P->unpancake;
        const float qx = h.px - dom_x, qy = h.py - dom_y, qz = h.pz - dom_z;
        const float parallel = qx * d.x + qy * d.y + qz * d.z;
        const float nx = qx - parallel * d.x;
        const float ny = qy - parallel * d.y;
        const float nz = qz - parallel * d.z;
        dom_x += unpancake * nx; dom_y += unpancake * ny; dom_z += unpancake * nz;
    }
    float theta, phi, stheta, sphi;
    sph_dir_from_car(d, theta, phi);
    sph_dir_from_car(born.d, stheta, sphi);
    const float weight = step_ptr->weight / wavelength_bias(P, born.wlen);
    rec[0] = dm::f2u(h.px + h.step_len * d.x - dom_x);
    rec[1] = dm::f2u(h.py + h.step_len * d.y - dom_y);
    rec[2] = dm::f2u(h.pz + h.step_len * d.z - dom_z);
    rec[3] = dm::f2u(h.pt + h.step_len * h.inv_groupvel);
    rec[4] = dm::f2u(theta);
    rec[5] = dm::f2u(phi);
    rec[6] = dm::f2u(born.wlen);
    rec[7] = dm::f2u(h.total_path + h.step_len);
    rec[8] = h.num_scatters;
    rec[9] = dm::f2u(weight);
    rec[10] = step_ptr->identifier;
    rec[11] = h.string_and_dom;                             // short stringID, ushort omID
    rec[12] = dm::f2u(born.x);
    rec[13] = dm::f2u(born.y);
    rec[14] = dm::f2u(born.z);
    rec[15] = dm::f2u(born.t);
    rec[16] = dm::f2u(stheta);
    rec[17] = dm::f2u(sphi);
    rec[18] = dm::f2u(1.0f / h.inv_groupvel);
    rec[19] = dm::f2u(born.abs_lens_initial - h.abs_lens_left);   // c.cl:718: after this step's update
    return born.abs_lens_initial;
}

// `count` staged 64-byte stubs of a wave into the photon buffer: one atomic on the hit counter, then every stub as one
// contiguous run of 16 dwords into its 80-byte slot, by all 64 lanes.  The counter keeps counting past max_hits; only the
// first max_hits arrivals are stored (c.cl:329-334).
DM uint32_t flush_hit_stubs(KP P, const uint32_t *stage, uint32_t count, uint32_t lane)      // returns the first stub's index
{
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(P->hit_count, count);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t max_hits = P->max_hits;
    const uint32_t room = (base < max_hits) ? (max_hits - base) : 0u;
    const uint32_t words = ((count < room) ? count : room) * (uint32_t)kStubWords;
    uint32_t *dst = reinterpret_cast<uint32_t *>(P->out) + (size_t)base * 20u;
    for (uint32_t w = lane; w < words; w += 64u) dst[(w >> 4) * 20u + (w & 15u)] = stage[w];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return base;
}

} // namespace clsimhip
