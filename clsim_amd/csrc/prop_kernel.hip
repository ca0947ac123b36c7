// Photon propagator for gfx950 (MI355X).
//
// Replaces the reference's run-time generated OpenCL program
//   resources/kernels/propagation_kernel.c.cl:406-913       (propKernel)
//   resources/kernels/sparse_collision_kernel.c.cl:27-587   (DOM intersection)
//   resources/kernels/mwcrng_kernel.cl:12-28                (MWC RNG)
//   + generated medium / spectrum / geometry functions (I3CLSimHelperGenerate*.cxx)
// with a hand-written kernel whose results are bit-identical for every step and
// RNG stream, but which is organised for the CDNA4 execution model:
//
//  * persistent workgroups (5-7 per CU, chosen from the bunch size) pull work units
//    from eight sub-queues: a unit is a slice of a step's photons, handed out
//    round-robin over the bunch, so that all steps advance together; a lane that
//    finishes a unit takes the next one instead of idling until the slowest of its
//    64 neighbours is done.  The RNG stream travels with the step, in a 64-byte
//    work record (propagation_kernel.c.cl:458-461, 911-912), never with the lane;
//  * the waves of a SIMD take turns at the issue priorities (s_setprio): the arbiter
//    alone serves the oldest wave first, which let young waves crawl and hold slices
//    that others wait for;
//  * one in-flight photon per lane; the scatter loop is a WAVE-UNIFORM loop
//    (ballot), so hit records are emitted at a convergent point by the whole wave;
//  * rare, heavy phases are batched: photon creation -- 1/29 of a lane's
//    iterations but paid by the whole wave whenever one lane needs it -- waits
//    until k_new lanes need it; the DOM search is skipped for steps that end before
//    the nearest string (proximity map) and the lanes that do need it park until
//    k_search of them do;
//  * hit write-out is wave-aggregated: one atomic per wave claims the slots,
//    records are staged in LDS and written as contiguous dwords by all lanes;
//  * ice layer records, tilt grid, spectra and the DOM cell/string/layer index are
//    staged in LDS once per workgroup (lanes index them divergently); wave-uniform
//    scalars are read from the kernarg segment with scalar loads next to their
//    use (keeping ~150 of them live in SGPRs spills into VGPR lanes);
//  * wavelength-only factors of the ice functions (lambda^-alpha, lambda^-kappa,
//    A*exp(-B/lambda)) are evaluated once per photon instead of once per layer
//    visit -- same operations on the same inputs, so the same bits;
//  * no MFMA: nothing here is a contraction.  The kernel is bound by fp32 VALU
//    issue (IEEE divides, polynomial transcendentals) and divergence, not HBM.
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (no implicit fma; all
// fused operations are explicit in detmath.hip.h).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>

#include "prop_device.hip.h"

namespace clsimhip {

// ---------------- TABULATE (c.cl:228-303) ----------------
DM float dot4(float ax, float ay, float az, float aw, float bx, float by, float bz, float bw)
{
    return ((ax * bx + ay * by) + az * bz) + aw * bw;
}
// Records the samples of one path segment: for d = remainder, remainder + step, ... < length the bin of the
// point gets weight * exp(-depth(d)).  The reference writes (index, weight) entries into a per-stream buffer that the
// host adds up (tabulator/I3CLSimStepToTableConverter.cxx:495-507) and re-runs streams whose buffer overflowed; here
// every sample goes straight into its bin with one hardware double-precision atomic add, so there is no buffer to
// overflow.  Returns true when the photon left the table (isOutOfBounds): it is dropped (c.cl:781-784).
// One path sample (the body of the loop of c.cl:256-287): the table bin of the point at distance d along the segment
// and whether it is out of bounds (isOutOfBounds, Axes.cxx:104-116, 140-151).
// LDS record at off_tab (tabulator.cpp): [0..4] scale, [5..9] offset, [10..14] bins, [15..19] stride, [20..24] sqrt axis,
// [25] max of axis 0, [26] max of axis 3, [27] min_invGroupVel, [28] tan_thetaC, [29] VOLUME_MODE_STEP, [30] dimensions
struct Segment { float px, py, pz, pt, dx, dy, dz, igv, wlen; };
// The wave-uniform constants of a path sample, read ONCE per savePath call into scalar registers (round 5, second half: the
// sample loop used to read each where it needed it -- some twenty scalar loads per 64 samples, every one followed by its own
// s_waitcnt: the loop was waiting for the scalar cache, not for its atomics; profiles/r05/ab_tab_bound.txt).
struct TabK {
    float ref[12];
    int32_t kind, full_azimuth;
    float scale[5], offset[5], inv_exp[5];
    int32_t inverse[5], nbins[5];
    uint32_t stride[5];
    uint32_t tiled, tile_stride[3], tile_bits[3];
    float max0, max3, min_inv_groupvel, tan_thetac;
};
template <bool ANGLE>
DM TabK tab_constants(KP P)
{
    TabK K;
#pragma unroll
    for (int k = 0; k < 12; ++k) K.ref[k] = P->tab_ref[k];
    K.kind = P->tab_axes_kind;
    K.full_azimuth = P->tab_full_azimuth;
    constexpr int ndim = ANGLE ? 5 : 4;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const bool used = k < ndim;
        K.scale[k] = used ? P->tab_scale[k] : 0.0f;
        K.offset[k] = used ? P->tab_offset[k] : 0.0f;
        K.inv_exp[k] = used ? P->tab_inv_exp[k] : 0.0f;
        K.inverse[k] = used ? P->tab_inverse[k] : 0;
        K.nbins[k] = used ? P->tab_nbins[k] : 0;
        K.stride[k] = used ? P->tab_stride[k] : 0u;
    }
    K.tiled = ANGLE ? 0u : P->tab_tiled;
#pragma unroll
    for (int k = 0; k < 3; ++k) { K.tile_stride[k] = ANGLE ? 0u : P->tab_tile_stride[k]; K.tile_bits[k] = ANGLE ? 0u : P->tab_tile_bits[k]; }
    K.max0 = P->tab_max0;
    K.max3 = P->tab_max3;
    K.min_inv_groupvel = P->tab_min_inv_groupvel;
    K.tan_thetac = P->tab_tan_thetac;
    return K;
}
// Axis::GetIndexCode (Axes.cxx:69-90, Axis.cxx:45-60): clamp(convert_int_sat_rtn(t), -1, n) + 1 for t = scale * inverse(x) - offset.
// axis_bin_generic_ spells the saturating floor conversion out (NaN -> 0, beyond the int range -> its ends); axis_bin_ is the same
// function in four instructions -- v_cvt_flr_i32_f32 floors and saturates by itself -- which
// clsimhip_check_math_exhaustive(19) compares on ALL 2^32 bit patterns on the device (tests/test_detmath_gpu.py).
DM uint32_t axis_bin_generic_(float t, int nbins)
{
    const float f = __builtin_floorf(t);
    const int b = (f != f) ? 0 : ((f >= 2147483648.0f) ? 2147483647 : ((f < -2147483648.0f) ? (-2147483647 - 1) : (int)f));
    return (uint32_t)(clampi(b, -1, nbins) + 1);
}
DM uint32_t axis_bin_(float t, int nbins)
{
    int b, r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(b) : "v"(t));
    b = (t != t) ? 0 : b;                                               // (the instruction does not send NaN to 0: measured)
    asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(b), "s"(nbins));
    return (uint32_t)(r + 1);
}
// FASTMATH (round 5): the sample's square roots and quotients through the range-restricted exact forms of detmath.hip.h (sqrt_near_: 8
// instructions for the IEEE sequence's 17; div_near_: 8 for 11, and no VCC), which return the IEEE results on their admitted ranges;
// `ok` comes back false for a lane with an operand outside them (a sample exactly on the table's axis or in its plane, a negative delay
// time), and the caller then runs the IEEE flavour for the whole wave (a wave-uniform decision: one batch in a few hundred).
DM bool sqrt_near_ok_(float x)      // +0, or 2^-96 ... 2^100
{
    const uint32_t u = dm::f2u(x);
    return (u == 0u) || ((u - 0x0f800000u) <= (0x71800000u - 0x0f800000u));
}
template <bool FASTMATH> DM float tab_sqrt_(float x, bool &ok)
{
    if (FASTMATH) { ok = ok && sqrt_near_ok_(x); return dm::sqrt_near_(x); }
    return dm::sqrt_(x);
}
template <bool FASTMATH> DM float tab_div_(float a, float b, bool &ok)       // (|b| within 2^-50 ... 2^50 wherever ok stays true)
{
    if (FASTMATH) { ok = ok && dm::div_near_ok_(a) && (__builtin_fabsf(a) <= 1.152921504606847e18f); return dm::div_near_(a, b); }
    return a / b;
}
// (x, a): the photon's stream before the two draws of this sample (ANGLE = TABULATE_IMPACT_ANGLE only)
// STD (round 6): the configuration python/tablemaker/tabulator.py:621-641 makes by default -- spherical axes, azimuth folded to 180 degrees,
// square-root axes for distance and time, linear ones for the two angles, the tiled device order with 4 x 2 x 1 bins to a sector (KParams::tab_std,
// set by tabulator.cpp).  The generic sampler asks about each of these once per batch of 64 samples -- wave-uniform, so every question is a scalar
// compare and a branch, and the code of every answer (cube roots, fractional powers, the cylindrical formulas) sits in the loop: 95 branches and
// 360 scalar instructions per batch next to 745 vector ones (profiles/r06/tab_scalar_summary.json), at three waves per SIMD.  With STD they are
// constants and the other answers' code is gone.  Same arithmetic, same bins.
template <bool ANGLE, bool FASTMATH, bool STD = false>
DM bool sample_bin(const TabK &K, const Segment &g, float d, uint64_t x, uint32_t a, uint32_t &index, bool &ok)
{
    static_assert(!(STD && ANGLE), "the standard configuration has four axes");
    const int kind = STD ? 0 : K.kind;
    const int full_azimuth = STD ? 0 : K.full_azimuth;
    auto R = [&](int k) { return K.ref[k]; };
    // spherical_coordinates.c.cl:39-81 / cylindrical_coordinates.c.cl:39-77
    const float ax = g.px + d * g.dx, ay = g.py + d * g.dy, az = g.pz + d * g.dz, aw = g.pt + d * g.igv;
    const float px = ax - R(0), py = ay - R(1), pz = az - R(2);
    const float pw = aw - R(3);
    const float ux = R(4), uy = R(5), uz = R(6), uw = R(7), qx = R(8), qy = R(9), qz = R(10), qw = R(11);
    const float l = dot4(px, py, pz, pw, ux, uy, uz, uw);
    const float rx_ = px - l * ux, ry_ = py - l * uy, rz_ = pz - l * uz, rw_ = pw - l * uw;
    const float n_rho = tab_sqrt_<FASTMATH>(rx_ * rx_ + ry_ * ry_ + rz_ * rz_, ok);
    constexpr int ndim = ANGLE ? 5 : 4;
    constexpr float kDegree = kPi / 180;
    float c0, c1, c2, c3, c4 = 0.0f;
    if (kind == 0) {
        c0 = tab_sqrt_<FASTMATH>(px * px + py * py + pz * pz, ok);
        float azimuth = 0.0f;
        if (n_rho > 0.0f) {
            const float angle = dm::acos_f(tab_div_<FASTMATH>(dot4(rx_, ry_, rz_, rw_, qx, qy, qz, qw), n_rho, ok));
            if (FASTMATH) { ok = ok && dm::div_near_ok_(angle); azimuth = dm::div_near_with_(angle, kDegree, 1.0f / kDegree); }
            else azimuth = angle / kDegree;
        }
        if (full_azimuth) {
            const float cx = ry_ * qz - rz_ * qy, cy = rz_ * qx - rx_ * qz, cz = rx_ * qy - ry_ * qx;
            const float azisign = dot4(cx, cy, cz, 0.0f, ux, uy, uz, uw);
            c1 = (azisign > 0.0f) ? 360.f - azimuth : azimuth;
        } else {
            c1 = azimuth;
        }
        c2 = (c0 > 0.0f) ? tab_div_<FASTMATH>(l, c0, ok) : 0.0f;
        c3 = pw - c0 * K.min_inv_groupvel;
    } else {
        c0 = n_rho;
        c1 = (c0 > 0.0f) ? dm::acos_f(tab_div_<FASTMATH>(dot4(rx_, ry_, rz_, rw_, qx, qy, qz, qw), c0, ok)) : 0.0f;
        c2 = R(2) + l * uz;
        c3 = pw - (l + c0 * K.tan_thetac) * 3.33564095f;
    }
    if (ANGLE) {
        // TABULATE_IMPACT_ANGLE (spherical :67-79, cylindrical :61-76): drawn before the bounds check, like the reference
        const float sina = dm::sqrt_(rng_co(x, a));
        Vec3 dd = {g.dx, g.dy, g.dz};
        scatter_direction(dm::sqrt_(1.0f - sina * sina), sina, dd, rng_co(x, a));
        if (kind == 0) {
            c4 = (c0 > 0.0f) ? tab_div_<FASTMATH>(dot4(dd.x, dd.y, dd.z, g.wlen, px, py, pz, pw), c0, ok) : 1.0f;
        } else {
            // (l - rho*recip(tan_thetaC))*dir, component by component as OpenCL evaluates it
            const float rt = 1.0f / K.tan_thetac;
            const float kx = ax - (R(0) + (l - rx_ * rt) * ux), ky = ay - (R(1) + (l - ry_ * rt) * uy);
            const float kz = az - (R(2) + (l - rz_ * rt) * uz), kw = aw - (R(3) + (l - rw_ * rt) * uw);
            const float cdist = dm::sqrt_(kx * kx + ky * ky + kz * kz);
            c4 = (cdist > 0.0f) ? (dot4(dd.x, dd.y, dd.z, g.wlen, kx, ky, kz, kw) / cdist) : 1.0f;
        }
    }
    if (kind == 0) {
        if ((c3 > K.max3) || (c0 > K.max0)) return true;
    } else {
        if (c3 > K.max3) return true;
    }
    const float c[5] = {c0, c1, c2, c3, c4};
    uint32_t bin[5] = {0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < ndim; ++k) {
        const int pw = STD ? ((k == 0 || k == 3) ? 2 : 1) : K.inverse[k];                // wave-uniform
        const float v = (pw <= 1) ? c[k] : (pw == 2) ? tab_sqrt_<FASTMATH>(c[k], ok) : (pw == 3) ? dm::cbrt_(c[k]) : dm::pow_frac_(c[k], K.inv_exp[k]);
        bin[k] = axis_bin_(K.scale[k] * v - K.offset[k], K.nbins[k]);
    }
    if (!ANGLE && (STD || K.tiled)) {
        // the device's own order (kparams.h: tab_tiled): 2^e0 x 2^e2 x 2^e3 = 8 bins of distance, polar angle and time in one 64-byte sector
        const uint32_t e0 = STD ? 2u : K.tile_bits[0], e2 = STD ? 1u : K.tile_bits[1], e3 = STD ? 0u : K.tile_bits[2];
        const uint32_t h0 = bin[0] >> e0, h2 = bin[2] >> e2, h3 = bin[3] >> e3;
        index = h0 * K.tile_stride[0] + bin[1] * K.tile_stride[1] + h2 * K.tile_stride[2] + (h3 << 3)
                + (((bin[0] - (h0 << e0)) << (e2 + e3)) | ((bin[2] - (h2 << e2)) << e3) | (bin[3] - (h3 << e3)));
    } else {
        index = 0;
#pragma unroll
        for (int k = 0; k < ndim; ++k) index += K.stride[k] * bin[k];
    }
    return false;
}
// Lane shifts of the sample loop's segmented sum as DPP moves (row_shr 1, 2, 4, 8, then the row broadcasts 15 and 31; lanes without
// a source read zero): a step is three register moves where __shfl_up is three trips through the LDS crossbar, and the six steps are
// one dependent chain.
template <int CTRL, int ROW_MASK>
DM uint32_t dpp_zero_(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
template <int CTRL, int ROW_MASK>
DM void segmented_step_(double &sum, double &sum_sq, int &flag, bool squares)
{
    const uint64_t bits = __builtin_bit_cast(uint64_t, sum);
    const uint32_t lo = dpp_zero_<CTRL, ROW_MASK>((uint32_t)bits), hi = dpp_zero_<CTRL, ROW_MASK>((uint32_t)(bits >> 32));
    const int up_flag = (int)dpp_zero_<CTRL, ROW_MASK>((uint32_t)flag);
    const double up = __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
    double up_sq = 0.0;
    if (squares) {
        const uint64_t sb = __builtin_bit_cast(uint64_t, sum_sq);
        const uint32_t slo = dpp_zero_<CTRL, ROW_MASK>((uint32_t)sb), shi = dpp_zero_<CTRL, ROW_MASK>((uint32_t)(sb >> 32));
        up_sq = __builtin_bit_cast(double, (uint64_t)slo | ((uint64_t)shi << 32));
    }
    if (!flag) { sum += up; sum_sq += up_sq; flag = up_flag; }
}
DM void add_to_bin(double *bins, double *sq_bins, uint32_t index, float w)
{
    unsafeAtomicAdd(bins + index, (double)w);
    if (sq_bins) unsafeAtomicAdd(sq_bins + index, (double)w * (double)w);
}

// savePath for a whole wave (called by all 64 lanes; `active` lanes bring one path segment each).
// The reference walks each segment in its own work item: d = remainder; while (d < length) { sample(d); d += step; }
// and drops the photon at the first sample that is out of bounds.  Segment lengths are exponentially distributed, so a
// wave that lets every lane walk its own segment runs the longest walk with a fifth of its lanes busy.  Here the
// wave pools its samples: every lane lists its d values (the same repeated float additions) in LDS, the pooled
// samples are evaluated 64 at a time by whichever lanes, and a sample is added to the table unless its segment
// went out of bounds at an earlier sample.  Bins and weights are those of the per-lane walk, bit for bit.
// Returns true for lanes whose photon left the table.
// Keeps a wave-uniform value where it is (a scalar register, loaded here): without it the compiler sinks each parameter load to its
// first use, and the prologue below becomes a chain of scalar loads that each wait for the scalar cache.
template <typename T>
DM T here_(T v)
{
    asm volatile("" : "+s"(v));
    return v;
}
template <bool ANGLE>
DM bool save_path_wave(KP P, uint32_t *wave_lds, bool active, const Photon &ph, float weight,
                       float length, float &remainder, float depth, float this_depth, uint64_t &rx, uint32_t ra
                       TAB_TIMED(, uint64_t &t_list, uint64_t &t_last)
                       )
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    const float vstep = here_(P->tab_volume_step);
    // ANGLE = TABULATE_IMPACT_ANGLE: every sample draws two numbers from the photon's stream (its impact point on the
    // DOM), the angular acceptance is a table axis instead of a weight (c.cl:246-251)
    constexpr bool angle_axis = ANGLE;
    float impact = active ? weight : 0.0f;
    if (!angle_axis) {
        // getAngularAcceptance (Polynomial.cxx:96-153), its parameters read in one go
        const int has_min = here_(P->ang_has_min), has_max = here_(P->ang_has_max), n_coeff = here_(P->ang_n);
        const float a_min = here_(P->ang_min), a_max = here_(P->ang_max), a_under = here_(P->ang_underflow), a_over = here_(P->ang_overflow);
        const uint32_t off = here_(P->off_ang);
        const float x = ph.d.z;
        float r = 0.0f;
        if (n_coeff > 0) {
            r = ldsf(off + (uint32_t)(n_coeff - 1));
            for (int i = n_coeff - 2; i >= 0; --i) r = ldsf(off + (uint32_t)i) + x * r;
        }
        if (has_max && x > a_max) r = a_over;
        if (has_min && x < a_min) r = a_under;
        impact = active ? weight * r : 0.0f;
    }
    // number of samples and the value d ends with
    uint32_t n = 0;
    float d_end = remainder;
    // (the cap only guards the GPU against a walk that cannot advance, d + step == d; the reference's own walk ends
    // after TABLE_ENTRIES_PER_STREAM = 5000 samples of the whole step)
    if (active) for (; (d_end < length) && (n < (1u << 16)); d_end += vstep) ++n;
    // inclusive prefix sum over the wave, as DPP moves (row shifts 1, 2, 4, 8, then the row broadcasts)
    uint32_t incl = n;
    incl += dpp_zero_<0x111, 0xf>(incl);
    incl += dpp_zero_<0x112, 0xf>(incl);
    incl += dpp_zero_<0x114, 0xf>(incl);
    incl += dpp_zero_<0x118, 0xf>(incl);
    incl += dpp_zero_<0x142, 0xa>(incl);
    incl += dpp_zero_<0x143, 0xc>(incl);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t first = incl - n;
    // pool layout: d and owner per sample, plus the stream state before the sample's draws when there is an angle axis
    // (then the pool holds half as many samples)
    const uint32_t slots = angle_axis ? (uint32_t)kTabSlots / 2u : (uint32_t)kTabSlots;
    const uint64_t rx_before = rx;
    bool stop = false;
    if (total == 0u) {
        // nothing to record
    } else if (total > slots) {
        const TabK K = tab_constants<ANGLE>(P);
        double *const bins = P->tab_bins, *const sq_bins = P->tab_sq_bins;
        // (rare) more samples than the pool holds: every lane walks its own segment
        if (active) {
            const Segment g = {ph.px, ph.py, ph.pz, ph.pt, ph.d.x, ph.d.y, ph.d.z, ph.inv_groupvel, ph.tab_wlen};
            float d = remainder;
            for (uint32_t taken = 0; (d < length) && (taken < n); d += vstep, ++taken) {
                uint32_t index;
                const uint64_t x_sample = rx;
                if (angle_axis) { (void)rng_co(rx, ra); (void)rng_co(rx, ra); }
                bool ok_ = true;
                if (sample_bin<ANGLE, false>(K, g, d, x_sample, ra, index, ok_)) { stop = true; break; }
                add_to_bin(bins, sq_bins, index, impact * dm::exp_(-(depth + (d / length) * this_depth)));
            }
            d_end = d;
        }
    } else {
        const TabK K = tab_constants<ANGLE>(P);
        double *const bins = P->tab_bins, *const sq_bins = P->tab_sq_bins;
        const bool squares = (sq_bins != nullptr);
        uint32_t *slot_d = wave_lds, *slot_owner = wave_lds + slots, *slot_xlo = wave_lds + 2u * slots, *slot_xhi = wave_lds + 3u * slots;
        if (active) {
            float d = remainder;
            for (uint32_t j = 0; j < n; ++j, d += vstep) {
                slot_d[first + j] = __builtin_bit_cast(uint32_t, d);
                slot_owner[first + j] = lane | (j << 8);
                if (angle_axis) {
                    slot_xlo[first + j] = (uint32_t)rx;
                    slot_xhi[first + j] = (uint32_t)(rx >> 32);
                    (void)rng_co(rx, ra);
                    (void)rng_co(rx, ra);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        TAB_TIMED({ const uint64_t now_ = __builtin_amdgcn_s_memtime(); t_list += now_ - t_last; t_last = now_; })
        // Out of bounds (isOutOfBounds ends the walk: c.cl:781-784) is rare -- once in a photon's life -- and is kept in
        // registers: `dead`, the wave-uniform mask of lanes whose segment has left the table at an earlier sample, and per
        // lane the index of its own segment's first such sample.
        uint64_t dead = 0ull;
        int my_first_oob = 0x7fffffff;
        // (the next 64 samples' slots are read while these 64 are worked on)
        uint32_t tag_next = (lane < total) ? slot_owner[lane] : 0u;
        uint32_t d_next = (lane < total) ? slot_d[lane] : 0u;
        for (uint32_t base = 0; base < total; base += 64u) {
            const uint32_t slot = base + lane;
            const bool have = slot < total;
            const uint32_t tag = tag_next;
            const float d = __builtin_bit_cast(float, d_next);
            if (base + 64u < total) {
                const bool more = slot + 64u < total;
                tag_next = more ? slot_owner[slot + 64u] : 0u;
                d_next = more ? slot_d[slot + 64u] : 0u;
            }
            const int owner = (int)(tag & 0xffu);
            const int j = (int)(tag >> 8);
            // the owner's segment
            Segment g;
            g.px = __shfl(ph.px, owner); g.py = __shfl(ph.py, owner); g.pz = __shfl(ph.pz, owner); g.pt = __shfl(ph.pt, owner);
            g.dx = __shfl(ph.d.x, owner); g.dy = __shfl(ph.d.y, owner); g.dz = __shfl(ph.d.z, owner);
            g.igv = __shfl(ph.inv_groupvel, owner);
            g.wlen = 0.0f;
            uint64_t x_sample = 0;
            uint32_t a_sample = 0;
            if (angle_axis) {
                g.wlen = __shfl(ph.tab_wlen, owner);
                a_sample = (uint32_t)__shfl((int)ra, owner);
                if (have) x_sample = (uint64_t)slot_xlo[slot] | ((uint64_t)slot_xhi[slot] << 32);
            }
            const float o_length = __shfl(length, owner), o_depth = __shfl(depth, owner), o_this = __shfl(this_depth, owner);
            const float o_impact = __shfl(impact, owner);
            uint32_t index = 0;
            bool oob = false;
            // the weight's quotient with the sample's (c.cl:270-272)
            bool ok = true;
            float along = 0.0f;
            if (have) {
                oob = sample_bin<ANGLE, true>(K, g, d, x_sample, a_sample, index, ok);
                along = tab_div_<true>(d, o_length, ok);
                ok = ok && (o_length <= 1.125899906842624e15f);       // (2^50; a segment is longer than its samples' d)
            }
            if (__builtin_expect(ballot(!ok) != 0ull, 0)) {
                // some lane's operand lies outside the exact forms' ranges: the IEEE sequences for the whole wave
                if (have) {
                    oob = sample_bin<ANGLE, false>(K, g, d, x_sample, a_sample, index, ok);
                    along = d / o_length;
                }
            }
            bool commit = have && !oob;
            const uint64_t m_oob = ballot(oob);
            if (__builtin_expect((m_oob | dead) != 0ull, 0)) {
                // a segment's samples sit on neighbouring lanes in walking order: the lanes of my segment before me are
                // [lane - j, lane) as far as they belong to this batch
                const uint32_t start = (lane > (uint32_t)j) ? lane - (uint32_t)j : 0u;
                const uint64_t mine_before = lanes_below & ~((1ull << start) - 1ull);
                commit = commit && ((m_oob & mine_before) == 0ull) && (((dead >> owner) & 1ull) == 0ull);
                for (uint64_t m = m_oob; m != 0ull; m &= m - 1ull) {
                    const int l = __builtin_ctzll(m);
                    const int o = __builtin_amdgcn_readlane(owner, l);
                    if (((dead >> o) & 1ull) == 0ull) {          // this segment's first sample out of bounds
                        dead |= 1ull << o;
                        const int jj = __builtin_amdgcn_readlane(j, l);
                        if ((int)lane == o) my_first_oob = jj;
                    }
                }
            }
            // Consecutive samples of a segment fall into the same bin 60 % of the time: equal-bin neighbours are summed
            // in the wave first (segmented scan over the lanes, in double: sums of a few floats are exact there) and
            // the last lane of each run issues the atomic.  2.5x fewer read-modify-writes on the 670 MB table.
            const float w = commit ? o_impact * dm::exp_(-(o_depth + along * o_this)) : 0.0f;
            const uint32_t key = commit ? index : 0xffffffffu;
            // (wave_shr:1 / wave_shl:1; the lane without a neighbour keeps a key that is not its own)
            const uint32_t prev_key = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x138, 0xf, 0xf, false);
            const uint32_t next_key = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x130, 0xf, 0xf, false);
            int flag = ((prev_key != key) || !commit) ? 1 : 0;     // first lane of its run
            double sum = (double)w, sum_sq = (double)w * (double)w;
            segmented_step_<0x111, 0xf>(sum, sum_sq, flag, squares);
            segmented_step_<0x112, 0xf>(sum, sum_sq, flag, squares);
            segmented_step_<0x114, 0xf>(sum, sum_sq, flag, squares);
            segmented_step_<0x118, 0xf>(sum, sum_sq, flag, squares);
            segmented_step_<0x142, 0xa>(sum, sum_sq, flag, squares);
            segmented_step_<0x143, 0xc>(sum, sum_sq, flag, squares);
            if (commit && (next_key != key)) {
                unsafeAtomicAdd(bins + index, sum);
                if (squares) unsafeAtomicAdd(sq_bins + index, sum_sq);
            }
        }
        if (active && (my_first_oob != 0x7fffffff)) {
            stop = true;
            d_end = __builtin_bit_cast(float, slot_d[first + (uint32_t)my_first_oob]);
            if (angle_axis) {
                // the walk ended at sample my_first_oob, whose two draws were made: the stream stands behind them
                rx = rx_before;
                for (int k = 0; k <= my_first_oob; ++k) { (void)rng_co(rx, ra); (void)rng_co(rx, ra); }
            }
        }
        // (the lists are this wave's own and the next trip writes them again: its reads above have to be done first)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (active) remainder = d_end - length;
    return stop;
}

// savePath for a whole wave, four-axis tables (round 5, second half; re-cut in round 6).  As above, but a segment's samples wait for the
// samples of the SAME LANE's next segment and the two are worked off together, neighbours in the pool.
//
// Why (round 6, profiles/r06/tab_atomics_counters.json): the table maker sits on the memory side's atomic request rate -- the L2 hands every
// one of the table's fp64 adds on (TCC_EA0_ATOMIC), one request per wave instruction and 64-byte sector, 4.4e10 per pass at 2.24e10 per
// second, which IS what the memory side delivers (2.06e10 in tools/micro/atomic_rate.hip).  A batch of 64 samples held some thirty segments of
// thirty different photons and met thirty sectors.  A photon's next segment starts where the last one ended: two segments in a row touch
// 0.42 sectors per sample where one touches 0.61 (oracle, profiles/r06/tab_requests_per_sample.txt).  So: one trip only notes its segment
// (record in LDS, first sample and count in two registers), the next trip lists the noted samples and its own, lane by lane, and works
// off everything.  Round 5's version carried the samples a trip's last batch left empty into the next trip instead (full batches only);
// full batches are worth less than fewer requests now that the requests are known to be the bound, and a carried rest would need the
// records of four trips (12 KB of the 9.5 a wave has).
//   * a sample names its segment by a record in LDS (two generations of 64 records, alternating per trip) instead of by its owner's
//     registers, which have moved on by the next trip;
//   * leaving the table must be known in the trip it happens (the photon is dropped and its stream is not drawn from again,
//     c.cl:781-784), so a trip may only note its samples when every one of its segments is CERTAINLY inside the table: the far end of
//     the segment stays below the distance axis' end and its latest delay time below the time axis' end, each with a margin four
//     orders of magnitude above the rounding of the sample's own arithmetic (conservative in one direction: a trip with a segment
//     that fails the test -- the last trip or two of a photon's life, every trip of a cylindrical table -- works off what it has at
//     once).  A noted sample is therefore never out of bounds.
// `held_n`, `held_d0`: the lane's noted segment (its record is in the other generation); `parity`: wave-uniform, the generation this trip's
// records go to; flush: work off what is noted (after the wave's last trip).
template <bool STD>
DM bool save_path_wave_carry(KP P, uint32_t *wave_lds, bool active, const Photon &ph, float weight, float length, float &remainder,
                             float depth, float this_depth, uint32_t &held_n, float &held_d0, uint32_t &parity, bool flush
                             TAB_TIMED(, uint64_t &t_list, uint64_t &t_last, uint64_t &t_add)
                             )
{
    typedef float row_t __attribute__((ext_vector_type(4)));
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    const float vstep = here_(P->tab_volume_step);
    float impact = active ? weight : 0.0f;
    {
        // getAngularAcceptance (Polynomial.cxx:96-153), its parameters read in one go
        const int has_min = here_(P->ang_has_min), has_max = here_(P->ang_has_max), n_coeff = here_(P->ang_n);
        const float a_min = here_(P->ang_min), a_max = here_(P->ang_max), a_under = here_(P->ang_underflow), a_over = here_(P->ang_overflow);
        const uint32_t off = here_(P->off_ang);
        const float x = ph.d.z;
        float r = 0.0f;
        if (n_coeff > 0) {
            r = ldsf(off + (uint32_t)(n_coeff - 1));
            for (int i = n_coeff - 2; i >= 0; --i) r = ldsf(off + (uint32_t)i) + x * r;
        }
        if (has_max && x > a_max) r = a_over;
        if (has_min && x < a_min) r = a_under;
        impact = active ? weight * r : 0.0f;
    }
    uint32_t n = 0;
    float d_end = remainder;
    if (active) for (; (d_end < length) && (n < (1u << 16)); d_end += vstep) ++n;
    const TabK K = tab_constants<false>(P);
    uint32_t *pool_d = wave_lds, *pool_tag = wave_lds + kTabPool;
    uint32_t *records = wave_lds + 2 * kTabPool;
    const uint32_t my_record = (parity << 6) | lane, my_held_record = ((parity ^ 1u) << 6) | lane;
    const bool held_any = ballot(held_n != 0u) != 0ull;
    // is every segment of this trip certainly inside the table?  (spherical axes; see above)
    bool note = !flush && !held_any;
    if (note) {
        bool inside = false;
        if (STD || K.kind == 0) {
            const float qx = ph.px - K.ref[0], qy = ph.py - K.ref[1], qz = ph.pz - K.ref[2];
            const float r0 = __builtin_amdgcn_sqrtf(qx * qx + qy * qy + qz * qz);
            const float far = (r0 + length) * 1.0001f + 0.01f;
            const float near = __builtin_fmaxf((r0 - length) * 0.9999f - 0.01f, 0.0f);
            const float t_end = (ph.pt - K.ref[3]) + length * ph.inv_groupvel;
            const float latest = (t_end + 1.0e-4f * __builtin_fabsf(t_end) + 0.01f) - near * K.min_inv_groupvel * 0.9999f;
            inside = (far < K.max0) && (latest < K.max3 - 1.0e-4f * __builtin_fabsf(K.max3) - 0.01f);
        }
        // (a segment longer than the pool could not be listed next to another one: worked off at once, by its own lane if need be)
        note = ballot(active && (n != 0u) && (!inside || (n > (uint32_t)kTabPool / 2u))) == 0ull;
    }
    if (active && (n != 0u)) {
        row_t *rec = reinterpret_cast<row_t *>(records + my_record * (uint32_t)kTabSegWords);
        rec[0] = row_t{ph.px, ph.py, ph.pz, ph.pt};
        rec[1] = row_t{ph.d.x, ph.d.y, ph.d.z, ph.inv_groupvel};
        rec[2] = row_t{length, depth, this_depth, impact};
    }
    if (note) {
        // this trip's samples wait for the lane's next segment
        held_n = n;
        held_d0 = remainder;
        parity ^= 1u;
        if (active) remainder = d_end - length;
        return false;
    }
    const uint32_t both = held_n + n;
    uint32_t incl = both;
    incl += dpp_zero_<0x111, 0xf>(incl);
    incl += dpp_zero_<0x112, 0xf>(incl);
    incl += dpp_zero_<0x114, 0xf>(incl);
    incl += dpp_zero_<0x118, 0xf>(incl);
    incl += dpp_zero_<0x142, 0xa>(incl);
    incl += dpp_zero_<0x143, 0xc>(incl);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t first = incl - both;              // the lane's noted samples, then its own
    bool stop = false;
    if (total == 0u) {
        if (active) remainder = d_end - length;      // (no lane's segment holds a sample)
        parity ^= 1u;
        return false;
    }
    double *const bins = P->tab_bins, *const sq_bins = P->tab_sq_bins;
    const bool squares = !STD && (sq_bins != nullptr);          // (STD: no squared weights, KParams::tab_std)
    // (rare) more samples than the pool holds: every lane walks its own segments, the noted one first
    const bool walk_alone = __builtin_expect(total > (uint32_t)kTabPool, 0);
    if (!walk_alone) {
        float d = held_d0;
        for (uint32_t j = 0; j < held_n; ++j, d += vstep) {
            pool_d[first + j] = __builtin_bit_cast(uint32_t, d);
            pool_tag[first + j] = my_held_record | (j << 8);
        }
        d = remainder;
        for (uint32_t j = 0; j < n; ++j, d += vstep) {
            pool_d[first + held_n + j] = __builtin_bit_cast(uint32_t, d);
            pool_tag[first + held_n + j] = my_record | (j << 8);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    TAB_TIMED({ const uint64_t now_ = __builtin_amdgcn_s_memtime(); t_list += now_ - t_last; t_last = now_; })
    const uint32_t work = walk_alone ? 0u : total;                                  // samples in the pool: all of them are worked off
    uint64_t dead = 0ull;
    int my_first_oob = 0x7fffffff;
    uint32_t tag_next = (lane < work) ? pool_tag[lane] : 0u;
    uint32_t d_next = (lane < work) ? pool_d[lane] : 0u;
    for (uint32_t base = 0; base < work; base += 64u) {
        const uint32_t slot = base + lane;
        const bool have = slot < work;
        const uint32_t tag = tag_next;
        const float d = __builtin_bit_cast(float, d_next);
        if (base + 64u < work) {
            const bool more = slot + 64u < work;
            tag_next = more ? pool_tag[slot + 64u] : 0u;
            d_next = more ? pool_d[slot + 64u] : 0u;
        }
        const uint32_t record = tag & 0x7fu;
        const int j = (int)(tag >> 8);
        const row_t *rec = reinterpret_cast<const row_t *>(records + record * (uint32_t)kTabSegWords);
        const row_t r0 = rec[0], r1 = rec[1], r2 = rec[2];
        const Segment g = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, 0.0f};
        const float o_length = r2.x, o_depth = r2.y, o_this = r2.z, o_impact = r2.w;
        uint32_t index = 0;
        bool oob = false;
        bool ok = true;
        float along = 0.0f;
        if (have) {
            oob = sample_bin<false, true, STD>(K, g, d, 0ull, 0u, index, ok);
            along = tab_div_<true>(d, o_length, ok);
            ok = ok && (o_length <= 1.125899906842624e15f);       // (2^50; a segment is longer than its samples' d)
        }
        if (__builtin_expect(ballot(!ok) != 0ull, 0)) {
            // some lane's operand lies outside the exact forms' ranges: the IEEE sequences for the whole wave
            if (have) {
                oob = sample_bin<false, false, STD>(K, g, d, 0ull, 0u, index, ok);
                along = d / o_length;
            }
        }
        bool commit = have && !oob;
        const uint64_t m_oob = ballot(oob);
        if (__builtin_expect((m_oob | dead) != 0ull, 0)) {
            // (only this trip's segments can leave the table; a segment's samples sit on neighbouring lanes in walking order)
            const bool mine = (record >> 6) == parity;
            const int owner = (int)(record & 63u);
            const uint32_t start = (lane > (uint32_t)j) ? lane - (uint32_t)j : 0u;
            const uint64_t mine_before = lanes_below & ~((1ull << start) - 1ull);
            commit = commit && !(mine && (((m_oob & mine_before) != 0ull) || (((dead >> owner) & 1ull) != 0ull)));
            for (uint64_t m = m_oob; m != 0ull; m &= m - 1ull) {
                const int l = __builtin_ctzll(m);
                const int o = __builtin_amdgcn_readlane(owner, l);
                if (((dead >> o) & 1ull) == 0ull) {          // this segment's first sample out of bounds
                    dead |= 1ull << o;
                    const int jj = __builtin_amdgcn_readlane(j, l);
                    if ((int)lane == o) my_first_oob = jj;
                }
            }
        }
        const float w = commit ? o_impact * dm::exp_(-(o_depth + along * o_this)) : 0.0f;
        const uint32_t key = commit ? index : 0xffffffffu;
        // equal-bin neighbours are summed within rows of 16 lanes (row_shr / row_shl 1: the lane at a row's end keeps a key that is
        // not its own, so a run ends there; four DPP steps instead of six, and what a run loses at a row's end -- a second atomic into
        // the same sector from the same instruction -- the memory side merges)
        const uint32_t prev_key = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x138, 0xf, 0xf, false);
        const uint32_t next_key = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x130, 0xf, 0xf, false);
        int flag = ((prev_key != key) || !commit) ? 1 : 0;     // first lane of its run
        double sum = (double)w, sum_sq = (double)w * (double)w;
        segmented_step_<0x111, 0xf>(sum, sum_sq, flag, squares);
        segmented_step_<0x112, 0xf>(sum, sum_sq, flag, squares);
        segmented_step_<0x114, 0xf>(sum, sum_sq, flag, squares);
        segmented_step_<0x118, 0xf>(sum, sum_sq, flag, squares);
        segmented_step_<0x142, 0xa>(sum, sum_sq, flag, squares);
        segmented_step_<0x143, 0xc>(sum, sum_sq, flag, squares);
        TAB_TIMED(const uint64_t t_before_add = __builtin_amdgcn_s_memtime();)
        if (commit && (next_key != key)) {
            unsafeAtomicAdd(bins + index, sum);
            if (squares) unsafeAtomicAdd(sq_bins + index, sum_sq);
        }
        TAB_TIMED(t_add += __builtin_amdgcn_s_memtime() - t_before_add;)
    }
    if (walk_alone) {
        // the noted segment (certainly inside the table), then this trip's
        if (held_n != 0u) {
            const row_t *rec = reinterpret_cast<const row_t *>(records + my_held_record * (uint32_t)kTabSegWords);
            const row_t r0 = rec[0], r1 = rec[1], r2 = rec[2];
            const Segment g = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, 0.0f};
            float d = held_d0;
            for (uint32_t taken = 0; taken < held_n; d += vstep, ++taken) {
                uint32_t index;
                bool ok_ = true;
                if (sample_bin<false, false, STD>(K, g, d, 0ull, 0u, index, ok_)) break;        // (never: a noted sample is inside the table)
                add_to_bin(bins, STD ? nullptr : sq_bins, index, r2.w * dm::exp_(-(r2.y + (d / r2.x) * r2.z)));
            }
        }
        if (active) {
            const Segment g = {ph.px, ph.py, ph.pz, ph.pt, ph.d.x, ph.d.y, ph.d.z, ph.inv_groupvel, 0.0f};
            float d = remainder;
            for (uint32_t taken = 0; (d < length) && (taken < n); d += vstep, ++taken) {
                uint32_t index;
                bool ok_ = true;
                if (sample_bin<false, false, STD>(K, g, d, 0ull, 0u, index, ok_)) { stop = true; break; }
                add_to_bin(bins, STD ? nullptr : sq_bins, index, impact * dm::exp_(-(depth + (d / length) * this_depth)));
            }
            d_end = d;
        }
    } else if (active && (my_first_oob != 0x7fffffff)) {
        stop = true;
        d_end = __builtin_bit_cast(float, pool_d[first + held_n + (uint32_t)my_first_oob]);
    }
    held_n = 0u;
    parity ^= 1u;
    // (the lists are this wave's own and a later trip writes them again: the reads above have to be done first)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (active) remainder = d_end - length;
    return stop;
}

// TAB: 0 = photon propagation, 1 = TABULATE, 2 = TABULATE + TABULATE_IMPACT_ANGLE (a kernel of its own, so that the
// four-dimensional table maker keeps its register allocation).  4 waves per SIMD: 86-110 VGPRs, nothing spilled, since
// the sampling constants are scalar loads from the parameter block.
// 3 = photon propagation without STOP_PHOTONS_ON_DETECTION (SetStopDetectedPhotons(false)): every DOM on a segment's way is
// saved and the photon travels on (find_collisions_keep); a translation unit of its own as well (prop_keep_kernel.hip)
// FAST: prop_device.hip.h (the standard configuration with every proof in hand; propagation only)
// (the pooled kernel reads these two from its parameters, KParams::k_aim / k_wait: constants here, the classic kernel has no scalar
// register to spare)
constexpr uint32_t kAimLanes = 8u, kParkedWait = 16u;

template <int MED, bool TILT, bool ANISO, bool FLASHER, int TAB, bool FAST = false>
__global__ void __launch_bounds__(kBlock, TAB ? 4 : kMinWavesPerSimd) prop_kernel(const KParams Pvalue)
{
    // the only kernel argument sits at offset 0 of the kernarg segment
    const KP P0 = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    (void)Pvalue;
    {   // stage the table image: one coalesced pass of the workgroup
        const uint32_t words = P0->table_words;
        const uint32_t *src = P0->tables;
        for (uint32_t i = threadIdx.x; i < words; i += kBlock) lds_words[i] = src[i];
    }
    uint32_t *stage = lds_words + P0->table_words + (threadIdx.x >> 6) * (kStageRecords * kStubWords);
    // TABULATE: no hits are staged; the first 12 words behind the image hold the reference particle instead
    constexpr bool TABULATE = (TAB == 1) || (TAB == 2);
    constexpr bool KEEP = (TAB == 3);
    if (TABULATE && threadIdx.x < 12u) lds_words[P0->table_words + threadIdx.x] = __builtin_bit_cast(uint32_t, P0->tab_ref[threadIdx.x]);
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    // Work units.  A step's photons share one RNG stream, so a step is sequential work (~5700 loop
    // iterations +-35 %).  Handing out whole steps leaves the last round of a bunch running in waves that are
    // two thirds empty (1M steps on 458k lanes: 17 % of the kernel time).  Steps are therefore cut into
    // `slices` slices of U photons that are handed out ROUND-ROBIN over the whole bunch -- slice 0 of every
    // step, then slice 1 of every step, ... -- so all steps advance together and finish within one slice of
    // each other.  Slice s of a step may start when slice s-1 has published the stream's state (it was
    // handed out n units earlier, so it practically always has); the photons of a step are still processed
    // in order from one RNG stream, whichever lanes do it.
    const uint32_t n_steps = P0->n_steps;
    // The queue head is one word that every wave increments: 1.2e8 requests per second is what one address sustains,
    // and 12 slices of 1M steps in 0.1 s are that many.  So there are kSubQueues heads on cache lines of their own;
    // sub-queue q hands out the steps i with i % kSubQueues == q -- slice 0 of each, then slice 1 of each ... -- so a
    // slice's predecessor is always an earlier unit of the same sub-queue.  A wave starts at the sub-queue of its
    // number and moves on when that one is used up; it is done when it has found them all used up in a row.
    uint32_t slice_photons, rounds;
    {
        const uint32_t max_photons = P0->queue[1];                 // scan_steps_kernel
        const uint32_t target = (uint32_t)P0->slices;
        slice_photons = (max_photons + target - 1u) / target;
        if (slice_photons == 0u) slice_photons = 1u;
        rounds = (max_photons + slice_photons - 1u) / slice_photons;
        if (rounds == 0u) rounds = 1u;
    }
    uint32_t sub_queue = (blockIdx.x * (uint32_t)kWavesPerBlock + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) % (uint32_t)kSubQueues;     // wave-uniform, and known to be
    uint32_t used_up = 0;                                                                                          // in a row
    uint32_t n_staged = 0;     // hit stubs waiting in the wave's staging area
    uint32_t parked_trips = 0; // trips since the first of the parked lanes parked
    uint32_t sidx = kNoStep;
    uint64_t rx = 0;
    uint32_t ra = 0;
    uint32_t photons_left = 0;
    uint32_t slice = 0;
    bool parked = false;       // has a step length and waits for the wave's next DOM search
    uint32_t search_kind = kSearchFull;     // of a parked lane: the full search, or kSearchNamed + the only DOM in reach
    uint32_t *pending = lds_words + P0->table_words + kWavesPerBlock * kStageRecords * kStubWords;     // per lane: that step length
    bool waiting = false;      // holds a unit whose previous slice has not been published yet
    bool last_slice = false;   // the unit ends its step
    bool alive = true;
    Vec3 step_dir = {0.0f, 0.0f, 1.0f};
    float unit_weight = 0.0f;   // TABULATE: the step's weight (c.cl:246-251), read when the lane takes the unit
    const bool tab_std = (TAB == 1) && (P0->tab_std != 0u);        // the table maker's standard configuration: the specialised sampler (sample_bin: STD)
    uint32_t tab_held_n = 0u, tab_parity = 0u;     // TABULATE, four axes: the lane's noted segment (samples, first sample) and the generation of this trip's
    float tab_held_d0 = 0.0f;                      // segment records (save_path_wave_carry)
    Photon ph;
    ph.abs_lens_left = 0.0f;    // "< epsilon" == this lane needs a photon
    ph.layer = 0;

    // The SIMD's arbiter issues from the oldest wave first, and the kernel is issue bound: left alone, the waves of a SIMD
    // advance at rates up to 8x apart (measured: trips per wave, p10/p90 = 4.6k/39.6k), so the slow ones stretch the
    // slice hand-offs and hold the last units of the bunch long after the queue is dry.  Each wave therefore takes
    // turns at the four issue priorities, offset by its wave slot.
    const uint32_t wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID.wave_id
    CENSUS(
    const unsigned long long t_start = wall_clock64();
    if (lane == 0 && !TAB) atomicMin(fresh_params(P0)->census + 8, t_start);
    unsigned long long c_trips = 0, t_dry = 0, c_run = 0, c_need = 0, c_wait = 0, c_parked = 0, c_dead = 0, c_phases = 0, c_created = 0;
    )
    // which lanes need a photon and which hold one, taken at the end of a trip for the next one (and for the loop's exit, a
    // plain backward branch)
    bool need_next = true;
    uint64_t m_need = ~0ull, m_ready = 0ull;
    // (analysis build of the table maker, tools/exp_tab_timers.py: shader-clock time per phase of a trip, summed per wave -- TAB_STAMP)
    TAB_TIMED(uint64_t t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();)
    for (uint32_t trip = 0;; ++trip) {
        if (!TAB && ((trip & ((1u << kPrioShift) - 1u)) == 0u)) switch (((trip >> kPrioShift) + wave_slot) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
        bool need = need_next;
        CENSUS(
        ++c_trips;
        if (t_dry == 0 && used_up > 0) t_dry = wall_clock64();
        c_need += __popcll(ballot(need && !waiting));
        c_wait += __popcll(ballot(need && waiting));
        c_parked += __popcll(ballot(parked));
        c_dead += __popcll(ballot(!alive));
        )

        // ---- new units / new photons, deferred until enough lanes wait for them ----
        // Photon creation is what is worth batching (k_new lanes), and taking new units goes with it (one atomic on the
        // queue head per wave and batch).  Handing a finished unit's stream on and looking for the predecessor's must
        // not wait for that: a wave with few running lanes would sit on finished units while their successors
        // elsewhere wait, which spreads (every waiting lane is one running lane less).  Finished units are published,
        // and predecessors polled for, at the latest every fourth trip.
        const uint64_t m_poll = ballot(need && waiting);
        const bool do_create = (m_ready == 0ull) || ((int)__popcll(m_need & ~m_poll) >= fresh_params(P0)->k_new);
        const bool finished = need && !waiting && (photons_left == 0) && (sidx != kNoStep);
        const uint64_t m_finished = ballot(finished);
        // (not the table maker: its waves have fp64 atomics in flight, which a poll would have to wait for first)
        if (do_create || (!TABULATE && ((m_finished | m_poll) != 0ull) && ((trip & 3u) == 0u))) {
            const KP P = fresh_params(P0);
            WorkRecord *work = P->work;
            if (m_finished != 0ull) {
                // publish the finished unit (c.cl:911-912).  The last slice of a step leaves the stream's state in the
                // converter's array for the next bunch; any other slice hands it to whoever takes the next slice:
                // state first, then the slice counter, both write-through (sc1) so that a lane on another XCD that
                // sees the counter sees the state
                if (finished) {
                    if (last_slice) P->rng_x[sidx] = rx;
                    else __hip_atomic_store(&work[sidx].x, rx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (finished) {
                    if (!last_slice) __hip_atomic_store(&work[sidx].done, slice + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sidx = kNoStep;
                }
            }
            const bool want_unit = do_create && need && (photons_left == 0) && !waiting;
            const uint64_t m_want = ballot(want_unit);
            if (m_want != 0ull) {
                // next units from the wave's sub-queue: one atomic per wave
                const uint32_t n_sub = (n_steps + (uint32_t)kSubQueues - 1u - sub_queue) / (uint32_t)kSubQueues;   // its steps
                const uint32_t total_sub = n_sub * rounds;
                const uint32_t count = (uint32_t)__popcll(m_want);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(P->queue + kQueueHeadStride * (sub_queue + 1u), count);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t this_queue = sub_queue;
                if (base + count > total_sub) {                      // (also when the head has run past the end)
                    sub_queue = (sub_queue + 1u == (uint32_t)kSubQueues) ? 0u : sub_queue + 1u;
                    ++used_up;
                } else {
                    used_up = 0;
                }
                if (want_unit) {
                    const uint32_t unit = base + (uint32_t)__popcll(m_want & lanes_below);
                    if ((base < total_sub) && (unit < total_sub)) {
                        const uint32_t s_new = unit / n_sub;
                        const uint32_t i_new = (unit - s_new * n_sub) * (uint32_t)kSubQueues + this_queue;
                        const uint32_t num = work[i_new].step.num_photons;
                        const uint32_t first = s_new * slice_photons;
                        if (first < num) {                  // otherwise this step is used up: ask again
                            sidx = i_new;
                            slice = s_new;
                            last_slice = (num - first <= slice_photons);
                            photons_left = last_slice ? (num - first) : slice_photons;
                            waiting = true;
                        }
                    } else if (used_up >= (uint32_t)kSubQueues) {
                        alive = false;                              // every sub-queue was found used up: no work is left
                        need = false;
                    }
                }
            }
            if (need && waiting) {
                WorkRecord *rec = P->work + sidx;
                const uint32_t published = (slice == 0u) ? 0u : __hip_atomic_load(&rec->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                DEBUG_COUNTED(if (published < slice) atomicAdd(P->queue + 2, 1u);)
                if (published >= slice) {
                    // c.cl:458-461; slice 0 reads the state left by the previous bunch
                    rx = (slice == 0u) ? rec->x : __hip_atomic_load(&rec->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ra = rec->a;
                    step_dir = work_direction(&rec->step);
                    if (TABULATE) unit_weight = P->steps[sidx].weight;      // (the work record carries the direction there)
                    waiting = false;
                }
            }
            CENSUS(
            if (do_create) ++c_phases;
            c_created += __popcll(ballot(do_create && need && !waiting && (photons_left > 0)));
            )
            if (do_create && need && !waiting && (photons_left > 0)) {
                create_photon<MED, TILT, FLASHER, TABULATE, FAST>(P, &P->work[sidx].step, step_dir, rx, ra, ph);
                need = false;
            }
            // nothing runnable in this wave: every lane waits for another wave's slice
            if ((m_ready == 0ull) && (ballot(alive && !need) == 0ull)) {
                DEBUG_COUNTED(if (lane == 0) atomicAdd(P->queue + 3, 1u);)
                __builtin_amdgcn_s_sleep(16);
            }
        }

        TAB_STAMP(0)        // units and creation
        // ---- one reference loop iteration for the lanes that hold a photon ----
        // A lane runs the layer walk; if its step could reach a string (2 % of the lanes) it parks with the step
        // length until `k_search` lanes of the wave are parked (or nothing else can advance), and the DOM search runs
        // for all of them at once: the search costs the wave the same whether 1 or 12 lanes need it.
        const bool run = alive && !need && !parked;
        CENSUS(c_run += __popcll(ballot(run));)
        float distance = 0.0f;
        bool hit = false;
        uint32_t hit_string = 0, hit_dom = 0;
        if (run) {
            const uint32_t near_string = TABULATE ? 0u : free_flight_bound(fresh_params(P0), ph.px, ph.py);
            distance = propagate_through_layers<MED, TILT, ANISO, FAST>(fresh_params(P0), ph, rx, ra);
            // the search cannot find a DOM closer than the nearest string cylinder: skipped when the step ends before
            // ... second: a step that can reach no other string touches this one only if it is aimed at it (not asked of photons
            // born at a DOM: they live inside the string's cylinder)
            // (asked when few lanes of the wave are at a string, prop_pool_kernel.hip)
            bool at_string = !TABULATE && !(distance < free_flight_of(near_string));
            if (!TABULATE && !FLASHER && (uint32_t)__popcll(ballot(at_string)) <= kAimLanes)
                at_string = at_string && !segment_misses_string(fresh_params(P0), ph, distance, near_string);
            if (at_string) {
                const uint32_t kind = dom_search_needed<FLASHER>(fresh_params(P0), ph, distance);
                if (kind != kSearchNone) {
                    parked = true;
                    search_kind = kind;
                    pending[threadIdx.x] = __builtin_bit_cast(uint32_t, distance);
                }
            }
        }
        bool advance = run && !parked;
        if (!TABULATE) {
            const uint64_t m_parked = ballot(parked);
            // (a parked lane waits for company at most kParkedWait trips -- in the instantiations without STOP_PHOTONS_ON_DETECTION,
            // which take bunches of every size; the others run the bunches the pooled kernel leaves them, where k_search is 1)
            if (KEEP) parked_trips = (m_parked != 0ull) ? parked_trips + 1u : 0u;
            if ((m_parked != 0ull) && (((int)__popcll(m_parked) >= fresh_params(P0)->k_search) || (ballot(advance) == 0ull) ||
                                       (KEEP && (parked_trips > kParkedWait)))) {
                if (KEEP) parked_trips = 0u;
                if (KEEP && parked) {
                    // without STOP_PHOTONS_ON_DETECTION (c.cl:704-750): the search saves what it finds, nothing is shortened or absorbed
                    const KP P = fresh_params(P0);
                    distance = __builtin_bit_cast(float, pending[threadIdx.x]);
                    KeepSink K;
                    K.step_index = sidx;
                    K.history_n = (uint32_t)P->history_n;
                    K.ring = reinterpret_cast<const float4 *>(P->hist_ring) + (size_t)(blockIdx.x * kBlock + threadIdx.x) * K.history_n;
                    K.string_mask = pending + kBlock + threadIdx.x;
                    K.mask_stride = (uint32_t)kBlock;
                    K.mask_words = ((uint32_t)P->num_strings + 63u) >> 6;
                    find_collisions_keep(P, ph, distance, K);
                    parked = false;
                    advance = true;
                }
                if (!KEEP && parked) {
                    distance = __builtin_bit_cast(float, pending[threadIdx.x]);
                    // (as in prop_pool_kernel.hip: the confined search in the flasher instantiations, for all parked lanes or none)
                    bool full = FLASHER ? (ballot(search_kind == kSearchFull) != 0ull) : true;
                    if (!full) {
                        // only one DOM is in reach: what the reference's search does for that DOM, and nothing else
                        const uint32_t id = search_kind - kSearchNamed;
                        const uint4 named = fresh_params(P0)->dom_named[id];
                        if (named.x != 0xffffffffu) hit = find_collision_named<FAST>(fresh_params(P0), ph, distance, id, named, hit_string, hit_dom);
                        else full = true;
                    }
                    if (full) hit = find_collision<FAST>(fresh_params(P0), ph, distance, hit_string, hit_dom);
                    parked = false;
                    advance = true;
                }
            }
        }
        TAB_STAMP(1)        // layer walk
        if (TABULATE) {
            // c.cl:755-785; the absorption budget is the fixed PROPAGATE_FOR_FIXED_NUMBER_OF_ABSORPTION_LENGTHS
            const KP P = fresh_params(P0);
            const float travelled = P->fixed_abs - ph.abs_lens_left;
            const float weight = run ? unit_weight : 0.0f;
            uint32_t *wave_lds = lds_words + ((P->table_words + 16u + 3u) & ~3u) + (threadIdx.x >> 6) * (uint32_t)kTabWaveWords;     // (16-byte rows)
            bool left_table;
            if (TAB == 2) left_table = save_path_wave<true>(P, wave_lds, run, ph, weight, distance,
                                                            ph.tab_remainder, ph.tab_depth, travelled - ph.tab_depth, rx, ra
                                                            TAB_TIMED(, t_acc[6], t_last)
                                                            );
            else if (tab_std) left_table = save_path_wave_carry<true>(P, wave_lds, run, ph, weight, distance, ph.tab_remainder, ph.tab_depth, travelled - ph.tab_depth,
                                                                      tab_held_n, tab_held_d0, tab_parity, false
                                                                      TAB_TIMED(, t_acc[6], t_last, t_acc[7])
                                                                      );
            else left_table = save_path_wave_carry<false>(P, wave_lds, run, ph, weight, distance, ph.tab_remainder, ph.tab_depth, travelled - ph.tab_depth,
                                                          tab_held_n, tab_held_d0, tab_parity, false
                                                          TAB_TIMED(, t_acc[6], t_last, t_acc[7])
                                                          );
            if (run) {
                if (left_table) ph.abs_lens_left = 0.0f;
                ph.tab_depth = P->fixed_abs - ph.abs_lens_left;
            }
        }
        TAB_STAMP(2)        // savePath
        // ---- hit write-out (c.cl:329-385, collision c.cl:557-578) ----
        // The stubs collect in the wave's staging area across trips and leave kStageRecords at a time (and at the end of the
        // kernel): one atomic on the chip-wide hit counter per eight hits (prop_pool_kernel.hip).  A photon history is
        // copied next to its hit and needs the hit's final index at once: with histories every chunk leaves right away.
        const uint64_t hit_mask = ballot(hit);
        if (hit_mask != 0ull) {
            const uint32_t total = (uint32_t)__popcll(hit_mask);
            const uint32_t rank = (uint32_t)__popcll(hit_mask & lanes_below);
            const uint32_t hn = (uint32_t)fresh_params(P0)->history_n;
            for (uint32_t done = 0; done < total;) {
                const uint32_t space = (uint32_t)kStageRecords - n_staged;
                const uint32_t take = (total - done < space) ? (total - done) : space;
                const bool mine = hit && rank >= done && rank < done + take;
                const uint32_t slot = n_staged + rank - done;
                if (mine) {
                    uint32_t *st = stage + slot * kStubWords;
                    st[0] = dm::f2u(ph.px); st[1] = dm::f2u(ph.py); st[2] = dm::f2u(ph.pz); st[3] = dm::f2u(ph.pt);
                    st[4] = dm::f2u(ph.d.x); st[5] = dm::f2u(ph.d.y); st[6] = dm::f2u(ph.d.z); st[7] = dm::f2u(distance);
                    st[8] = dm::f2u(ph.total_path); st[9] = dm::f2u(ph.abs_lens_left); st[10] = dm::f2u(ph.inv_groupvel);
                    st[11] = ph.num_scatters; st[12] = sidx;
                    st[13] = (uint32_t)ph.rx_start; st[14] = (uint32_t)(ph.rx_start >> 32);
                    st[15] = (hit_string & 0xffffu) | (hit_dom << 16);
                }
                n_staged += take;
                done += take;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if ((n_staged == (uint32_t)kStageRecords) || (hn != 0u)) {
                    const KP P = fresh_params(P0);
                    const uint32_t base = flush_hit_stubs(P, stage, n_staged, lane);
                    n_staged = 0u;
                    if ((hn != 0u) && mine && (base + slot < P->max_hits)) {    // c.cl:387-392 (the staging area was empty before this chunk)
                        const float4 *ring = reinterpret_cast<const float4 *>(P->hist_ring) + (size_t)(blockIdx.x * kBlock + threadIdx.x) * hn;
                        float4 *dst = reinterpret_cast<float4 *>(P->hist_out) + (size_t)(base + slot) * hn;
                        for (uint32_t k = 0; k < hn; ++k) dst[k] = ring[k];
                    }
                }
            }
        }
        TAB_STAMP(8)        // (analysis build: up to the advance)
        if (advance) {
            if (hit) ph.abs_lens_left = 0.0f;                                   // c.cl:741-744
            ph.px += ph.d.x * distance;
            ph.py += ph.d.y * distance;
            ph.pz += ph.d.z * distance;
            ph.pt += ph.inv_groupvel * distance;
            ph.total_path += distance;
            if (ph.abs_lens_left < kEpsilon) {
                --photons_left;                                                 // absorbed or detected
            } else {
                const KP P = fresh_params(P0);
                const uint32_t hn = (uint32_t)P->history_n;
                if (hn != 0u) {                                                 // c.cl:833-837
                    float4 *ring = reinterpret_cast<float4 *>(P->hist_ring) + (size_t)(blockIdx.x * kBlock + threadIdx.x) * hn;
                    ring[ph.num_scatters % hn] = make_float4(ph.px, ph.py, ph.pz, ph.abs_lens_left);
                }
                if (ANISO && P->has_pre) apply_matrix(P->pre, P->pre_renorm, ph.d, FAST || (P->div_ok & kFastMatrices) != 0u);
                TAB_STAMP(9)        // (position update, history)
                const float cos_s = scattering_cos<FAST>(P, rx, ra);
                TAB_STAMP(10)       // (scattering angle)
                const float sin_s = dm::sqrt_near_(1.0f - sqr(cos_s));       // |cos_s| <= 1: 0 or >= 2^-24
                scatter_direction(cos_s, sin_s, ph.d, rng_co(rx, ra));
                if (ANISO && P->has_post) apply_matrix(P->post, P->post_renorm, ph.d, FAST || (P->div_ok & kFastMatrices) != 0u);
                ++ph.num_scatters;
            }
        }
        need_next = alive && !parked && (ph.abs_lens_left < kEpsilon);
        m_need = ballot(need_next);
        m_ready = ballot(alive && !need_next);
        TAB_STAMP(3)        // advance, scattering
        TAB_TIMED(
        t_acc[4] += 1;      // trips
        t_acc[5] += (uint64_t)__popcll(ballot(run));
        )
        if ((m_need | m_ready) == 0ull) break;
    }
    if ((TAB == 1) && (ballot(tab_held_n != 0u) != 0ull)) {
        // the samples the last trip noted
        const KP P = fresh_params(P0);
        uint32_t *wave_lds = lds_words + ((P->table_words + 16u + 3u) & ~3u) + (threadIdx.x >> 6) * (uint32_t)kTabWaveWords;
        float no_remainder = 0.0f;
        (void)save_path_wave_carry<false>(P, wave_lds, false, ph, 0.0f, 0.0f, no_remainder, 0.0f, 0.0f, tab_held_n, tab_held_d0, tab_parity, true      // (once per wave: the generic sampler)
                                   TAB_TIMED(, t_acc[6], t_last, t_acc[7])
                                   );
    }
    TAB_TIMED(
    if (TABULATE && lane == 0) {
        // (the table's first words take the sums: its contents are meaningless in this build)
        double *out = fresh_params(P0)->tab_bins;
        for (int k = 0; k < 12; ++k) unsafeAtomicAdd(out + k, (double)t_acc[k]);
    }
    )
    if (n_staged != 0u) flush_hit_stubs(fresh_params(P0), stage, n_staged, lane);
    CENSUS(
    if (lane == 0 && !TAB) {
        unsigned long long *d = fresh_params(P0)->census;
        const uint32_t w = blockIdx.x * (uint32_t)kWavesPerBlock + (threadIdx.x >> 6);
        d[16 + 3 * w] = wall_clock64();
        d[16 + 3 * w + 1] = t_dry;
        d[16 + 3 * w + 2] = c_trips;
        atomicAdd(d + 0, c_trips); atomicAdd(d + 1, c_run); atomicAdd(d + 2, c_need); atomicAdd(d + 3, c_wait);
        atomicAdd(d + 4, c_parked); atomicAdd(d + 5, c_dead); atomicAdd(d + 6, c_phases); atomicAdd(d + 7, c_created);
    }
    )
}

#ifndef CLSIMHIP_TAB_UNIT      // (prop_tab_kernel.hip compiles this file for the TABULATE instantiations only)
// meta[1] = largest numPhotons of the bunch (sizes the slices of the unit queue)
__global__ void __launch_bounds__(256) scan_steps_kernel(const DevStep *steps, uint32_t n, uint32_t *meta, WorkRecord *work,
                                                         const uint64_t *rng_x, const uint32_t *rng_a, uint32_t num_generators)
{
    // one pass over the bunch: largest numPhotons (-> slice size) and the work records
    uint32_t m = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        WorkRecord r;
        r.step = steps[i];
        r.x = rng_x[i];
        r.a = rng_a[i];
        r.done = 0u;
        {   // A step with a non-finite field makes the reference's photon loop spin forever (a NaN absorption budget
            // never drops below EPSILON); here that would hang the GPU.  Such a step propagates no photons and is
            // counted in meta[2]; its RNG stream is left alone.
            const float f[8] = {r.step.x, r.step.y, r.step.z, r.step.t, r.step.theta, r.step.phi, r.step.length, r.step.beta};
            bool finite = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) finite = finite && (__builtin_fabsf(f[k]) <= 3.0e38f);
            // likewise a source type without a wavelength generator: generateWavelength() returns 0 for it
            // (MediumPropertiesSource.cxx:392-432) and every length of that photon becomes 0/0
            const bool no_spectrum = (num_generators > 1u) && ((r.step.source_type_and_pad & 0xffu) >= num_generators);
            if ((!finite || no_spectrum) && r.step.num_photons != 0u) { r.step.num_photons = 0u; atomicAdd(meta + 2, 1u); }
        }
        {   // The work record's step is the view photon creation needs: the direction of the step (c.cl:482-489), two
            // sincos per step here instead of per photon, takes the place of theta, phi and of the weight, which only a hit
            // record needs -- and that reads the caller's step array (make_hit_record, save_path_wave)
            const Vec3 d = step_direction(&r.step);
            r.step.theta = d.x; r.step.phi = d.y; r.step.weight = d.z;
        }
        work[i] = r;
        const uint32_t v = r.step.num_photons;
        m = v > m ? v : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63u) == 0u && m != 0u) atomicMax(meta + 1, m);
}

// Expands the hit stubs of one launch into I3CLSimPhoton records, in place (slot i -> record i).
template <bool FLASHER>
__global__ void __launch_bounds__(256) assemble_hits_kernel(const KParams Pvalue)
{
    const KP P = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    (void)Pvalue;
    {
        const uint32_t words = P->table_words;
        const uint32_t *src = P->tables;
        for (uint32_t i = threadIdx.x; i < words; i += 256) lds_words[i] = src[i];
    }
    __syncthreads();
    const uint32_t counted = *P->hit_count;
    const uint32_t n = counted < P->max_hits ? counted : P->max_hits;
    uint32_t *out_words = reinterpret_cast<uint32_t *>(P->out);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint32_t *slot = out_words + (size_t)i * 20u;
        HitStub h;
        uint32_t *hw = reinterpret_cast<uint32_t *>(&h);
#pragma unroll
        for (int w = 0; w < kStubWords; ++w) hw[w] = slot[w];
        // A stub whose indices name no step / string / DOM (a corrupted record) is not expanded -- expanding it would read the step, the
        // stream multiplier and the DOM tables out of range -- it stays as it is and is counted in queue[4]: the host path then fails
        // the bunch like its own conversion would (converter.cpp: finish / replace_indices).  id_dom_start has one entry more than
        // there are strings.
        const uint32_t s_index = h.string_and_dom & 0xffffu, d_index = h.string_and_dom >> 16;
        bool named = (h.step_index < P->n_steps) && (s_index < (uint32_t)P->num_strings);
        if (named && P->id_strings) named = d_index < P->id_dom_start[s_index + 1u] - P->id_dom_start[s_index];
        if (!named) {
            atomicAdd(P->queue + 4, 1u);
            continue;
        }
        uint32_t rec[20];
        const float abs_lens_initial = make_hit_record<FLASHER>(P, h, rec);
        if (P->id_strings)                   // index -> ID (OpenCL.cxx:1565-1600), same for every record: wave-uniform branch
            rec[11] = (uint32_t)(uint16_t)P->id_strings[s_index] | ((uint32_t)P->id_doms[P->id_dom_start[s_index] + d_index] << 16);
#pragma unroll
        for (int w = 0; w < 20; ++w) slot[w] = rec[w];
        // c.cl:836: the ring holds the absorption lengths LEFT at each scatter; the reference stores initial - left
        const uint32_t hn = (uint32_t)P->history_n;
        for (uint32_t k = 0; k < hn; ++k) {
            float *w = P->hist_out + ((size_t)i * hn + k) * 4u + 3u;
            *w = abs_lens_initial - *w;
        }
    }
}

// ---- math probe used by tests/test_detmath_gpu.py ----
__global__ void eval_math_kernel(int what, const float *xs, const float *ys, uint32_t n, float *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xs[i], y = ys ? ys[i] : 0.0f;
    float r = 0.0f, s, c;
    switch (what) {
    case 0: r = dm::log_(x); break;
    case 1: r = dm::exp_(x); break;
    case 2: dm::sincos_(x, s, c); r = s; break;
    case 3: dm::sincos_(x, s, c); r = c; break;
    case 4: r = dm::powr_(x, y); break;
    case 5: r = dm::acos_(x); break;
    case 6: r = dm::atan2_(x, y); break;
    case 7: r = dm::rsqrt_(x); break;
    case 8: r = dm::sqrt_(x); break;
    case 9: r = x / y; break;
    case 10: r = dm::acos_f(x); break;
    case 11: r = dm::rcp_(x); break;
    case 12: r = dm::sqrt_near_(x); break;
    case 13: r = dm::rsqrt_near_(x); break;
    case 14: r = dm::powr_unit_(x, y); break;
    case 15: r = dm::cbrt_(x); break;
    case 16: r = dm::div_near_(x, y); break;
    default: break;
    }
    out[i] = r;
}

// Exhaustive proof runs for the range-restricted operations of detmath.hip.h: every significand (2^23) x every binary
// exponent in [exp_lo, exp_hi], both signs for the reciprocal, against the IEEE operation.  what: 11 rcp_, 12 sqrt_near_,
// 13 rsqrt_near_; 16 div_near_ (two-argument: see the kernel); 17 rcp_of_rcp_(rcp_(x), x) against 1/(1/x); 18 rsqrt_unit_ on its window;
// 19 the table maker's axis_bin_ on every bit pattern (exponents do not apply).
// result[0] = mismatches, result[1..] = bit patterns of the first few mismatching arguments.
__global__ void check_math_kernel(int what, int exp_lo, int exp_hi, uint32_t *result, uint32_t result_cap)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;        // significand bits
    if (m >= (1u << 23)) return;
    if (what == 19) {
        // axis_bin_ against axis_bin_generic_: every one of the 2^32 bit patterns (512 per thread) x five bin counts
        for (uint32_t k = 0; k < 512u; ++k) {
            const float t = dm::u2f((m << 9) | k);
            const int counts[5] = {1, 36, 105, 200, 65534};
            for (int c = 0; c < 5; ++c) {
                const int nb = __builtin_amdgcn_readfirstlane(counts[c]);
                if (axis_bin_(t, nb) != axis_bin_generic_(t, nb)) {
                    const uint32_t k2 = atomicAdd(result, 1u);
                    if (k2 + 1u < result_cap) result[k2 + 1u] = dm::f2u(t);
                }
            }
        }
        return;
    }
    if (what == 16) {
        // div_near_: every divisor significand x the divisor exponents [exp_lo, exp_hi] x both divisor signs x 40 numerators:
        // 32 pseudo-random ones over the whole admissible range and both signs, and 8 built from the divisor (exact and
        // nearly exact quotients, all-ones and power-of-two significands), where a wrong last bit would show first
        for (int e = exp_lo; e <= exp_hi; ++e) {
            const uint32_t bbits = ((uint32_t)(e + 127) << 23) | m;
            for (int j = 0; j < 40; ++j) {
                uint32_t h = (m * 2654435761u) ^ ((uint32_t)(e + 1000) * 40503u) ^ ((uint32_t)j * 2246822519u);
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
                uint32_t abits;
                if (j < 32) {
                    const uint32_t ae = 127u - 40u + (h >> 23) % 101u;                     // exponents -40 ... 60
                    abits = (h & 0x807fffffu) | (ae << 23);
                } else {
                    const uint32_t ae = 127u - 40u + (h >> 9) % 101u;
                    const uint32_t k = (h & 7u) + 1u;
                    const float bb = dm::u2f((127u << 23) | m);                            // the divisor's significand in [1, 2)
                    float t;
                    switch (j - 32) {
                        case 0: t = bb * (float)k; break;                                   // exact small quotients (when the product is exact)
                        case 1: t = dm::u2f(dm::f2u(bb * (float)k) + 1u); break;            // ... and their neighbours
                        case 2: t = dm::u2f(dm::f2u(bb * (float)k) - 1u); break;
                        case 3: t = dm::u2f((127u << 23) | 0x7fffffu); break;               // all ones
                        case 4: t = 1.0f; break;                                           // a power of two: the quotient is RN(1/b) scaled
                        case 5: t = dm::u2f((127u << 23) | (0x7fffffu & ~m)); break;        // complement of the divisor's bits
                        case 6: t = bb * bb; break;                                        // quotient close to the divisor itself
                        default: t = dm::u2f((127u << 23) | ((m + k) & 0x7fffffu)); break;  // quotient within a few ulp of one
                    }
                    abits = (dm::f2u(t) & 0x007fffffu) | (ae << 23) | (h & 0x80000000u);
                }
                for (int sign = 0; sign < 2; ++sign) {
                    const float b = dm::u2f(bbits | ((uint32_t)sign << 31)), a = dm::u2f(abits);
                    const float want = a / b, got = dm::div_near_(a, b);
                    if (dm::f2u(want) != dm::f2u(got)) {
                        const uint32_t k2 = atomicAdd(result, 1u);
                        if (k2 + 2u < result_cap && (k2 & 1u) == 0u) { result[k2 + 1u] = dm::f2u(a); result[k2 + 2u] = dm::f2u(b); }
                    }
                }
            }
        }
        return;
    }
    for (int e = exp_lo; e <= exp_hi; ++e) {
        const uint32_t bits = ((uint32_t)(e + 127) << 23) | m;
        for (int sign = 0; sign < ((what == 11 || what == 17) ? 2 : 1); ++sign) {
            const float x = dm::u2f(bits | ((uint32_t)sign << 31));
            float want, got;
            if (what == 18) {       // rsqrt_unit_: the 2047 patterns of its window (exponents and the rest of the significands do not apply)
                if (e != exp_lo || sign != 0 || m > 2046u) continue;
                const float xx = dm::u2f(0x3f800000u - 1023u + m);
                want = 1.0f / __builtin_sqrtf(xx); got = dm::rsqrt_unit_(xx);
                if (!dm::rsqrt_unit_ok_(xx) || dm::rsqrt_unit_ok_(dm::u2f(0x3f800000u + 1024u)) || dm::rsqrt_unit_ok_(dm::u2f(0x3f800000u - 1024u))) got = 0.0f;
                if (dm::f2u(want) != dm::f2u(got)) { const uint32_t k = atomicAdd(result, 1u); if (k + 1u < result_cap) result[k + 1u] = dm::f2u(xx); }
                continue;
            }
            if (what == 17) { const float b = 1.0f / x; want = 1.0f / b; got = dm::rcp_of_rcp_(dm::rcp_(x), x); }       // (rcp_(x) == b: what = 11)
            else if (what == 11) { want = 1.0f / x; got = dm::rcp_(x); }
            else if (what == 12) { want = __builtin_sqrtf(x); got = dm::sqrt_near_(x); }
            else { want = 1.0f / __builtin_sqrtf(x); got = dm::rsqrt_near_(x); }
            if (dm::f2u(want) != dm::f2u(got)) {
                const uint32_t k = atomicAdd(result, 1u);
                if (k + 1u < result_cap) result[k + 1u] = dm::f2u(x);
            }
        }
    }
}

// ---- host-side launchers (called from converter.cpp) ----
hipError_t launch_scan_steps(const KParams &P, hipStream_t stream)
{
    const uint32_t sgrid = (P.n_steps + 255u) / 256u;
    hipLaunchKernelGGL(scan_steps_kernel, dim3(sgrid < 1024u ? sgrid : 1024u), dim3(256), 0, stream, P.steps, P.n_steps, P.queue,
                       P.work, P.rng_x, P.rng_a, (uint32_t)P.num_gen);
    return hipGetLastError();
}

// second pass (same stream): stubs -> I3CLSimPhoton records.  Hits are ~1e-3 of the photons.
hipError_t launch_assemble_hits(const KParams &P, bool flasher, int device, hipStream_t stream)
{
    const size_t image_bytes = (size_t)P.table_words * 4;
    if (image_bytes > 64 * 1024) {
        // once per (device, kernel, image size)
        static std::mutex m;
        static std::map<std::pair<int, size_t>, int> ready;
        std::lock_guard<std::mutex> lk(m);
        int &done = ready[std::make_pair(2 * device + (flasher ? 1 : 0), image_bytes)];
        if (!done) {
            const hipError_t err = flasher ? hipFuncSetAttribute(reinterpret_cast<const void *>(&assemble_hits_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)image_bytes)
                                           : hipFuncSetAttribute(reinterpret_cast<const void *>(&assemble_hits_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)image_bytes);
            if (err != hipSuccess) return err;
            done = 1;
        }
    }
    if (flasher) hipLaunchKernelGGL((assemble_hits_kernel<true>), dim3(512), dim3(256), image_bytes, stream, P);
    else hipLaunchKernelGGL((assemble_hits_kernel<false>), dim3(512), dim3(256), image_bytes, stream, P);
    return hipGetLastError();
}

#else
hipError_t launch_scan_steps(const KParams &P, hipStream_t stream);
hipError_t launch_assemble_hits(const KParams &P, bool flasher, int device, hipStream_t stream);
#endif

template <int MED, bool TILT, bool ANISO, bool FLASHER, int TAB, bool FAST = false>
static hipError_t launch_variant(const KParams &Pin, hipStream_t stream, int grid_wanted = 0)
{
    KParams P = Pin;
    constexpr bool TABULATE = (TAB == 1) || (TAB == 2);
    // (without STOP_PHOTONS_ON_DETECTION: one more word per lane and 64 strings, find_collisions_keep's string mask)
    const size_t lds_bytes = TABULATE ? (size_t)(((P.table_words + 16 + 3) & ~3u) + kWavesPerBlock * kTabWaveWords) * 4
                                      : (size_t)(P.table_words + kWavesPerBlock * kStageRecords * kStubWords + kBlock
                                                 + ((TAB == 3) ? kBlock * (((size_t)P.num_strings + 63u) >> 6) : 0u)) * 4;
    if (lds_bytes > 160u * 1024u) return hipErrorInvalidValue;
    // persistent grid: as many workgroups as the chip holds at once (queue-fed), never more than the work.
    // Occupancy, CU count and the function attributes are per (device, variant): one process may drive converters on
    // several GPUs (the reference's usual model, I3CLSimServer.cxx:77-137) and from several threads.
    struct Plan { int cus = 0, resident = 0; };
    static std::mutex plan_mutex;
    static std::map<std::pair<int, size_t>, Plan> plans;     // (device, LDS bytes of the workgroup: the image differs per configuration)
    int dev = 0;
    {
        const hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
    }
    Plan plan;
    {
        std::lock_guard<std::mutex> lk(plan_mutex);
        Plan &pl = plans[std::make_pair(dev, lds_bytes)];
        if (pl.resident == 0) {
            int cus = 0, per_cu = 0;
            hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (e == hipSuccess && lds_bytes > 64 * 1024)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(&prop_kernel<MED, TILT, ANISO, FLASHER, TAB, FAST>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e == hipSuccess)
                e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, prop_kernel<MED, TILT, ANISO, FLASHER, TAB, FAST>, kBlock, lds_bytes);
            if (e != hipSuccess) return e;
            if (per_cu < 1) per_cu = 1;
            if (cus < 1) cus = 1;
            pl.cus = cus;
            pl.resident = cus * per_cu;
        }
        plan = pl;
    }
    const int resident = plan.resident;
    // Grid and slices per step, from a scan on MI355X (200-photon steps, SPICE-Mie, n = 0.13M ... 4M, 5/6/7 workgroups per
    // CU x 8/12/16/24 slices x 3/5 parked lanes per DOM search; r = steps per lane):
    //   * more resident waves hide more latency (4M steps: 7 per CU 2.20e9 photons/s, 6: 2.14e9, 5: 2.03e9), but every
    //     resident lane is one more consumer of the same n steps and a bunch ends with every lane finishing what it
    //     holds: the largest grid (7, 6, 5 workgroups per CU) that leaves r >= 2 (0.5M steps: 5 per CU 1.81e9, 7: 1.53e9;
    //     1M steps: 7 per CU 2.06e9, 5: 1.99e9);
    //   * 16 slices (12 ... 24 are within 0.5 % of each other everywhere; 8 loses 1-2 %); whole steps for r < 1.
    // clsimhip_set_tuning("grid" / "slices") overrides.
    const uint32_t needed = (P.n_steps + kBlock - 1) / kBlock;
    uint32_t grid = (uint32_t)resident;
    {
        // concurrent launches of one converter (clsimhip_set_concurrent_device_launches): each takes its share of the CUs
        const int share = (P.chip_share > 1) ? P.chip_share : 1;
        const int cus = (plan.cus / share > 0) ? plan.cus / share : 1;
        const int per_cu = resident / plan.cus;
        // TABULATE (round 4, 200-photon steps at the origin, 4 axes): a lane that has a second unit to take balances the end of the launch;
        // 262 144 steps on 3 workgroups per CU (r = 1.33) 2.306e7 photons/s, 3.5: 2.29, 4 (r = 1): 2.236, 2.5: 2.07; 524 288 steps on 5, 4, 3
        // per CU: 2.34 / 2.25 / 2.35 (profiles/r04/tab_grid_scan.txt); with the impact-angle axis 3 and 4 per CU are level
        const int floor_per_cu = TABULATE ? (per_cu < 3 ? per_cu : 3) : (per_cu < 5 ? per_cu : 5);
        const double steps_per_lane_wanted = TABULATE ? 1.3 : 2.0;
        int chosen = floor_per_cu;
        for (int k = per_cu; k >= floor_per_cu; --k)
            if ((double)P.n_steps / ((double)cus * k * kBlock) >= steps_per_lane_wanted) { chosen = k; break; }
        grid = (uint32_t)(cus * chosen);
    }
    if (grid_wanted >= 1 && grid_wanted <= resident) grid = (uint32_t)grid_wanted;          // clsimhip_set_tuning("grid")
    if (needed < grid) grid = needed;
    {
        const double r = (double)P.n_steps / ((double)grid * kBlock);
        if (P.slices <= 0) P.slices = (r < 1.0) ? 1 : 16;
        // lanes without a photon before a wave creates (round 4, profiles/r04/scan_classic_k_new.txt: cascade steps, 262 144 / 393 216 per bunch,
        // 12: 1.978 / 2.591e9 photons/s, 16: 1.985 / 2.607, 20: 1.973 / 2.611, 8: 1.92 / 2.48; flasher steps, 312 320 / 458 752: 8: 1.630 / 1.969,
        // 10: 1.629 / 1.965, 12: 1.616 / 1.954, 16: 1.57 / -)
        if (P.k_new <= 0) P.k_new = TABULATE ? 12 : (FLASHER ? 8 : 16);
        // lanes parked before a wave searches for DOMs: pays when lanes have plenty of steps (1.5M steps: 3 -> 5 is
        // +1.6 %), costs when they are scarce (0.8M steps: -1.7 %)
        // (flasher instantiations: searches are rare since the filter knows about photons inside their DOM of birth, prop_device.hip.h:
        // 312 500 flasher steps, 1 parked lane 1.53e9 photons/s, 2: 1.49, 3: 1.46, 5: 1.40)
        // Since the filter asks whether a photon is aimed at the string it passes, searches are rare (0.009 per trip on cascade steps)
        // and a lane that waits for company waits long: this kernel, which has no scalar register left for the pooled kernel's
        // waiting limit, searches for the first parked lane (0.5M cascade steps: threshold 3 2.29e9 photons/s, 1: below; the
        // instantiations without STOP_PHOTONS_ON_DETECTION have the limit and keep the thresholds)
        if (P.k_search <= 0) P.k_search = (TAB != 3 || FLASHER || r < 1.5) ? 1 : (r < 2.2) ? 3 : 5;
        if ((uint64_t)P.n_steps * (uint64_t)P.slices >= 0x7fffffffull) P.slices = 1;    // 32-bit unit counters
    }
    hipError_t err = launch_scan_steps(P, stream);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL((prop_kernel<MED, TILT, ANISO, FLASHER, TAB, FAST>), dim3(grid), dim3(kBlock), lds_bytes, stream, P);
    err = hipGetLastError();
    if (err != hipSuccess || TABULATE) return err;
    return launch_assemble_hits(P, FLASHER, dev, stream);
}

#ifndef CLSIMHIP_TAB_UNIT
hipError_t launch_prop_kernel(const KParams &P, const KVariant &v, hipStream_t stream)
{
    if (P.n_steps == 0) return hipSuccess;
    if (v.lengths < CLSIMHIP_LENGTHS_CONSTANT || v.lengths > CLSIMHIP_LENGTHS_TABLE) return hipErrorInvalidValue;
    if (v.lengths == CLSIMHIP_LENGTHS_TABLE && (!P.len_table || P.len_tab_n < 2)) return hipErrorInvalidValue;
    const int key = 8 * v.lengths + (v.tilt ? 4 : 0) + (v.aniso ? 2 : 0) + (v.flasher ? 1 : 0);
    const bool fast = v.fast && P.history_n == 0 && !v.generic_only;
    switch (key) {
#define CASE(k, a, b, c, d) case k: return fast ? launch_variant<a, b, c, d, 0, true>(P, stream, v.grid) : launch_variant<a, b, c, d, 0, false>(P, stream, v.grid);
#define CASES(m) \
    CASE(8 * m + 0, m, false, false, false) CASE(8 * m + 1, m, false, false, true) \
    CASE(8 * m + 2, m, false, true, false)  CASE(8 * m + 3, m, false, true, true)  \
    CASE(8 * m + 4, m, true, false, false)  CASE(8 * m + 5, m, true, false, true)  \
    CASE(8 * m + 6, m, true, true, false)   CASE(8 * m + 7, m, true, true, true)
    CASES(CLSIMHIP_LENGTHS_CONSTANT) CASES(CLSIMHIP_LENGTHS_ICECUBE) CASES(CLSIMHIP_LENGTHS_TABLE)
#undef CASES
#undef CASE
    }
    return hipErrorInvalidValue;
}

#elif defined(CLSIMHIP_KEEP_UNIT)
// propagation without STOP_PHOTONS_ON_DETECTION (TAB = 3)
hipError_t launch_keep_kernel(const KParams &P, const KVariant &v, hipStream_t stream)
{
    if (P.n_steps == 0) return hipSuccess;
    if (!v.keep_detected || v.tabulate) return hipErrorInvalidValue;
    if (v.lengths < CLSIMHIP_LENGTHS_CONSTANT || v.lengths > CLSIMHIP_LENGTHS_TABLE) return hipErrorInvalidValue;
    if (v.lengths == CLSIMHIP_LENGTHS_TABLE && (!P.len_table || P.len_tab_n < 2)) return hipErrorInvalidValue;
    const int key = 8 * v.lengths + (v.tilt ? 4 : 0) + (v.aniso ? 2 : 0) + (v.flasher ? 1 : 0);
    const bool fast = v.fast && P.history_n == 0 && !v.generic_only;       // (as launch_prop_kernel)
    switch (key) {
#define CASE(k, a, b, c, d) case k: return fast ? launch_variant<a, b, c, d, 3, true>(P, stream, v.grid) : launch_variant<a, b, c, d, 3, false>(P, stream, v.grid);
#define CASES(m) \
    CASE(8 * m + 0, m, false, false, false) CASE(8 * m + 1, m, false, false, true) \
    CASE(8 * m + 2, m, false, true, false)  CASE(8 * m + 3, m, false, true, true)  \
    CASE(8 * m + 4, m, true, false, false)  CASE(8 * m + 5, m, true, false, true)  \
    CASE(8 * m + 6, m, true, true, false)   CASE(8 * m + 7, m, true, true, true)
    CASES(CLSIMHIP_LENGTHS_CONSTANT) CASES(CLSIMHIP_LENGTHS_ICECUBE) CASES(CLSIMHIP_LENGTHS_TABLE)
#undef CASES
#undef CASE
    }
    return hipErrorInvalidValue;
}

#else
// TABULATE variants: FLASHER is always compiled in (the source type is looked at per step)
hipError_t launch_tab_kernel(const KParams &P, const KVariant &v, hipStream_t stream)
{
    if (P.n_steps == 0) return hipSuccess;
    if (!v.tabulate || !P.tab_bins || !P.has_fixed_abs) return hipErrorInvalidValue;
    if (v.lengths < CLSIMHIP_LENGTHS_CONSTANT || v.lengths > CLSIMHIP_LENGTHS_TABLE) return hipErrorInvalidValue;
    if (v.lengths == CLSIMHIP_LENGTHS_TABLE && (!P.len_table || P.len_tab_n < 2)) return hipErrorInvalidValue;
    if (P.tab_ndim != 4 && P.tab_ndim != 5) return hipErrorInvalidValue;
    const int key = 4 * v.lengths + (v.tilt ? 2 : 0) + (v.aniso ? 1 : 0);
    // (round 4) FAST: the instantiation without the wave-uniform tests of the medium's proofs, as in the propagation kernels.  Built, tested
    // (tests/test_tabulator.py) and measured -- 200x36x100x105 table: 2.15e7 photons/s against 2.24e7 for the generic instantiation, the
    // impact-angle table 1.92e7 both (profiles/r04/tab_fast_vs_generic.txt): this kernel waits for its memory-side fp64 atomics in 55 % of
    // its wave cycles and issues vector instructions in 37 % of the slots, fewer scalar branches buy nothing and the other register
    // allocation costs.  So the generic instantiation runs; clsimhip_tabulator_set_tuning("fast_kernels", 1) selects the other one.
    const bool fast = v.fast && v.tab_fast;
    switch (key) {
#define CASE(k, m, t, a) case k: return (P.tab_ndim > 4) ? (fast ? launch_variant<m, t, a, true, 2, true>(P, stream, v.grid) : launch_variant<m, t, a, true, 2, false>(P, stream, v.grid)) \
                                                         : (fast ? launch_variant<m, t, a, true, 1, true>(P, stream, v.grid) : launch_variant<m, t, a, true, 1, false>(P, stream, v.grid));
#define CASES(m) CASE(4 * m + 0, m, false, false) CASE(4 * m + 1, m, false, true) CASE(4 * m + 2, m, true, false) CASE(4 * m + 3, m, true, true)
    CASES(CLSIMHIP_LENGTHS_CONSTANT) CASES(CLSIMHIP_LENGTHS_ICECUBE) CASES(CLSIMHIP_LENGTHS_TABLE)
#undef CASES
#undef CASE
    }
    return hipErrorInvalidValue;
}

#endif
#ifndef CLSIMHIP_TAB_UNIT
size_t prop_kernel_lds_bytes(uint32_t table_words) { return (size_t)(table_words + kWavesPerBlock * kStageRecords * kStubWords + kBlock) * 4; }
int prop_kernel_block_size() { return kBlock; }
// upper bound of the lanes of one launch (persistent grid: at most 2048 resident threads per CU)
size_t prop_kernel_max_lanes()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return (size_t)cus * 2048u;
}
// LDS bytes one workgroup may use so that the intended number of workgroups fits the CU's 160 KB
size_t prop_kernel_lds_budget() { return (size_t)(160 * 1024) / (size_t)((kMinWavesPerSimd * 256) / kBlock) - 1024; }

hipError_t launch_check_math(int what, int exp_lo, int exp_hi, uint32_t *result, uint32_t result_cap, hipStream_t stream)
{
    hipLaunchKernelGGL(check_math_kernel, dim3((1u << 23) / 256), dim3(256), 0, stream, what, exp_lo, exp_hi, result, result_cap);
    return hipGetLastError();
}

hipError_t launch_eval_math(int what, const float *xs, const float *ys, uint32_t n, float *out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(eval_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, what, xs, ys, n, out);
    return hipGetLastError();
}

#endif
} // namespace clsimhip
