// Compile(): configuration (doubles) -> kernel constants (float literals) + LDS image.
//
// The reference does this by printing OpenCL source
// (private/opencl/I3CLSimHelperGenerateMediumPropertiesSource{,_Optimizers}.cxx,
// the GetOpenCLFunction() members under private/clsim/function and random_value,
// I3CLSimStepToPhotonConverterOpenCL.cxx:390-533).  Each constant below is the
// float that source text would hold; per-layer combinations that the generated
// code recomputes on every call ((D*aDust+E), (1+0.01*deltaTau), OM_RADIUS^2,
// maxZ+OM_RADIUS ...) are folded here with the same single precision operations.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "converter.h"
#include "math_tables.h"

namespace clsimhip {
namespace {

struct Image {
    std::vector<uint32_t> words;
    uint32_t add_floats(const std::vector<float> &v)
    {
        const uint32_t off = static_cast<uint32_t>(words.size());
        for (float f : v) { uint32_t u; std::memcpy(&u, &f, 4); words.push_back(u); }
        return off;
    }
    uint32_t add_words(const std::vector<uint32_t> &v)
    {
        const uint32_t off = static_cast<uint32_t>(words.size());
        words.insert(words.end(), v.begin(), v.end());
        return off;
    }
    void align(size_t words) { while (words && words_size() % words) this->words.push_back(0u); }
    size_t words_size() const { return words.size(); }
    uint32_t add_u16(const std::vector<uint16_t> &v)
    {
        const uint32_t off = static_cast<uint32_t>(words.size());
        for (size_t i = 0; i < v.size(); i += 2) {
            const uint32_t lo = v[i], hi = (i + 1 < v.size()) ? v[i + 1] : 0xFFFFu;
            words.push_back(lo | (hi << 16));
        }
        return off;
    }
};

// Proof by exhaustion that a/b == fma(fma(-b, a*r, a), r, a*r) with r = RN(1/b) for EVERY float a: the
// three operations scale exactly with the exponent of a and are odd in a, so one binade of a covers all
// normal a (the kernel's dividends are 0 or far from the subnormal range).
bool division_by_reciprocal_is_exact(float b, float &r)
{
    r = 1.0f / b;
    if (!(b > 1e-18f && b < 1e18f)) return false;
    for (uint32_t m = 0; m < (1u << 23); ++m) {
        const uint32_t bits = 0x3f800000u | m;
        float a;
        std::memcpy(&a, &bits, 4);
        const float q = a * r;
        const float rem = std::fmaf(-b, q, a);
        const float q1 = std::fmaf(rem, r, q);
        if (q1 != a / b) return false;
    }
    return true;
}

template <class T>
std::vector<double> as_doubles(const std::vector<T> &v) { return std::vector<double>(v.begin(), v.end()); }

std::vector<float> literals(const std::vector<double> &v)
{
    std::vector<float> out(v.size());
    for (size_t i = 0; i < v.size(); ++i) out[i] = to_float_literal(v[i]);
    return out;
}

} // namespace

CompiledTables compile_tables(const MediumData &m, const GeometryInput &geometry,
                              const std::vector<RandomValueData> &generators, const FunctionData &bias, double pancake,
                              const TableTuning &tuning)
{
    m.validate();
    CompiledTables C;
    KParams &P = C.params;
    Image img;
    auto name = [&](const std::string &k, const std::vector<double> &v) { C.named[k] = v; };
    auto scalar = [&](const std::string &k, double v) { C.named[k] = std::vector<double>(1, v); };

    // ---------------- the math tables of detmath.hip.h (round 5): ALWAYS the first words of the image, at fixed offsets --
    // 32 log rows {INV, H, L, 0}, 33 sincos rows {S, C}, padded to a multiple of four words (prop_device.hip.h: lds_log, lds_sincos_2pi)
    {
        static const float log_rows[] = MT_LOG_TABLE;
        static const float sc_rows[] = MT_SC_TABLE;
        static_assert(sizeof(log_rows) == 4 * 4 * MT_LOG_ROWS && sizeof(sc_rows) == 4 * 2 * MT_SC_ROWS, "math table sizes");
        const uint32_t at_log = img.add_floats(std::vector<float>(log_rows, log_rows + 4 * MT_LOG_ROWS));
        const uint32_t at_sc = img.add_floats(std::vector<float>(sc_rows, sc_rows + 2 * MT_SC_ROWS));
        img.align(4);
        if (at_log != 0u || at_sc != 4u * MT_LOG_ROWS) throw Error(CLSIMHIP_ERR_CONFIG, "math tables are not at the front of the LDS image");
        name("math_log_table", std::vector<double>(log_rows, log_rows + 4 * MT_LOG_ROWS));
        name("math_sincos_table", std::vector<double>(sc_rows, sc_rows + 2 * MT_SC_ROWS));
    }

    // ---------------- medium (MediumPropertiesSource.cxx:207-389) ----------------
    P.num_layers = m.num_layers;
    P.layer_bottom = to_float_literal(m.layers_z_start);
    P.layer_thickness = to_float_literal(m.layers_height);
    P.recip_thickness = 1.0f / P.layer_thickness;                  // (ONE/MEDIUM_LAYER_THICKNESS), c.cl:639
    if (division_by_reciprocal_is_exact(P.layer_thickness, P.rcp_layer_thickness)) P.div_ok |= 2u;
    scalar("MEDIUM_LAYERS", m.num_layers);
    scalar("MEDIUM_LAYER_BOTTOM_POS", P.layer_bottom);
    scalar("MEDIUM_LAYER_THICKNESS", P.layer_thickness);
    P.nanometer = to_float_literal(units::nanometer);
    P.micrometer = to_float_literal(units::micrometer);
    P.c_light = to_float_literal(units::c_light);
    for (int i = 0; i < 5; ++i) { P.n[i] = to_float_literal(m.n[i]); P.g[i] = to_float_literal(m.g[i]); }
    for (int which = 0; which < 2; ++which) {
        // FromTable with float data (FromTable.cxx:200-211), one function for all layers
        const int kind = which ? m.group_kind : m.phase_kind;
        (which ? P.group_kind : P.phase_kind) = kind;
        if (kind != CLSIMHIP_REFINDEX_TABLE) continue;
        const FunctionData &f = which ? m.group_table : m.phase_table;
        const std::vector<float> data = literals(f.values);
        (which ? P.group_n : P.phase_n) = static_cast<int32_t>(data.size());
        (which ? P.group_start : P.phase_start) = to_float_literal(f.start);
        (which ? P.group_step : P.phase_step) = to_float_literal(f.step);
        (which ? P.off_group : P.off_phase) = img.add_floats(data);
        name(which ? "getGroupRefIndex_func0_data" : "getPhaseRefIndex_func0_data", as_doubles(data));
    }
    scalar("MEDIUM_MIN_WLEN", to_float_literal(m.min_wlen)); scalar("MEDIUM_MAX_WLEN", to_float_literal(m.max_wlen));
    C.variant.lengths = m.lengths_kind;
    if (m.lengths_kind == CLSIMHIP_LENGTHS_TABLE) {
        // FromTable.cxx:167-300 per layer.  With 16-bit storage the generated function rebuilds
        //   convert_float(data[bin]) * ((LARGEST-SMALLEST)/65535.f) + SMALLEST
        // on every call; that value depends on (layer, bin) only, so it is formed here with the same
        // single precision operations and the kernel only interpolates.
        const size_t nl = static_cast<size_t>(m.num_layers), nw = static_cast<size_t>(m.table_n);
        P.len_tab_n = m.table_n;
        P.len_tab_start = to_float_literal(m.table_start);
        P.len_tab_step = to_float_literal(m.table_step);
        std::vector<float> value[2];
        for (int which = 0; which < 2; ++which) {
            const std::vector<double> &src = which ? m.sca_table : m.abs_table;
            std::vector<float> &dst = value[which];
            dst.resize(nl * nw);
            std::vector<double> quantised(nl * nw), lo_hi(2 * nl);
            for (size_t l = 0; l < nl; ++l) {
                const double *v = &src[l * nw];
                if (!m.table_16bit) {
                    for (size_t i = 0; i < nw; ++i) dst[l * nw + i] = to_float_literal(v[i]);
                    continue;
                }
                double smallest = v[0], largest = v[0];
                for (size_t i = 1; i < nw; ++i) {
                    if (v[i] < smallest) smallest = v[i];
                    if (v[i] > largest) largest = v[i];
                }
                const float lo = to_float_literal(smallest), hi = to_float_literal(largest);
                const float scale = (hi - lo) / 65535.f;
                lo_hi[2 * l] = lo; lo_hi[2 * l + 1] = hi;
                for (size_t i = 0; i < nw; ++i) {
                    const double q = 65535. * (v[i] - smallest) / (largest - smallest);
                    const uint16_t stored = (q >= 0. && q < 65536.) ? static_cast<uint16_t>(q) : 0;   // static_cast<uint16_t>
                    const float t = static_cast<float>(stored) * scale;
                    dst[l * nw + i] = t + lo;
                    quantised[l * nw + i] = stored;
                }
            }
            const std::string fn = which ? "getScatteringLength" : "getAbsorptionLength";
            if (m.table_16bit) { name(fn + "_data16", quantised); name(fn + "_smallest_largest", lo_hi); }
            name(fn + "_values", as_doubles(dst));
        }
        C.len_table.assign(4 * (nw - 1) * nl, 0.f);
        for (size_t b = 0; b + 1 < nw; ++b)
            for (size_t l = 0; l < nl; ++l) {
                float *rec = &C.len_table[4 * (b * nl + l)];
                rec[0] = value[0][l * nw + b]; rec[1] = value[0][l * nw + b + 1];
                rec[2] = value[1][l * nw + b]; rec[3] = value[1][l * nw + b + 1];
            }
        P.off_layers = 0;
    } else if (m.lengths_kind == CLSIMHIP_LENGTHS_ICECUBE) {
        // _Optimizers.cxx:123-250 (the per-function form, AbsLenIceCube.cxx:70-93, has the same arithmetic)
        const std::vector<float> a_dust = literals(m.a_dust400), d_tau = literals(m.delta_tau), b400 = literals(m.b400);
        const float D = to_float_literal(m.D), E = to_float_literal(m.E);
        std::vector<float> abs_a(a_dust.size()), abs_b(a_dust.size());
        for (size_t l = 0; l < a_dust.size(); ++l) {
            const float t = D * a_dust[l];
            abs_a[l] = t + E;                                       // (D*aDust400[layer]+E)
            const float u = 0.01f * d_tau[l];
            abs_b[l] = 1.f + u;                                     // (1.f + 0.01f*deltaTau[layer])
        }
        std::vector<float> rec(4 * a_dust.size(), 0.f);
        for (size_t l = 0; l < a_dust.size(); ++l) { rec[4 * l] = abs_a[l]; rec[4 * l + 1] = abs_b[l]; rec[4 * l + 2] = b400[l]; }
        P.off_layers = img.add_floats(rec);
        P.neg_kappa = -to_float_literal(m.kappa);
        P.abs_A = to_float_literal(m.A);
        P.neg_B = -to_float_literal(m.B);
        P.neg_alpha = -to_float_literal(m.alpha);
        P.ref_wlen_recip = to_float_literal(1. / (400. * units::nanometer));
        name("aDust400", as_doubles(a_dust)); name("deltaTau", as_doubles(d_tau)); name("b400", as_doubles(b400));
        scalar("kappa", -P.neg_kappa); scalar("A", P.abs_A); scalar("B", -P.neg_B); scalar("D", D); scalar("E", E);
        scalar("alpha", -P.neg_alpha);
    } else {
        const std::vector<float> abs_c = literals(m.abs_length), sca_c = literals(m.sca_length);
        std::vector<float> rec(4 * abs_c.size(), 0.f);
        for (size_t l = 0; l < abs_c.size(); ++l) { rec[4 * l] = abs_c[l]; rec[4 * l + 2] = sca_c[l]; }
        P.off_layers = img.add_floats(rec);
        name("absorptionLength", as_doubles(abs_c)); name("scatteringLength", as_doubles(sca_c));
    }
    {   // Mixed.cxx:115-157, SimplifiedLiu.cxx:64-88, HenyeyGreenstein.cxx:69-92
        const double g = m.mean_cosine;
        P.scatter_kind = m.scatter_kind;
        P.liu_beta = to_float_literal((1. - g) / (1. + g));
        P.hg_g = to_float_literal(g);
        const float g2 = to_float_literal(g * g);
        P.hg_one_minus_g2 = 1.f - g2;
        P.hg_one_plus_g2 = 1.f + g2;
        P.hg_two_g = 2.f * P.hg_g;
        P.mix_frac = to_float_literal(m.liu_fraction);
        P.mix_frac_rest = to_float_literal(1. - m.liu_fraction);
        if (division_by_reciprocal_is_exact(P.mix_frac, P.rcp_mix_frac)) P.div_ok |= 4u;
        if (division_by_reciprocal_is_exact(P.mix_frac_rest, P.rcp_mix_frac_rest)) P.div_ok |= 8u;
        if (division_by_reciprocal_is_exact(P.hg_two_g, P.rcp_hg_two_g)) P.div_ok |= 16u;
        scalar("liu_beta", P.liu_beta); scalar("hg_g", P.hg_g); scalar("hg_g2", g2);
        scalar("mix_frac", P.mix_frac); scalar("mix_frac_rest", P.mix_frac_rest);
    }
    C.variant.aniso = m.has_aniso || m.has_pre || m.has_post;
    P.abs_corr_const = to_float_literal(1.);
    P.has_abs_corr = m.has_aniso ? 1 : 0;
    if (m.has_aniso) {
        // ScalarFieldAnisotropyAbsLenScaling.cxx:96-108
        const double azx = std::cos(m.aniso_azimuth), azy = std::sin(m.aniso_azimuth);
        const double k1 = std::exp(m.aniso_k1), k2 = std::exp(m.aniso_k2), kz = 1. / (k1 * k2);
        const double l1 = k1 * k1, l2 = k2 * k2, l3 = kz * kz;
        const double B2 = 1. / l1 + 1. / l2 + 1. / l3;
        P.an_l[0] = to_float_literal(l1); P.an_l[1] = to_float_literal(l2); P.an_l[2] = to_float_literal(l3);
        P.an_rl[0] = to_float_literal(1. / l1); P.an_rl[1] = to_float_literal(1. / l2); P.an_rl[2] = to_float_literal(1. / l3);
        P.an_azx = to_float_literal(azx); P.an_azy = to_float_literal(azy); P.an_mazy = to_float_literal(-azy);
        P.an_B2 = to_float_literal(B2);
        name("anisotropy", {P.an_l[0], P.an_l[1], P.an_l[2], P.an_rl[0], P.an_rl[1], P.an_rl[2], P.an_azx, P.an_azy, P.an_mazy, P.an_B2});
        // div_ok bit 7: the divisor (B2 - nB) * An of the correction factor stays far inside the exponent range.  For a
        // unit direction nB and An are convex combinations of the 1/l_i and of the l_i, so the divisor lies between
        // (B2 - max 1/l) * min l and (B2 - min 1/l) * max l.
        const double rl_lo = std::min({1. / l1, 1. / l2, 1. / l3}), rl_hi = std::max({1. / l1, 1. / l2, 1. / l3});
        const double x_lo = (B2 - rl_hi) * std::min({l1, l2, l3}), x_hi = (B2 - rl_lo) * std::max({l1, l2, l3});
        if (x_lo > 1e-10 && x_hi < 1e10 && std::isfinite(x_hi)) P.div_ok |= 128u;
    }
    {   // div_ok bit 6: renormalisation after the direction transforms may use dm::rsqrt_near_: |M d|^2 of a unit d lies
        // within [sigma_min^2, sigma_max^2], sigma_max <= |M|_F, sigma_min >= |det M| / |M|_F^2
        bool ok = true;
        for (int which = 0; which < 2; ++which) {
            if (!(which ? (m.has_post && m.post_renorm) : (m.has_pre && m.pre_renorm))) continue;
            const double *M = which ? m.post : m.pre;
            double fro2 = 0.;
            for (int i = 0; i < 9; ++i) fro2 += M[i] * M[i];
            const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
            if (!(fro2 < 1e16) || !(std::abs(det) / fro2 > 1e-8)) ok = false;
        }
        if (ok) P.div_ok |= 64u;
    }
    P.has_pre = m.has_pre ? 1 : 0; P.pre_renorm = m.pre_renorm ? 1 : 0;
    P.has_post = m.has_post ? 1 : 0; P.post_renorm = m.post_renorm ? 1 : 0;
    for (int i = 0; i < 9; ++i) { P.pre[i] = to_float_literal(m.pre[i]); P.post[i] = to_float_literal(m.post[i]); }
    if (m.has_pre) name("transformDirectionPreScatter", std::vector<double>(P.pre, P.pre + 9));
    if (m.has_post) name("transformDirectionPostScatter", std::vector<double>(P.post, P.post + 9));
    C.variant.tilt = m.has_tilt;
    P.tilt_const = to_float_literal(0.);
    if (m.has_tilt) {
        // ScalarFieldIceTiltZShift.cxx:62-100 (constructor) and :145-213
        const size_t nd = m.tilt_distances.size(), nz = m.tilt_z.size();
        // (interpolation between neighbours in both dimensions, :170-205: a table with a single row or column has none; the
        // kernel's index clamps rely on nz - 2 >= 0)
        if (nd < 2 || nz < 2) throw Error(CLSIMHIP_ERR_ARGUMENT, "the ice tilt table needs at least two distances and two z coordinates");
        double mean_spacing = 0.;
        for (size_t i = 0; i + 1 < nz; ++i) {
            const double sp = m.tilt_z[i + 1] - m.tilt_z[i];
            if (sp <= 0.) throw Error(CLSIMHIP_ERR_ARGUMENT, "zCoordinates (dimension 2) are not in ascending order.");
            mean_spacing += sp;
        }
        mean_spacing /= static_cast<double>(nz - 1);
        for (size_t i = 0; i + 1 < nz; ++i)
            if (std::abs((m.tilt_z[i + 1] - m.tilt_z[i]) - mean_spacing) > 1e-5)
                throw Error(CLSIMHIP_ERR_ARGUMENT, "zCoordinates (dimension 2) are not in equally spaced.");
        for (size_t i = 0; i + 1 < nd; ++i)
            if (m.tilt_distances[i + 1] - m.tilt_distances[i] <= 0.)
                throw Error(CLSIMHIP_ERR_ARGUMENT, "distancesFromOriginAlongTilt (dimension 1) is not in ascending order.");
        P.tilt_nd = static_cast<int>(nd);
        P.tilt_nz = static_cast<int>(nz);
        P.tilt_first_z = to_float_literal(m.tilt_z[0]);
        P.tilt_dz = to_float_literal(mean_spacing);
        P.tilt_lnx = to_float_literal(std::cos(m.tilt_azimuth));
        P.tilt_lny = to_float_literal(std::sin(m.tilt_azimuth));
        const std::vector<float> dist = literals(m.tilt_distances), corr = literals(m.tilt_corr);
        // the kernel counts bins on the literals: they must stay strictly ordered after rounding
        for (size_t i = 0; i + 1 < nd; ++i)
            if (!(dist[i] < dist[i + 1])) throw Error(CLSIMHIP_ERR_CONFIG, "tilt distances collapse in single precision");
        P.off_tilt_dist = img.add_floats(dist);
        P.off_tilt_zcorr = img.add_floats(corr);
        if (division_by_reciprocal_is_exact(P.tilt_dz, P.rcp_tilt_dz)) P.div_ok |= 1u;
        for (size_t t6 = 0; t6 < 6; ++t6) P.tilt_inner_dist[t6] = (t6 + 1 < nd - 1) ? dist[t6 + 1] : INFINITY;
        {
            std::vector<uint32_t> bins(4 * nd, 0u);
            for (size_t j = 1; j < nd; ++j) {
                const float width = dist[j] - dist[j - 1];          // thisDistanceBinWidth
                float rcp = 0.f;
                const bool ok = division_by_reciprocal_is_exact(width, rcp);
                std::memcpy(&bins[4 * j], &dist[j], 4);
                std::memcpy(&bins[4 * j + 1], &width, 4);
                std::memcpy(&bins[4 * j + 2], &rcp, 4);
                bins[4 * j + 3] = ok ? 1u : 0u;
            }
            img.align(4);
            P.off_tilt_bins = img.add_words(bins);
        }
        name("getTiltZShift_data_distancesFromOriginAlongTilt", as_doubles(dist));
        name("getTiltZShift_data_zCorrections", as_doubles(corr));
        scalar("getTiltZShift_data_firstZCoord", P.tilt_first_z);
        scalar("getTiltZShift_data_zCoordSpacing", P.tilt_dz);
        scalar("getTiltZShift_lnx", P.tilt_lnx); scalar("getTiltZShift_lny", P.tilt_lny);
    }

    // ---------------- spectra ----------------
    if (generators.empty()) throw Error(CLSIMHIP_ERR_CONFIG, "no wavelength generator set");
    if (generators.size() > static_cast<size_t>(kMaxGenerators)) throw Error(CLSIMHIP_ERR_CONFIG, "too many wavelength generators");
    P.num_gen = static_cast<int>(generators.size());
    C.variant.flasher = generators.size() > 1;                     // NO_FLASHER, OpenCL.cxx:648-650
    for (size_t k = 0; k < generators.size(); ++k) {
        const RandomValueData &g = generators[k];
        P.gen_kind[k] = g.kind;
        if (g.kind == CLSIMHIP_RANDOM_CONSTANT) {
            P.gen_value[k] = to_float_literal(g.value);
            continue;
        }
        if (g.kind == CLSIMHIP_RANDOM_CHERENKOV_NO_DISPERSION) {
            // WlenCherenkovNoDispersion.cxx:72-92: 1.f/(minVal + r * range) with the two literals
            const double min_val = 1. / g.spacing, range = (1. / g.first) - min_val;
            P.gen_first[k] = to_float_literal(min_val);
            P.gen_spacing[k] = to_float_literal(range);
            continue;
        }
        // InterpolatedDistribution.cxx:134-175 (InitTables) and :177-234 (WriteTableCode)
        const size_t n = g.y.size();
        const bool own_x = (g.kind == CLSIMHIP_RANDOM_INTERPOLATED_X);
        if (n < 2) throw Error(CLSIMHIP_ERR_ARGUMENT, "At least two entries have to be specified for an interpolated distribution.");
        if (own_x && g.x.size() != n) throw Error(CLSIMHIP_ERR_ARGUMENT, "The \"x\" and \"y\" vectors must have the same size!");
        if (!own_x && !(g.spacing > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "\"xSpacing\" must not be <= 0!");
        std::vector<double> acu(n, 0.), beta(n, 0.);
        if (own_x) for (size_t j = 1; j < n; ++j) acu[j] = acu[j - 1] + (g.x[j] - g.x[j - 1]) * (g.y[j] + g.y[j - 1]) / 2.;     // :148-154
        else for (size_t j = 1; j < n; ++j) acu[j] = acu[j - 1] + (g.spacing) * (g.y[j] + g.y[j - 1]) / 2.;
        const double total = acu[n - 1];
        for (size_t j = 0; j < n; ++j) { beta[j] = g.y[j] / total; acu[j] = acu[j] / total; }
        const std::vector<float> yv = literals(beta), ycum = literals(acu);
        for (size_t j = 0; j + 1 < n; ++j)
            if (ycum[j] > ycum[j + 1]) throw Error(CLSIMHIP_ERR_CONFIG, "cumulative spectrum is not monotonic (negative density?)");
        P.gen_n[k] = static_cast<int>(n);
        P.gen_first[k] = own_x ? 0.f : to_float_literal(g.first);
        P.gen_spacing[k] = own_x ? 0.f : to_float_literal(g.spacing);
        P.off_gen_yv[k] = img.add_floats(yv);
        P.off_gen_ycum[k] = img.add_floats(ycum);
        const std::string prefix = "_generateWavelength_" + std::to_string(k);
        if (own_x) {                                        // WriteTableCode, :203-211
            const std::vector<float> xv = literals(g.x);
            for (size_t j = 0; j + 1 < n; ++j)
                if (!(xv[j] < xv[j + 1])) throw Error(CLSIMHIP_ERR_CONFIG, "the wavelengths of a distribution do not ascend in single precision");
            P.off_gen_xv[k] = img.add_floats(xv);
            name(prefix + "distXValues", as_doubles(xv));
        }
        name(prefix + "distYValues", as_doubles(yv));
        name(prefix + "distYCumulativeValues", as_doubles(ycum));
    }
    {   // div_ok bit 5: every scattering / absorption length a photon can meet is far inside the exponent range, so the
        // kernel may form the reciprocals of the layer walk (1/(b400 * ...), 1/length) with dm::rcp_ -- v_rcp_f32 + one
        // Newton step, RN(1/x) for every x in [2^-100, 2^100] (tests/test_detmath_gpu.py runs all of them) -- instead
        // of the IEEE divide sequence.  Bounds in double over the wavelengths the generators can produce (widened 2x).
        double w_lo = 1e300, w_hi = 0.;
        for (const RandomValueData &g : generators) {
            double a = g.value, b = g.value;
            if (g.kind == CLSIMHIP_RANDOM_CHERENKOV_NO_DISPERSION) { a = std::min(g.first, g.spacing); b = std::max(g.first, g.spacing); }
            else if (g.kind == CLSIMHIP_RANDOM_INTERPOLATED_X) { a = g.x.empty() ? 0. : g.x.front(); b = g.x.empty() ? 0. : g.x.back(); }
            else if (g.kind != CLSIMHIP_RANDOM_CONSTANT) { a = g.first - g.spacing; b = g.first + g.spacing * static_cast<double>(g.y.size()); }
            w_lo = std::min(w_lo, a); w_hi = std::max(w_hi, b);
        }
        w_lo *= 0.5; w_hi *= 2.;
        double len_lo = 1e300, len_hi = 0.;
        auto see = [&](double len) { if (!(len > 0.) || !std::isfinite(len)) { len_lo = 0.; return; } len_lo = std::min(len_lo, len); len_hi = std::max(len_hi, len); };
        if (!(w_lo > 0.) || !std::isfinite(w_hi)) len_lo = 0.;
        else if (m.lengths_kind == CLSIMHIP_LENGTHS_TABLE) { for (double v : m.abs_table) see(v); for (double v : m.sca_table) see(v); }
        else if (m.lengths_kind == CLSIMHIP_LENGTHS_CONSTANT) { for (double v : m.abs_length) see(v); for (double v : m.sca_length) see(v); }
        else {
            const int grid = 64;
            for (int i = 0; i <= grid; ++i) {
                const double w = w_lo * std::pow(w_hi / w_lo, static_cast<double>(i) / grid), x = w / units::nanometer;
                for (size_t l = 0; l < m.b400.size(); ++l) {
                    see(1. / (m.b400[l] * std::pow(w / (400. * units::nanometer), -m.alpha)));
                    see(1. / ((m.D * m.a_dust400[l] + m.E) * std::pow(x, -m.kappa) + m.A * std::exp(-m.B / x) * (1. + 0.01 * m.delta_tau[l])));
                }
            }
        }
        if (len_lo > 1e-15 && len_hi < 1e15) P.div_ok |= 32u;
        scalar("length_bounds_lo", len_lo); scalar("length_bounds_hi", len_hi);
    }
    // a wavelength bias the DEVICE evaluates: an equally spaced table or a constant.  The host-only kinds (a table with its own
    // wavelengths, a delta peak: spectra for make_wlen_generator / the flasher front end) have no device form -- the reference's
    // FromTable throws for unequal spacing when its OpenCL code is asked for (FromTable.cxx:169-170).  Every caller ends here
    // (converter, table maker), so this is where it is refused.
    if (!bias.on_device())
        throw Error(CLSIMHIP_ERR_ARGUMENT, "the wavelength bias / acceptance must be a table with equal spacing or a constant (FromTable.cxx:169-170)");
    P.bias_kind = bias.kind;
    if (bias.kind == CLSIMHIP_FUNCTION_TABLE) {
        // FunctionFromTable.cxx:167-300
        if (bias.values.size() < 2) throw Error(CLSIMHIP_ERR_ARGUMENT, "values must contain at least 2 elements!");
        if (!(bias.step > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "wlenStep must not be <= 0!");
        const std::vector<float> data = literals(bias.values);
        P.bias_n = static_cast<int>(data.size());
        P.bias_start = to_float_literal(bias.start);
        P.bias_step = to_float_literal(bias.step);
        P.off_bias = img.add_floats(data);
        name("getWavelengthBias_data", as_doubles(data));
    } else {
        P.bias_value = to_float_literal(bias.value);
    }

    // KVariant::fast (prop_device.hip.h: FAST), the medium's part: the standard configuration with every proof in hand
    auto medium_proofs_complete = [&]() {
        bool fast = (m.scatter_kind == CLSIMHIP_SCATTER_MIXED) && (m.group_kind != CLSIMHIP_REFINDEX_DISPERSION) && (P.liu_beta <= 0.09f) && ((P.div_ok & (2u | 4u | 8u | 16u | 32u | 64u)) == (2u | 4u | 8u | 16u | 32u | 64u));
        if (m.has_aniso && !(P.div_ok & 128u)) fast = false;
        // hg_cos divides 1 - g^2 by 1 + g s, |s| <= 1, with dm::div_near_: numerator >= 2^-40, divisors in [2^-50, 2^50]
        if (!(P.hg_one_minus_g2 >= 9.094947017729282e-13f) || !(1.0f - std::abs(P.hg_g) >= 8.881784197001252e-16f) || !(std::abs(P.hg_g) <= 1.0f)) fast = false;
        if (m.has_tilt) {
            if (!(P.div_ok & 1u) || m.tilt_distances.size() > 8) fast = false;      // kTiltScalarBins + 2
            for (size_t j = 1; j < m.tilt_distances.size(); ++j) {
                float rcp = 0.f;
                const float lo = to_float_literal(m.tilt_distances[j - 1]), hi = to_float_literal(m.tilt_distances[j]);
                if (!division_by_reciprocal_is_exact(hi - lo, rcp)) fast = false;
            }
        }
        return fast;
    };
    if (geometry.string_ids.empty()) {
        // the tabulator has no detector (tabulator/I3CLSimStepToTableConverter.cxx:196-207 assembles no geometry source)
        C.variant.fast = medium_proofs_complete();         // (round 4: the TABULATE kernels have FAST instantiations too)
        scalar("fast_variant", C.variant.fast ? 1. : 0.);
        P.table_words = static_cast<uint32_t>(img.words.size());
        C.lds_image = std::move(img.words);
        return C;
    }
    // ---------------- detector (GeometrySource.cxx) ----------------
    C.geo = build_geometry(geometry);
    const GeoTables &G = C.geo;
    if (G.cells.size() > static_cast<size_t>(kMaxSubdetectors)) throw Error(CLSIMHIP_ERR_CONFIG, "more than 9 subdetectors are currently not supported.");
    P.has_pancake = (pancake != 1.) ? 1 : 0;                       // OpenCL.cxx:432
    P.pancake = to_float_literal(pancake);
    P.unpancake = (P.pancake - 1.f) / P.pancake;                   // c.cl:351
    P.om_radius = G.om_radius;
    P.om_radius_sq = G.om_radius * G.om_radius;                    // OM_RADIUS*OM_RADIUS, collision c.cl:118
    P.string_max_radius_sq = G.string_max_radius * G.string_max_radius;   // sqr(GEO_STRING_MAX_RADIUS), collision c.cl:64
    P.string_max_radius = G.string_max_radius;
    P.num_strings = G.num_strings;
    P.num_sets = G.num_sets;
    P.max_layers = G.max_layers;
    P.num_subdet = static_cast<int>(G.cells.size());
    std::vector<float> top(G.num_strings), bottom(G.num_strings);
    std::vector<uint32_t> info(G.num_strings);
    for (int s = 0; s < G.num_strings; ++s) {
        top[s] = G.str_maxz[s] + G.om_radius;                      // collision c.cl:68-69
        bottom[s] = G.str_minz[s] - G.om_radius;
        if (G.dom_start[s] >= (1u << 24)) throw Error(CLSIMHIP_ERR_CONFIG, "too many DOMs");
        info[s] = static_cast<uint32_t>(G.str_set[s]) | (G.dom_start[s] << 8);
    }
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    img.align(4);
    {
        std::vector<uint32_t> rec(8 * static_cast<size_t>(G.num_strings), 0u);
        for (int s = 0; s < G.num_strings; ++s) {
            rec[8 * s + 0] = bits(G.str_x[s]); rec[8 * s + 1] = bits(G.str_y[s]);
            rec[8 * s + 2] = bits(top[s]); rec[8 * s + 3] = bits(bottom[s]);
            rec[8 * s + 4] = info[s];
            rec[8 * s + 5] = bits(G.dom_meanx[s]); rec[8 * s + 6] = bits(G.dom_meany[s]);
        }
        P.off_strings = img.add_words(rec);
    }
    {
        std::vector<uint32_t> rec(4 * static_cast<size_t>(G.num_sets), 0u);
        for (int k = 0; k < G.num_sets; ++k) {
            rec[4 * k + 0] = G.set_nlayers[k]; rec[4 * k + 1] = bits(G.set_startz[k]); rec[4 * k + 2] = bits(G.set_height[k]);
        }
        P.off_sets = img.add_words(rec);
    }
    P.off_layer_to_om = img.add_u16(G.layer_to_om);
    std::vector<uint32_t> subdet(12 * G.cells.size(), 0u);
    for (size_t k = 0; k < G.cells.size(); ++k) {
        const GeoTables::Cells &c = G.cells[k];
        subdet[12 * k + 0] = static_cast<uint32_t>(c.nx); subdet[12 * k + 1] = static_cast<uint32_t>(c.ny);
        subdet[12 * k + 2] = bits(c.wx); subdet[12 * k + 3] = bits(c.wy);
        subdet[12 * k + 4] = bits(c.sx); subdet[12 * k + 5] = bits(c.sy);
        subdet[12 * k + 6] = img.add_u16(c.index);
        float rwx = 0.f, rwy = 0.f;
        subdet[12 * k + 7] = (division_by_reciprocal_is_exact(c.wx, rwx) ? 1u : 0u) | (division_by_reciprocal_is_exact(c.wy, rwy) ? 2u : 0u);
        subdet[12 * k + 8] = bits(rwx); subdet[12 * k + 9] = bits(rwy);
        const std::string sfx = "_" + std::to_string(k);
        name("geoCellIndex" + sfx, as_doubles(c.index));
        name("GEO_CELL" + sfx, {double(c.nx), double(c.ny), c.wx, c.wy, c.sx, c.sy});
    }
    img.align(4);
    P.off_subdet = img.add_words(subdet);
    P.dom_mul_x = G.dom_mul_x;
    P.dom_mul_y = G.dom_mul_y;
    {   // DOM templates go to LDS too when the whole image stays within the budget of two workgroups per CU
        std::vector<uint32_t> xy(G.dom_tx.size());
        for (size_t i = 0; i < xy.size(); ++i)
            xy[i] = static_cast<uint32_t>(static_cast<uint16_t>(G.dom_tx[i])) | (static_cast<uint32_t>(static_cast<uint16_t>(G.dom_ty[i])) << 16);
        const size_t with_doms = img.words.size() + 2 * xy.size();
        if (prop_kernel_lds_bytes(static_cast<uint32_t>(with_doms)) <= prop_kernel_lds_budget()) {
            P.dom_in_lds = 1;
            P.off_dom_xy = img.add_words(xy);
            P.off_dom_z = img.add_floats(G.dom_tz);
        }
    }
    {   // string proximity map (kparams.h).  Everything in double, rounded towards "search anyway".
        // Resolution: the bound of a cell is taken at its far corner, so a 10 m cell (128^2) gives away up to 14 m of free flight.
        // That hardly mattered (32^2 ... 1024^2 within 1.5 %) while a lane that got through the first level cost a DOM-map look-up
        // anyway; since the level behind it asks whether the photon is aimed at the string (segment_misses_string) it does:
        // 1M cascade steps 64^2 3.60e9 photons/s, 128^2 3.70, 256^2 3.73, 512^2 3.77, 768^2 3.78, 1024^2 3.78, 2048^2 (16 MB) 3.42;
        // SPICE-Lea 3.09 / 3.15 / 3.19 / 3.21 / - / 3.22.  512^2 words = 1 MB, L2 resident.  (A 64^2 copy in LDS was 1.5 % SLOWER
        // than the L2-resident map: the load is issued before the layer walk and is long back when it is needed.)
        const int n = std::max(8, std::min(4096, tuning.string_map_cells));         // 512 unless clsimhip_set_tuning("string_map_cells") said otherwise
        double x_lo = INFINITY, x_hi = -INFINITY, y_lo = INFINITY, y_hi = -INFINITY, reach = 0.;
        for (int s = 0; s < G.num_strings; ++s) {
            x_lo = std::min<double>(x_lo, G.str_x[s]); x_hi = std::max<double>(x_hi, G.str_x[s]);
            y_lo = std::min<double>(y_lo, G.str_y[s]); y_hi = std::max<double>(y_hi, G.str_y[s]);
        }
        // largest xy offset of a DOM (as the kernel reconstructs it from the int16 templates) from its string's axis
        for (int s = 0; s < G.num_strings; ++s) {
            // (strings with the same DOM offsets share one template: dom_start is not a running sum; a string's DOMs are
            // the entries dom_start[s] ... + its number of DOMs)
            const size_t first = G.dom_start[s], last = first + G.dom_index_to_id[static_cast<size_t>(s)].size();
            for (size_t i = first; i < last; ++i) {
                const double dx = double(G.dom_tx[i]) * G.dom_mul_x + G.dom_meanx[s] - G.str_x[s];
                const double dy = double(G.dom_ty[i]) * G.dom_mul_y + G.dom_meany[s] - G.str_y[s];
                reach = std::max(reach, std::sqrt(dx * dx + dy * dy));
            }
        }
        reach += double(G.om_radius) + 0.05;                        // DOM sphere (never pancaked laterally) + safety
        const double margin = 30.;                                  // the map extends a little beyond the outer strings
        x_lo -= margin; y_lo -= margin; x_hi += margin; y_hi += margin;
        const double cell = std::max(std::max(x_hi - x_lo, y_hi - y_lo) / n, 0.5);
        P.prox_n = n;
        P.prox_x0 = static_cast<float>(x_lo);
        P.prox_y0 = static_cast<float>(y_lo);
        P.prox_inv_cell = static_cast<float>(1. / cell);
        P.prox_reach = static_cast<float>(reach * 1.00001);
        // cell the kernel computes for a point: (int)((x - x0) * inv_cell) in float; each cell is grown by `slack`
        // for that arithmetic (relative error < 4e-7 of |x| + |x0|, n cells) and border cells reach to infinity
        const double x0f = P.prox_x0, y0f = P.prox_y0, cellf = 1. / double(P.prox_inv_cell);
        const double slack = 1e-3 * cellf + 1e-5 * (std::fabs(x0f) + std::fabs(y0f) + n * cellf);
        C.prox_map.assign(static_cast<size_t>(n) * n, 0);
        for (int iy = 0; iy < n; ++iy)
            for (int ix = 0; ix < n; ++ix) {
                const double rx0 = (ix == 0) ? -INFINITY : x0f + ix * cellf - slack, rx1 = (ix == n - 1) ? INFINITY : x0f + (ix + 1) * cellf + slack;
                const double ry0 = (iy == 0) ? -INFINITY : y0f + iy * cellf - slack, ry1 = (iy == n - 1) ? INFINITY : y0f + (iy + 1) * cellf + slack;
                double nearest = INFINITY, second = INFINITY;
                int which = -1;
                for (int s = 0; s < G.num_strings; ++s) {
                    const double ax = G.str_x[s], ay = G.str_y[s];
                    const double dx = std::max(std::max(rx0 - ax, ax - rx1), 0.), dy = std::max(std::max(ry0 - ay, ay - ry1), 0.);
                    const double d = std::sqrt(dx * dx + dy * dy);
                    if (d < nearest) { second = nearest; nearest = d; which = s; }
                    else if (d < second) second = d;
                }
                // a segment of length L moves at most L * |d_xy| <= L * (1 + 1e-5) in xy
                auto quantised = [&](double distance) {
                    const double q = std::floor(((distance - reach) / 1.00001) / 0.25);
                    return static_cast<uint32_t>(q < 0. ? 0. : (q > 255. ? 255. : q));
                };
                // (the string the cell names is the one whose bound is `nearest`; every other string is at least `second` away from
                // every point of the cell)
                const uint32_t id = (which >= 0 && which < 0xffff) ? static_cast<uint32_t>(which) : 0xffffu;
                C.prox_map[static_cast<size_t>(iy) * n + ix] = quantised(nearest) | ((id == 0xffffu ? 0u : quantised(second)) << 8) | (id << 16);
            }
        name("string_proximity_map", as_doubles(C.prox_map));
        name("STRING_PROXIMITY_GRID", {double(n), P.prox_x0, P.prox_y0, P.prox_inv_cell, reach, double(P.prox_reach)});
    }
    {   // DOM proximity map (kparams.h), the second level of the search filter.  Everything in double, rounded towards
        // "search anyway".  Cubic cells; at most 256 per axis (64 MB of words here, 256 MB as the device's 16-byte cells); border cells reach to infinity.
        const int n_max = std::max(4, std::min(512, tuning.dom_map_cells));         // 256 unless clsimhip_set_tuning("dom_map_cells") said otherwise
        // DOM numbers of the maps: strings in index order, each with its DOMs in order (NOT the template index: strings with
        // equal DOM offsets share a template, GeometrySource.cxx:449-495)
        std::vector<size_t> first_dom(static_cast<size_t>(G.num_strings) + 1, 0);
        for (int s = 0; s < G.num_strings; ++s) first_dom[s + 1] = first_dom[s] + G.dom_index_to_id[static_cast<size_t>(s)].size();
        const size_t n_doms = first_dom[static_cast<size_t>(G.num_strings)];
        if (n_doms >= 0xffffu) throw Error(CLSIMHIP_ERR_CONFIG, "more than 65534 DOMs");     // (GEO_MAX_DOM_INDEX is a ushort in the reference too)
        std::vector<double> dx(n_doms), dy(n_doms), dz(n_doms);
        C.dom_centres.assign(4 * n_doms, 0.f);
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int s = 0; s < G.num_strings; ++s) {
            for (size_t i = first_dom[s]; i < first_dom[s + 1]; ++i) {
                // the position the kernel reconstructs (dom_position), same float operations
                const size_t t = G.dom_start[s] + (i - first_dom[s]);           // entry of the string's template
                const float fx = static_cast<float>(G.dom_tx[t]) * G.dom_mul_x + G.dom_meanx[s];
                const float fy = static_cast<float>(G.dom_ty[t]) * G.dom_mul_y + G.dom_meany[s];
                C.dom_centres[4 * i] = fx; C.dom_centres[4 * i + 1] = fy; C.dom_centres[4 * i + 2] = G.dom_tz[t];
                dx[i] = fx; dy[i] = fy; dz[i] = G.dom_tz[t];
                const double p[3] = {dx[i], dy[i], dz[i]};
                for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
            }
        }
        const double margin = 30., range = 63.75;                   // a byte in 0.25 m units
        double extent = 0.;
        for (int k = 0; k < 3; ++k) { lo[k] -= margin; hi[k] += margin; extent = std::max(extent, hi[k] - lo[k]); }
        const double cell = std::max(extent / n_max, 1.0);
        P.dprox_inv_cell = static_cast<float>(1. / cell);
        P.dprox_x0 = static_cast<float>(lo[0]); P.dprox_y0 = static_cast<float>(lo[1]); P.dprox_z0 = static_cast<float>(lo[2]);
        const double cellf = 1. / double(P.dprox_inv_cell);
        const double o[3] = {double(P.dprox_x0), double(P.dprox_y0), double(P.dprox_z0)};
        int nn[3];
        for (int k = 0; k < 3; ++k) nn[k] = std::max(1, std::min(n_max, static_cast<int>(std::ceil((hi[k] - o[k]) / cellf))));
        P.dprox_nx = nn[0]; P.dprox_ny = nn[1]; P.dprox_nz = nn[2];
        // cell the kernel computes for a point: (int)((x - x0) * inv_cell) in float; each cell is grown by `slack` for
        // that arithmetic
        const double slack = 1e-3 * cellf + 1e-5 * (std::fabs(o[0]) + std::fabs(o[1]) + std::fabs(o[2]) + n_max * cellf);
        const double radius = double(G.om_radius) + 0.05;            // DOM sphere + safety
        P.dprox_radius = static_cast<float>(radius);
        const size_t cells = static_cast<size_t>(nn[0]) * nn[1] * nn[2];
        // per cell: quantised bound of the nearest DOM with its index, and of the second nearest
        std::vector<uint8_t> q1(cells, 255), q2(cells, 255);
        std::vector<uint16_t> id1(cells, 0xffffu);
        auto axis_gap = [&](int k, int i, double a) {                // distance along axis k from coordinate a to (grown) cell i
            const double r0 = (i == 0) ? -INFINITY : o[k] + i * cellf - slack, r1 = (i == nn[k] - 1) ? INFINITY : o[k] + (i + 1) * cellf + slack;
            return std::max(std::max(r0 - a, a - r1), 0.);
        };
        for (size_t d = 0; d < n_doms; ++d) {
            const double p[3] = {dx[d], dy[d], dz[d]};
            int i0[3], i1[3];
            for (int k = 0; k < 3; ++k) {
                i0[k] = std::max(0, static_cast<int>(std::floor((p[k] - range - radius - slack - o[k]) / cellf)) - 1);
                i1[k] = std::min(nn[k] - 1, static_cast<int>(std::floor((p[k] + range + radius + slack - o[k]) / cellf)) + 1);
            }
            // z runs fastest: the lanes that get here are near a string, and a string's cells are then a few contiguous
            // columns (about 1 KB each) that stay in the L2 instead of one cache line per cell
            for (int ix = i0[0]; ix <= i1[0]; ++ix) {
                const double gx = axis_gap(0, ix, p[0]);
                for (int iy = i0[1]; iy <= i1[1]; ++iy) {
                    const double gy = axis_gap(1, iy, p[1]);
                    const size_t row = (static_cast<size_t>(ix) * nn[1] + iy) * nn[2];
                    for (int iz = i0[2]; iz <= i1[2]; ++iz) {
                        const double gz = axis_gap(2, iz, p[2]);
                        const double bound = (std::sqrt(gx * gx + gy * gy + gz * gz) - radius) / 1.00001;
                        const double q = std::floor(bound / 0.25);
                        const uint8_t v = static_cast<uint8_t>(q < 0. ? 0. : (q > 255. ? 255. : q));
                        const size_t c = row + iz;
                        if (v < q1[c]) { q2[c] = q1[c]; q1[c] = v; id1[c] = static_cast<uint16_t>(d); }
                        else if (v < q2[c]) q2[c] = v;
                    }
                }
            }
        }
        // Where the reference's search meets each DOM (find_collision_named): its string's cell in the subdetector's grid and
        // the z layers of the string's layering that name it.  A DOM whose string's cells are not one full rectangle of one
        // subdetector's grid, or whose layers are not one contiguous run, is marked 0xffffffff and always takes the full search.
        C.dom_named.assign(4 * n_doms, 0u);
        {
            // clsimhip_set_tuning("named_search", 0): every DOM takes the full search (tests compare the two on whole bunches)
            const bool no_named = !tuning.named_search;
            // a string lies in every cell its bounding square overlaps (GeometrySource.cxx:135-271): a rectangle of cells
            struct Rect { int x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1, count = 0, sd = -1; bool bad = false; };
            std::vector<Rect> rect(static_cast<size_t>(G.num_strings));
            for (size_t k = 0; k < G.cells.size(); ++k) {
                const GeoTables::Cells &c = G.cells[k];
                for (int cy = 0; cy < c.ny; ++cy)
                    for (int cx = 0; cx < c.nx; ++cx) {
                        const uint16_t str = c.index[static_cast<size_t>(cy) * c.nx + cx];
                        if (str == 0xFFFFu || str >= G.num_strings) continue;
                        Rect &r = rect[str];
                        if (r.sd >= 0 && r.sd != static_cast<int>(k)) r.bad = true;       // (a string belongs to one subdetector)
                        r.sd = static_cast<int>(k);
                        r.x0 = std::min(r.x0, cx); r.x1 = std::max(r.x1, cx); r.y0 = std::min(r.y0, cy); r.y1 = std::max(r.y1, cy);
                        ++r.count;
                    }
            }
            std::vector<uint32_t> cell_lo(static_cast<size_t>(G.num_strings), 0xffffffffu), cell_hi(static_cast<size_t>(G.num_strings), 0u);
            for (int str = 0; str < G.num_strings; ++str) {
                const Rect &r = rect[str];
                const bool whole = !r.bad && r.count > 0 && r.count == (r.x1 - r.x0 + 1) * (r.y1 - r.y0 + 1) && r.x1 < 4096 && r.y1 < 4096 && r.sd < 256;
                if (!whole) continue;
                cell_lo[str] = static_cast<uint32_t>(r.x0) | (static_cast<uint32_t>(r.y0) << 12) | (static_cast<uint32_t>(r.sd) << 24);
                cell_hi[str] = static_cast<uint32_t>(r.x1) | (static_cast<uint32_t>(r.y1) << 12);
            }
            for (int str = 0; str < G.num_strings; ++str) {
                const size_t first = first_dom[str], last = first_dom[str + 1];
                const unsigned set = G.str_set[str];
                const unsigned nl = G.set_nlayers[set];
                for (size_t i = first; i < last; ++i) {
                    const uint32_t dom = static_cast<uint32_t>(i - first);
                    int lmin = -1, lmax = -1, count = 0;
                    for (unsigned l = 0; l < nl; ++l)
                        if (G.layer_to_om[static_cast<size_t>(set) * G.max_layers + l] == dom) { if (lmin < 0) lmin = static_cast<int>(l); lmax = static_cast<int>(l); ++count; }
                    const bool ok = !no_named && (cell_lo[str] != 0xffffffffu) && (count > 0) && (lmax - lmin + 1 == count) && (str < 0x10000) && (dom < 0x10000);
                    C.dom_named[4 * i] = ok ? (static_cast<uint32_t>(str) | (dom << 16)) : 0xffffffffu;
                    C.dom_named[4 * i + 1] = ok ? cell_lo[str] : 0u;
                    C.dom_named[4 * i + 2] = ok ? (static_cast<uint32_t>(lmin) | (static_cast<uint32_t>(lmax) << 16)) : 0u;
                    C.dom_named[4 * i + 3] = ok ? cell_hi[str] : 0u;
                }
            }
        }
        C.dom_prox.resize(cells);
        // a cell without a named DOM keeps the bound of its nearest one (255 = nothing within range) in the same place
        for (size_t c = 0; c < cells; ++c)
            C.dom_prox[c] = static_cast<uint32_t>(id1[c]) | (static_cast<uint32_t>(id1[c] == 0xffffu ? q1[c] : q2[c]) << 16);
        name("DOM_PROXIMITY_GRID", {double(nn[0]), double(nn[1]), double(nn[2]), P.dprox_x0, P.dprox_y0, P.dprox_z0, P.dprox_inv_cell, radius});
    }
    scalar("NUM_STRINGS", G.num_strings); scalar("OM_RADIUS", G.om_radius);
    scalar("GEO_STRING_MAX_RADIUS", G.string_max_radius);
    scalar("GEO_LAYER_STRINGSET_NUM", G.num_sets); scalar("GEO_LAYER_STRINGSET_MAX_NUM_LAYERS", G.max_layers);
    scalar("GEO_MAX_DOM_INDEX", G.max_dom_index);
    scalar("GEO_DOM_POS_MAX_ABS_X_MULTIPLIER_IN_TEMPLATE", G.dom_mul_x);
    scalar("GEO_DOM_POS_MAX_ABS_Y_MULTIPLIER_IN_TEMPLATE", G.dom_mul_y);
    name("geoStringPosX", as_doubles(G.str_x)); name("geoStringPosY", as_doubles(G.str_y));
    name("geoStringRadius", as_doubles(G.str_radius));
    name("geoStringMinZ", as_doubles(G.str_minz)); name("geoStringMaxZ", as_doubles(G.str_maxz));
    name("geoStringInStringSet", as_doubles(G.str_set));
    name("geoLayerNum", as_doubles(G.set_nlayers));
    name("geoLayerStartZ", as_doubles(G.set_startz)); name("geoLayerHeight", as_doubles(G.set_height));
    name("geoLayerToOMNumIndexPerStringSet", as_doubles(G.layer_to_om));
    name("geoDomPosTemplatePositionsX_flat", as_doubles(G.dom_tx));
    name("geoDomPosTemplatePositionsY_flat", as_doubles(G.dom_ty));
    name("geoDomPosTemplatePositionsZ_flat", as_doubles(G.dom_tz));
    name("geoDomPosStringStartIndexInTemplateDomList", as_doubles(G.dom_start));
    name("geoDomPosStringMeanPosX", as_doubles(G.dom_meanx)); name("geoDomPosStringMeanPosY", as_doubles(G.dom_meany));
    name("stringIndexToStringID", as_doubles(G.string_index_to_id));
    scalar("PANCAKE_FACTOR", P.pancake);

    scalar("div_ok", P.div_ok);
    {
        std::vector<double> flags;
        for (size_t k = 0; k < G.cells.size(); ++k) flags.push_back(subdet[12 * k + 7]);
        name("div_ok_cells", flags);
    }
    {   // KVariant::fast: the medium's proofs (above the detector section) and every cell width proven
        bool fast = medium_proofs_complete();
        for (size_t k = 0; k < G.cells.size(); ++k)
            if (subdet[12 * k + 7] != 3u) fast = false;
        C.variant.fast = fast;
        scalar("fast_variant", fast ? 1. : 0.);
    }
    P.table_words = static_cast<uint32_t>(img.words.size());
    scalar("lds_image_words", P.table_words);
    scalar("lds_bytes_per_workgroup", static_cast<double>(prop_kernel_lds_bytes(P.table_words)));
    C.lds_image = std::move(img.words);
    // An image beyond the budget of seven workgroups per CU (a detector of several hundred strings) is not refused: the
    // launchers ask the runtime how many workgroups of that size a CU holds and run with fewer (the DOM templates, the
    // largest table, already stay in HBM/L2 then).  Only an image that does not fit a CU's 160 KB even once is.
    if (prop_kernel_lds_bytes(P.table_words) > static_cast<size_t>(159 * 1024))
        throw Error(CLSIMHIP_ERR_CONFIG, "medium / geometry tables (" + std::to_string(prop_kernel_lds_bytes(P.table_words)) +
                                             " bytes) do not fit the 160 KB of LDS of a compute unit");
    return C;
}

} // namespace clsimhip
