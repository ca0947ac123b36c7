// IceTray side of the drop-in: clsim's configuration objects -> the C descriptors of include/clsimhip.h.
//
// Compiled only with CLSIMHIP_WITH_ICETRAY, against the real clsim headers (in this repository's tests: against the
// stand-ins of tests/stubs/, which have the same names).  It is the counterpart of what the reference does with the same
// objects: I3CLSimHelperGenerateMediumPropertiesSource (private/opencl/I3CLSimHelperGenerateMediumPropertiesSource.cxx:
// 207-389) asks each object for OpenCL source text; here each object is asked for its NUMBERS, and a class this
// propagator has no kernel for is refused with log_fatal, like the reference refuses objects without native implementation.
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include <boost/shared_ptr.hpp>
#include <icetray/I3Logging.h>
#include <clsim/I3CLSimMediumProperties.h>
#include <clsim/I3CLSimSimpleGeometry.h>
#include <clsim/function/I3CLSimFunctionConstant.h>
#include <clsim/function/I3CLSimFunctionFromTable.h>
#include <clsim/function/I3CLSimFunctionAbsLenIceCube.h>
#include <clsim/function/I3CLSimFunctionScatLenIceCube.h>
#include <clsim/function/I3CLSimFunctionRefIndexIceCube.h>
#include <clsim/function/I3CLSimScalarFieldConstant.h>
#include <clsim/function/I3CLSimScalarFieldIceTiltZShift.h>
#include <clsim/function/I3CLSimScalarFieldAnisotropyAbsLenScaling.h>
#include <clsim/function/I3CLSimVectorTransformConstant.h>
#include <clsim/function/I3CLSimVectorTransformMatrix.h>
#include <clsim/random_value/I3CLSimRandomValueConstant.h>
#include <clsim/random_value/I3CLSimRandomValueMixed.h>
#include <clsim/random_value/I3CLSimRandomValueHenyeyGreenstein.h>
#include <clsim/random_value/I3CLSimRandomValueSimplifiedLiu.h>
#include <clsim/random_value/I3CLSimRandomValueInterpolatedDistribution.h>
#include <clsim/random_value/I3CLSimRandomValueWlenCherenkovNoDispersion.h>

#include "../../include/clsimhip.h"
#include "private_access.h"

// ---- parameters without a public getter (reference header : line of the member) ----
#ifndef CLSIMHIP_HAVE_PARAMETER_GETTERS
CLSIMHIP_PRIVATE_MEMBER(fconst_value, I3CLSimFunctionConstant, double, value_)                                  // function/I3CLSimFunctionConstant.h:102
CLSIMHIP_PRIVATE_MEMBER(ftable_half, I3CLSimFunctionFromTable, bool, storeDataAsHalfPrecision_)                 // function/I3CLSimFunctionFromTable.h:135
CLSIMHIP_PRIVATE_MEMBER(ref_mode, I3CLSimFunctionRefIndexIceCube, std::string, mode_)                           // function/I3CLSimFunctionRefIndexIceCube.h:126
CLSIMHIP_PRIVATE_MEMBER(ref_n0, I3CLSimFunctionRefIndexIceCube, double, n0_)                                    // :127-136
CLSIMHIP_PRIVATE_MEMBER(ref_n1, I3CLSimFunctionRefIndexIceCube, double, n1_)
CLSIMHIP_PRIVATE_MEMBER(ref_n2, I3CLSimFunctionRefIndexIceCube, double, n2_)
CLSIMHIP_PRIVATE_MEMBER(ref_n3, I3CLSimFunctionRefIndexIceCube, double, n3_)
CLSIMHIP_PRIVATE_MEMBER(ref_n4, I3CLSimFunctionRefIndexIceCube, double, n4_)
CLSIMHIP_PRIVATE_MEMBER(ref_g0, I3CLSimFunctionRefIndexIceCube, double, g0_)
CLSIMHIP_PRIVATE_MEMBER(ref_g1, I3CLSimFunctionRefIndexIceCube, double, g1_)
CLSIMHIP_PRIVATE_MEMBER(ref_g2, I3CLSimFunctionRefIndexIceCube, double, g2_)
CLSIMHIP_PRIVATE_MEMBER(ref_g3, I3CLSimFunctionRefIndexIceCube, double, g3_)
CLSIMHIP_PRIVATE_MEMBER(ref_g4, I3CLSimFunctionRefIndexIceCube, double, g4_)
CLSIMHIP_PRIVATE_MEMBER(sconst_value, I3CLSimScalarFieldConstant, double, value_)                               // function/I3CLSimScalarFieldConstant.h:79
CLSIMHIP_PRIVATE_MEMBER(tilt_distances, I3CLSimScalarFieldIceTiltZShift, std::vector<double>, distancesFromOriginAlongTilt_)   // function/I3CLSimScalarFieldIceTiltZShift.h:82-85
CLSIMHIP_PRIVATE_MEMBER(tilt_z, I3CLSimScalarFieldIceTiltZShift, std::vector<double>, zCoordinates_)
CLSIMHIP_PRIVATE_MEMBER(tilt_corrections, I3CLSimScalarFieldIceTiltZShift, I3Matrix, zCorrections_)
CLSIMHIP_PRIVATE_MEMBER(tilt_azimuth, I3CLSimScalarFieldIceTiltZShift, double, directionOfTiltAzimuth_)
CLSIMHIP_PRIVATE_MEMBER(aniso_azimuth, I3CLSimScalarFieldAnisotropyAbsLenScaling, double, anisotropyDirAzimuth_)   // function/I3CLSimScalarFieldAnisotropyAbsLenScaling.h:87-89
CLSIMHIP_PRIVATE_MEMBER(aniso_along, I3CLSimScalarFieldAnisotropyAbsLenScaling, double, magnitudeAlongDir_)
CLSIMHIP_PRIVATE_MEMBER(aniso_perp, I3CLSimScalarFieldAnisotropyAbsLenScaling, double, magnitudePerpToDir_)
CLSIMHIP_PRIVATE_MEMBER(xform_matrix, I3CLSimVectorTransformMatrix, I3Matrix, matrix_)                          // function/I3CLSimVectorTransformMatrix.h:67-68
CLSIMHIP_PRIVATE_MEMBER(xform_renormalize, I3CLSimVectorTransformMatrix, bool, renormalize_)
CLSIMHIP_PRIVATE_MEMBER(rconst_value, I3CLSimRandomValueConstant, double, value_)                               // random_value/I3CLSimRandomValueConstant.h:72
CLSIMHIP_PRIVATE_MEMBER(hg_mean, I3CLSimRandomValueHenyeyGreenstein, double, meanCosine_)                       // random_value/I3CLSimRandomValueHenyeyGreenstein.h:72
CLSIMHIP_PRIVATE_MEMBER(liu_mean, I3CLSimRandomValueSimplifiedLiu, double, meanCosine_)                         // random_value/I3CLSimRandomValueSimplifiedLiu.h:72
CLSIMHIP_PRIVATE_MEMBER(mixed_fraction, I3CLSimRandomValueMixed, double, fractionOfFirstDistribution_)          // random_value/I3CLSimRandomValueMixed.h:65-67
CLSIMHIP_PRIVATE_MEMBER(mixed_first, I3CLSimRandomValueMixed, I3CLSimRandomValueConstPtr, firstDistribution_)
CLSIMHIP_PRIVATE_MEMBER(mixed_second, I3CLSimRandomValueMixed, I3CLSimRandomValueConstPtr, secondDistribution_)
CLSIMHIP_PRIVATE_MEMBER(interp_x, I3CLSimRandomValueInterpolatedDistribution, std::vector<double>, x_)          // random_value/I3CLSimRandomValueInterpolatedDistribution.h:80-83
CLSIMHIP_PRIVATE_MEMBER(interp_y, I3CLSimRandomValueInterpolatedDistribution, std::vector<double>, y_)
CLSIMHIP_PRIVATE_MEMBER(interp_spacing, I3CLSimRandomValueInterpolatedDistribution, double, constantXSpacing_)
CLSIMHIP_PRIVATE_MEMBER(interp_first, I3CLSimRandomValueInterpolatedDistribution, double, firstX_)
CLSIMHIP_PRIVATE_MEMBER(nodisp_from, I3CLSimRandomValueWlenCherenkovNoDispersion, double, fromWlen_)           // random_value/I3CLSimRandomValueWlenCherenkovNoDispersion.h:69-70
CLSIMHIP_PRIVATE_MEMBER(nodisp_to, I3CLSimRandomValueWlenCherenkovNoDispersion, double, toWlen_)
#define CLSIMHIP_PARAM(obj, tag, getter) ((obj).*member(clsimhip_private::tag()))      /* found by argument-dependent lookup */
#else
#define CLSIMHIP_PARAM(obj, tag, getter) ((obj).getter())
#endif

namespace clsimhip_glue {

// a C function descriptor together with the storage its pointer refers to
struct FunctionHolder {
    clsimhip_function f;
    std::vector<double> values;
    FunctionHolder() { std::memset(&f, 0, sizeof f); }
    FunctionHolder(const FunctionHolder &o) : f(o.f), values(o.values) { f.values = values.empty() ? 0 : values.data(); }
    FunctionHolder &operator=(const FunctionHolder &o) { f = o.f; values = o.values; f.values = values.empty() ? 0 : values.data(); return *this; }
};

// I3CLSimFunctionFromTable (equal spacing) or I3CLSimFunctionConstant -> clsimhip_function (the wavelength bias, the
// tabulated refractive indices)
inline FunctionHolder MakeHIPFunction(const I3CLSimFunction &fn, const char *what)
{
    FunctionHolder h;
    if (const I3CLSimFunctionFromTable *t = dynamic_cast<const I3CLSimFunctionFromTable *>(&fn)) {
        if (!t->GetInEqualSpacingMode()) log_fatal("HIP propagator: %s is a table with unequal wavelength spacing", what);
        h.f.kind = CLSIMHIP_FUNCTION_TABLE;
        h.f.n = static_cast<int32_t>(t->GetNumEntries());
        h.f.start = t->GetFirstWavelength();
        h.f.step = t->GetWavelengthStepping();
        h.values.resize(t->GetNumEntries());
        for (std::size_t i = 0; i < h.values.size(); ++i) h.values[i] = t->GetEntryValue(i);
        h.f.values = h.values.data();
    } else if (const I3CLSimFunctionConstant *c = dynamic_cast<const I3CLSimFunctionConstant *>(&fn)) {
        h.f.kind = CLSIMHIP_FUNCTION_CONSTANT;
        h.f.value = CLSIMHIP_PARAM(*c, fconst_value, GetConstantValue);
    } else {
        log_fatal("HIP propagator: %s is neither an I3CLSimFunctionFromTable nor an I3CLSimFunctionConstant", what);
    }
    return h;
}

struct RandomValueHolder {
    clsimhip_random_value r;
    std::vector<double> y, x;
    RandomValueHolder() { std::memset(&r, 0, sizeof r); }
    RandomValueHolder(const RandomValueHolder &o) : r(o.r), y(o.y), x(o.x) { r.y = y.empty() ? 0 : y.data(); r.x = x.empty() ? 0 : x.data(); }
    RandomValueHolder &operator=(const RandomValueHolder &o) { r = o.r; y = o.y; x = o.x; r.y = y.empty() ? 0 : y.data(); r.x = x.empty() ? 0 : x.data(); return *this; }
};

// wavelength generators (I3CLSimModuleHelper.cxx:73-298 builds them): InterpolatedDistribution with constant spacing or its own x values,
// Constant (delta peak), WlenCherenkovNoDispersion
inline RandomValueHolder MakeHIPWlenGenerator(const I3CLSimRandomValue &rv, std::size_t index)
{
    RandomValueHolder h;
    if (const I3CLSimRandomValueInterpolatedDistribution *d = dynamic_cast<const I3CLSimRandomValueInterpolatedDistribution *>(&rv)) {
        const double spacing = CLSIMHIP_PARAM(*d, interp_spacing, GetConstantXSpacing);
        h.y = CLSIMHIP_PARAM(*d, interp_y, GetYValues);
        h.r.n = static_cast<int32_t>(h.y.size());
        h.r.y = h.y.data();
        if (std::isnan(spacing)) {              // (x, y): the flasher LEDs' measured spectra (InterpolatedDistribution.cxx:40-55)
            h.r.kind = CLSIMHIP_RANDOM_INTERPOLATED_X;
            h.x = CLSIMHIP_PARAM(*d, interp_x, GetXValues);
            h.r.x = h.x.data();
        } else {
            h.r.kind = CLSIMHIP_RANDOM_INTERPOLATED;
            h.r.first = CLSIMHIP_PARAM(*d, interp_first, GetFirstX);
            h.r.spacing = spacing;
        }
    } else if (const I3CLSimRandomValueConstant *c = dynamic_cast<const I3CLSimRandomValueConstant *>(&rv)) {
        h.r.kind = CLSIMHIP_RANDOM_CONSTANT;
        h.r.value = CLSIMHIP_PARAM(*c, rconst_value, GetValue);
    } else if (const I3CLSimRandomValueWlenCherenkovNoDispersion *n = dynamic_cast<const I3CLSimRandomValueWlenCherenkovNoDispersion *>(&rv)) {
        h.r.kind = CLSIMHIP_RANDOM_CHERENKOV_NO_DISPERSION;
        h.r.first = CLSIMHIP_PARAM(*n, nodisp_from, GetFromWlen);
        h.r.spacing = CLSIMHIP_PARAM(*n, nodisp_to, GetToWlen);
    } else {
        log_fatal("HIP propagator: wavelength generator %zu is of a class without a HIP implementation", index);
    }
    return h;
}

// owns a clsimhip_medium
struct MediumHolder {
    clsimhip_medium *m;
    MediumHolder() : m(0) {}
    ~MediumHolder() { if (m) clsimhip_medium_destroy(m); }
private:
    MediumHolder(const MediumHolder &);
    MediumHolder &operator=(const MediumHolder &);
};

namespace detail {
inline void CopyMatrix(const I3CLSimVectorTransformMatrix &t, double out[9], int32_t &renormalize)
{
    const I3Matrix &mat = CLSIMHIP_PARAM(t, xform_matrix, GetMatrix);
    if (mat.size1() != 3 || mat.size2() != 3) log_fatal("HIP propagator: direction transform matrix is not 3x3");
    for (std::size_t i = 0; i < 3; ++i)
        for (std::size_t j = 0; j < 3; ++j) out[3 * i + j] = mat(i, j);
    renormalize = CLSIMHIP_PARAM(t, xform_renormalize, GetRenormalize) ? 1 : 0;
}
}

// I3CLSimMediumProperties -> clsimhip_medium (deep copy; *out owns it).  Covers what the IceCube ice models use
// (python/MakeIceCubeMediumProperties.py, MakeIceCubeMediumPropertiesPhotonics.py) and the homogeneous test medium:
//   absorption / scattering length per layer: AbsLenIceCube + ScatLenIceCube | Constant | FromTable (one binning);
//   phase index, group index override: RefIndexIceCube | FromTable, the same object semantics for every layer;
//   scattering angle: Mixed(SimplifiedLiu, HenyeyGreenstein) | HenyeyGreenstein | SimplifiedLiu;
//   ScalarFieldAnisotropyAbsLenScaling, VectorTransformMatrix, ScalarFieldIceTiltZShift or their constant forms.
inline void MakeHIPMedium(const I3CLSimMediumProperties &m, MediumHolder &out)
{
    if (!m.IsReady()) log_fatal("HIP propagator: medium properties are not ready");
    const uint32_t n = m.GetLayersNum();
    if (n == 0) log_fatal("HIP propagator: medium without layers");
    clsimhip_medium_desc d;
    std::memset(&d, 0, sizeof d);
    d.num_layers = static_cast<int32_t>(n);
    d.layers_z_start = m.GetLayersZStart();
    d.layers_height = m.GetLayersHeight();
    d.min_wavelength = m.GetMinWavelength();
    d.max_wavelength = m.GetMaxWavelength();

    // ---- lengths ----
    std::vector<double> a(n), b(n), c(n), abs_table, sca_table;
    const I3CLSimFunctionConstPtr abs0 = m.GetAbsorptionLength(0), sca0 = m.GetScatteringLength(0);
    if (boost::dynamic_pointer_cast<const I3CLSimFunctionAbsLenIceCube>(abs0)) {
        d.lengths_kind = CLSIMHIP_LENGTHS_ICECUBE;
        for (uint32_t i = 0; i < n; ++i) {
            const boost::shared_ptr<const I3CLSimFunctionAbsLenIceCube> al = boost::dynamic_pointer_cast<const I3CLSimFunctionAbsLenIceCube>(m.GetAbsorptionLength(i));
            const boost::shared_ptr<const I3CLSimFunctionScatLenIceCube> sl = boost::dynamic_pointer_cast<const I3CLSimFunctionScatLenIceCube>(m.GetScatteringLength(i));
            if (!al || !sl) log_fatal("HIP propagator: layer %u does not use I3CLSimFunctionAbsLenIceCube / ScatLenIceCube like layer 0", i);
            if (i == 0) {
                d.kappa = al->GetKappa(); d.A = al->GetA(); d.B = al->GetB(); d.D = al->GetD(); d.E = al->GetE(); d.alpha = sl->GetAlpha();
            } else if (al->GetKappa() != d.kappa || al->GetA() != d.A || al->GetB() != d.B || al->GetD() != d.D || al->GetE() != d.E || sl->GetAlpha() != d.alpha) {
                // the reference's optimised generator requires the same (_Optimizers.cxx:123-135, 195-207)
                log_fatal("HIP propagator: layer %u has other wavelength parameters (kappa, A, B, D, E, alpha) than layer 0", i);
            }
            a[i] = al->GetADust400(); b[i] = al->GetDeltaTau(); c[i] = sl->GetB400();
        }
        d.a_dust400 = a.data(); d.delta_tau = b.data(); d.b400 = c.data();
    } else if (boost::dynamic_pointer_cast<const I3CLSimFunctionConstant>(abs0)) {
        d.lengths_kind = CLSIMHIP_LENGTHS_CONSTANT;
        for (uint32_t i = 0; i < n; ++i) {
            const boost::shared_ptr<const I3CLSimFunctionConstant> al = boost::dynamic_pointer_cast<const I3CLSimFunctionConstant>(m.GetAbsorptionLength(i));
            const boost::shared_ptr<const I3CLSimFunctionConstant> sl = boost::dynamic_pointer_cast<const I3CLSimFunctionConstant>(m.GetScatteringLength(i));
            if (!al || !sl) log_fatal("HIP propagator: layer %u does not use I3CLSimFunctionConstant lengths like layer 0", i);
            a[i] = CLSIMHIP_PARAM(*al, fconst_value, GetConstantValue);
            b[i] = CLSIMHIP_PARAM(*sl, fconst_value, GetConstantValue);
        }
        d.abs_length = a.data(); d.sca_length = b.data();
    } else if (const boost::shared_ptr<const I3CLSimFunctionFromTable> t0 = boost::dynamic_pointer_cast<const I3CLSimFunctionFromTable>(abs0)) {
        d.lengths_kind = CLSIMHIP_LENGTHS_TABLE;
        if (!t0->GetInEqualSpacingMode()) log_fatal("HIP propagator: tabulated lengths need equal wavelength spacing");
        const std::size_t nw = t0->GetNumEntries();
        const bool half = CLSIMHIP_PARAM(*t0, ftable_half, GetStoreDataAsHalfPrecision);
        d.table_num_wavelengths = static_cast<int32_t>(nw);
        d.table_start_wavelength = t0->GetFirstWavelength();
        d.table_wavelength_step = t0->GetWavelengthStepping();
        d.table_store_as_16bit = half ? 1 : 0;
        abs_table.resize(static_cast<std::size_t>(n) * nw);
        sca_table.resize(static_cast<std::size_t>(n) * nw);
        for (uint32_t i = 0; i < n; ++i) {
            const boost::shared_ptr<const I3CLSimFunctionFromTable> al = boost::dynamic_pointer_cast<const I3CLSimFunctionFromTable>(m.GetAbsorptionLength(i));
            const boost::shared_ptr<const I3CLSimFunctionFromTable> sl = boost::dynamic_pointer_cast<const I3CLSimFunctionFromTable>(m.GetScatteringLength(i));
            if (!al || !sl) log_fatal("HIP propagator: layer %u does not use I3CLSimFunctionFromTable lengths like layer 0", i);
            const I3CLSimFunctionFromTable *both[2] = {al.get(), sl.get()};
            for (int k = 0; k < 2; ++k) {
                const I3CLSimFunctionFromTable &t = *both[k];
                if (!t.GetInEqualSpacingMode() || t.GetNumEntries() != nw || t.GetFirstWavelength() != d.table_start_wavelength ||
                    t.GetWavelengthStepping() != d.table_wavelength_step || CLSIMHIP_PARAM(t, ftable_half, GetStoreDataAsHalfPrecision) != half)
                    log_fatal("HIP propagator: the length tables of layer %u are binned or stored differently from layer 0's", i);
                std::vector<double> &dst = k ? sca_table : abs_table;
                for (std::size_t w = 0; w < nw; ++w) dst[static_cast<std::size_t>(i) * nw + w] = t.GetEntryValue(w);
            }
        }
        d.abs_length_table = abs_table.data(); d.sca_length_table = sca_table.data();
    } else {
        log_fatal("HIP propagator: the absorption length of layer 0 is of a class without a HIP implementation");
    }

    // ---- refractive indices: one function for all layers (propagation_kernel.c.cl:525-527 needs a layer
    // independent group velocity; the reference finds that out by comparing the generated functions) ----
    for (uint32_t i = 1; i < n; ++i) {
        if (!m.GetPhaseRefractiveIndex(i)->CompareTo(*m.GetPhaseRefractiveIndex(0))) log_fatal("HIP propagator: the phase refractive index depends on the layer");
        const I3CLSimFunctionConstPtr g0 = m.GetGroupRefractiveIndexOverride(0), gi = m.GetGroupRefractiveIndexOverride(i);
        if ((!g0) != (!gi) || (g0 && !gi->CompareTo(*g0))) log_fatal("HIP propagator: the group refractive index depends on the layer");
    }
    const I3CLSimFunctionConstPtr phase = m.GetPhaseRefractiveIndex(0), group = m.GetGroupRefractiveIndexOverride(0);
    FunctionHolder phase_table, group_table;
    const boost::shared_ptr<const I3CLSimFunctionRefIndexIceCube> pr = boost::dynamic_pointer_cast<const I3CLSimFunctionRefIndexIceCube>(phase);
    const boost::shared_ptr<const I3CLSimFunctionRefIndexIceCube> gr = boost::dynamic_pointer_cast<const I3CLSimFunctionRefIndexIceCube>(group);
    if (pr) {
        if (CLSIMHIP_PARAM(*pr, ref_mode, GetMode) != "phase") log_fatal("HIP propagator: the phase refractive index object is not in mode \"phase\"");
        d.phase_index_kind = CLSIMHIP_REFINDEX_ICECUBE;
        d.n[0] = CLSIMHIP_PARAM(*pr, ref_n0, GetN0); d.n[1] = CLSIMHIP_PARAM(*pr, ref_n1, GetN1); d.n[2] = CLSIMHIP_PARAM(*pr, ref_n2, GetN2);
        d.n[3] = CLSIMHIP_PARAM(*pr, ref_n3, GetN3); d.n[4] = CLSIMHIP_PARAM(*pr, ref_n4, GetN4);
    } else {
        d.phase_index_kind = CLSIMHIP_REFINDEX_TABLE;
        phase_table = MakeHIPFunction(*phase, "the phase refractive index");
        if (phase_table.f.kind != CLSIMHIP_FUNCTION_TABLE) log_fatal("HIP propagator: a constant phase refractive index is not implemented");
        d.phase_index_table = phase_table.f;
    }
    if (!group) {
        // no override: group velocity from the phase index's dispersion (I3CLSimHelperGenerateMediumPropertiesSource.cxx:274-300)
        if (!pr) log_fatal("HIP propagator: no group refractive index override is set and the phase refractive index has no derivative");
        d.group_index_kind = CLSIMHIP_REFINDEX_DISPERSION;
    } else if (gr) {
        if (CLSIMHIP_PARAM(*gr, ref_mode, GetMode) != "group") log_fatal("HIP propagator: the group refractive index object is not in mode \"group\"");
        if (!pr) log_fatal("HIP propagator: I3CLSimFunctionRefIndexIceCube group index with a tabulated phase index is not implemented");
        // the group index of that class is n_phase(lambda) * correction(lambda) with its own copy of n0..n4 (RefIndexIceCube.cxx:158-163)
        const double gn[5] = {CLSIMHIP_PARAM(*gr, ref_n0, GetN0), CLSIMHIP_PARAM(*gr, ref_n1, GetN1), CLSIMHIP_PARAM(*gr, ref_n2, GetN2),
                              CLSIMHIP_PARAM(*gr, ref_n3, GetN3), CLSIMHIP_PARAM(*gr, ref_n4, GetN4)};
        for (int k = 0; k < 5; ++k)
            if (gn[k] != d.n[k]) log_fatal("HIP propagator: the group index object's phase coefficients differ from the phase index object's");
        d.group_index_kind = CLSIMHIP_REFINDEX_ICECUBE;
        d.g[0] = CLSIMHIP_PARAM(*gr, ref_g0, GetG0); d.g[1] = CLSIMHIP_PARAM(*gr, ref_g1, GetG1); d.g[2] = CLSIMHIP_PARAM(*gr, ref_g2, GetG2);
        d.g[3] = CLSIMHIP_PARAM(*gr, ref_g3, GetG3); d.g[4] = CLSIMHIP_PARAM(*gr, ref_g4, GetG4);
    } else {
        d.group_index_kind = CLSIMHIP_REFINDEX_TABLE;
        group_table = MakeHIPFunction(*group, "the group refractive index override");
        if (group_table.f.kind != CLSIMHIP_FUNCTION_TABLE) log_fatal("HIP propagator: a constant group refractive index is not implemented");
        d.group_index_table = group_table.f;
    }

    // ---- scattering angle ----
    const I3CLSimRandomValueConstPtr scat = m.GetScatteringCosAngleDistribution();
    if (const boost::shared_ptr<const I3CLSimRandomValueMixed> mix = boost::dynamic_pointer_cast<const I3CLSimRandomValueMixed>(scat)) {
        const boost::shared_ptr<const I3CLSimRandomValueSimplifiedLiu> liu =
            boost::dynamic_pointer_cast<const I3CLSimRandomValueSimplifiedLiu>(CLSIMHIP_PARAM(*mix, mixed_first, GetFirstDistribution));
        const boost::shared_ptr<const I3CLSimRandomValueHenyeyGreenstein> hg =
            boost::dynamic_pointer_cast<const I3CLSimRandomValueHenyeyGreenstein>(CLSIMHIP_PARAM(*mix, mixed_second, GetSecondDistribution));
        if (!liu || !hg) log_fatal("HIP propagator: the mixed scattering distribution is not Mixed(SimplifiedLiu, HenyeyGreenstein)");
        d.scatter_kind = CLSIMHIP_SCATTER_MIXED;
        d.liu_fraction = CLSIMHIP_PARAM(*mix, mixed_fraction, GetFractionOfFirstDistribution);
        d.mean_cosine = CLSIMHIP_PARAM(*hg, hg_mean, GetMeanCosine);
        if (CLSIMHIP_PARAM(*liu, liu_mean, GetMeanCosine) != d.mean_cosine)
            log_fatal("HIP propagator: SimplifiedLiu and HenyeyGreenstein with different mean cosines are not implemented");
    } else if (const boost::shared_ptr<const I3CLSimRandomValueHenyeyGreenstein> hg = boost::dynamic_pointer_cast<const I3CLSimRandomValueHenyeyGreenstein>(scat)) {
        d.scatter_kind = CLSIMHIP_SCATTER_HG;
        d.mean_cosine = CLSIMHIP_PARAM(*hg, hg_mean, GetMeanCosine);
    } else if (const boost::shared_ptr<const I3CLSimRandomValueSimplifiedLiu> liu = boost::dynamic_pointer_cast<const I3CLSimRandomValueSimplifiedLiu>(scat)) {
        d.scatter_kind = CLSIMHIP_SCATTER_LIU;
        d.mean_cosine = CLSIMHIP_PARAM(*liu, liu_mean, GetMeanCosine);
    } else {
        log_fatal("HIP propagator: the scattering angle distribution is of a class without a HIP implementation");
    }

    // ---- anisotropy (python/util/GetSpiceLeaAnisotropyTransforms.py builds these for SPICE-Lea) ----
    if (const I3CLSimScalarFieldConstPtr corr = m.GetDirectionalAbsorptionLengthCorrection()) {
        if (const boost::shared_ptr<const I3CLSimScalarFieldAnisotropyAbsLenScaling> an = boost::dynamic_pointer_cast<const I3CLSimScalarFieldAnisotropyAbsLenScaling>(corr)) {
            d.has_anisotropy = 1;
            d.aniso_azimuth = CLSIMHIP_PARAM(*an, aniso_azimuth, GetAnisotropyDirAzimuth);
            d.aniso_k1 = CLSIMHIP_PARAM(*an, aniso_along, GetMagnitudeAlongDir);
            d.aniso_k2 = CLSIMHIP_PARAM(*an, aniso_perp, GetMagnitudePerpToDir);
        } else if (const boost::shared_ptr<const I3CLSimScalarFieldConstant> k = boost::dynamic_pointer_cast<const I3CLSimScalarFieldConstant>(corr)) {
            if (CLSIMHIP_PARAM(*k, sconst_value, GetConstantValue) != 1.) log_fatal("HIP propagator: a constant absorption length correction other than 1 is not implemented");
        } else {
            log_fatal("HIP propagator: the directional absorption length correction is of a class without a HIP implementation");
        }
    }
    const I3CLSimVectorTransformConstPtr xf[2] = {m.GetPreScatterDirectionTransform(), m.GetPostScatterDirectionTransform()};
    for (int k = 0; k < 2; ++k) {
        if (!xf[k] || boost::dynamic_pointer_cast<const I3CLSimVectorTransformConstant>(xf[k])) continue;
        const boost::shared_ptr<const I3CLSimVectorTransformMatrix> mt = boost::dynamic_pointer_cast<const I3CLSimVectorTransformMatrix>(xf[k]);
        if (!mt) log_fatal("HIP propagator: the %s-scatter direction transform is of a class without a HIP implementation", k ? "post" : "pre");
        if (k == 0) { d.has_pre_transform = 1; detail::CopyMatrix(*mt, d.pre_matrix, d.pre_renormalize); }
        else { d.has_post_transform = 1; detail::CopyMatrix(*mt, d.post_matrix, d.post_renormalize); }
    }

    // ---- ice tilt ----
    std::vector<double> tilt_corr;
    if (const I3CLSimScalarFieldConstPtr tilt = m.GetIceTiltZShift()) {
        if (const boost::shared_ptr<const I3CLSimScalarFieldIceTiltZShift> tz = boost::dynamic_pointer_cast<const I3CLSimScalarFieldIceTiltZShift>(tilt)) {
            const std::vector<double> &dist = CLSIMHIP_PARAM(*tz, tilt_distances, GetDistancesFromOriginAlongTilt);
            const std::vector<double> &zc = CLSIMHIP_PARAM(*tz, tilt_z, GetZCoordinates);
            const I3Matrix &corr = CLSIMHIP_PARAM(*tz, tilt_corrections, GetZCorrections);
            if (corr.size1() != dist.size() || corr.size2() != zc.size()) log_fatal("HIP propagator: inconsistent ice tilt tables");
            d.has_tilt = 1;
            d.tilt_num_distances = static_cast<int32_t>(dist.size());
            d.tilt_num_z = static_cast<int32_t>(zc.size());
            d.tilt_distances = dist.data();
            d.tilt_z_coordinates = zc.data();
            tilt_corr.resize(dist.size() * zc.size());
            for (std::size_t i = 0; i < dist.size(); ++i)
                for (std::size_t j = 0; j < zc.size(); ++j) tilt_corr[i * zc.size() + j] = corr(i, j);      // [distance][z], ScalarFieldIceTiltZShift.cxx:168-172
            d.tilt_z_corrections = tilt_corr.data();
            d.tilt_azimuth = CLSIMHIP_PARAM(*tz, tilt_azimuth, GetDirectionOfTiltAzimuth);
        } else if (const boost::shared_ptr<const I3CLSimScalarFieldConstant> k = boost::dynamic_pointer_cast<const I3CLSimScalarFieldConstant>(tilt)) {
            if (CLSIMHIP_PARAM(*k, sconst_value, GetConstantValue) != 0.) log_fatal("HIP propagator: a constant ice tilt shift other than 0 is not implemented");
        } else {
            log_fatal("HIP propagator: the ice tilt is of a class without a HIP implementation");
        }
    }

    if (out.m) { clsimhip_medium_destroy(out.m); out.m = 0; }
    if (clsimhip_medium_create(&d, &out.m) != CLSIMHIP_OK) log_fatal("HIP propagator: %s", clsimhip_last_error(0));
}

} // namespace clsimhip_glue
