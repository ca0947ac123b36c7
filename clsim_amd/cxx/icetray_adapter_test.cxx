// The IceTray build of the adapter (-DCLSIMHIP_WITH_ICETRAY), exercised through the REAL interface signatures
// (public/clsim/I3CLSimStepToPhotonConverter.h:91-124): clsim configuration objects in, I3CLSimPhotonSeries out.
// Compiled by tests/test_icetray_adapter.py against the stand-in headers of tests/stubs/ (IceTray is not in this image).
//
//   icetray_adapter_test check <ppc ice directory> [<photonics table file>]
//       no GPU: builds I3CLSimMediumProperties objects the way python/MakeIceCubeMediumProperties.py does, sends them
//       through the glue and compares the C description that arrives with the one they were built from; refusal of
//       classes without a HIP kernel; the configuration sequence of I3CLSimModuleHelper::initializeOpenCL up to Compile()
//   icetray_adapter_test run <ppc ice directory> <geometry file> <steps file> <photons file> <workitems>
//       GPU: I3CLSimModuleHelper::initializeHIP(...) (= initializeOpenCL, ModuleHelper.cxx:303-372), one bunch through
//       EnqueueSteps / GetConversionResult, the photons written as 80-byte records for the Python test to compare with the
//       ctypes path on the same steps and streams
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>

#include <clsim/I3CLSimSimpleGeometryUserConfigurable.h>
#include <icetray/I3Units.h>
#include "I3CLSimStepToPhotonConverterHIP.h"
#include "I3CLSimLightSourceToStepConverterHIP.h"

namespace {

// deterministic I3RandomService: splitmix64, Integer(imax) = high word % imax (tests/test_icetray_adapter.py has the same)
class TestRandomService : public I3RandomService {
public:
    explicit TestRandomService(uint64_t seed) : s_(seed) {}
    unsigned int Integer(unsigned int imax) override
    {
        s_ += 0x9e3779b97f4a7c15ull;
        uint64_t z = s_;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        return static_cast<unsigned int>((z >> 32) % imax);
    }
private:
    uint64_t s_;
};

#define REQUIRE(cond)                                                                        \
    do {                                                                                     \
        if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

// python/MakeIceCubeMediumProperties.py:49-256 in C++: the objects that loader creates, from the numbers of `d`
I3CLSimMediumPropertiesPtr MediumFromDescription(const clsimhip_medium_desc &d)
{
    I3CLSimMediumPropertiesPtr m(new I3CLSimMediumProperties(0.9216 * I3Units::g / I3Units::cm3, static_cast<uint32_t>(d.num_layers), d.layers_z_start, d.layers_height, -870., 1948.07));
    I3CLSimFunctionConstPtr phase, group;
    if (d.phase_index_kind == CLSIMHIP_REFINDEX_ICECUBE) {
        phase.reset(new I3CLSimFunctionRefIndexIceCube("phase", d.n[0], d.n[1], d.n[2], d.n[3], d.n[4], d.g[0], d.g[1], d.g[2], d.g[3], d.g[4]));
        group.reset(new I3CLSimFunctionRefIndexIceCube("group", d.n[0], d.n[1], d.n[2], d.n[3], d.n[4], d.g[0], d.g[1], d.g[2], d.g[3], d.g[4]));
    } else {
        phase.reset(new I3CLSimFunctionFromTable(d.phase_index_table.start, d.phase_index_table.step,
                                                 std::vector<double>(d.phase_index_table.values, d.phase_index_table.values + d.phase_index_table.n)));
        group.reset(new I3CLSimFunctionFromTable(d.group_index_table.start, d.group_index_table.step,
                                                 std::vector<double>(d.group_index_table.values, d.group_index_table.values + d.group_index_table.n)));
    }
    for (int i = 0; i < d.num_layers; ++i) {
        if (d.lengths_kind == CLSIMHIP_LENGTHS_ICECUBE) {
            m->SetAbsorptionLength(i, I3CLSimFunctionConstPtr(new I3CLSimFunctionAbsLenIceCube(d.kappa, d.A, d.B, d.D, d.E, d.a_dust400[i], d.delta_tau[i])));
            m->SetScatteringLength(i, I3CLSimFunctionConstPtr(new I3CLSimFunctionScatLenIceCube(d.alpha, d.b400[i])));
        } else if (d.lengths_kind == CLSIMHIP_LENGTHS_CONSTANT) {
            m->SetAbsorptionLength(i, I3CLSimFunctionConstPtr(new I3CLSimFunctionConstant(d.abs_length[i])));
            m->SetScatteringLength(i, I3CLSimFunctionConstPtr(new I3CLSimFunctionConstant(d.sca_length[i])));
        } else {
            const int nw = d.table_num_wavelengths;
            m->SetAbsorptionLength(i, I3CLSimFunctionConstPtr(new I3CLSimFunctionFromTable(d.table_start_wavelength, d.table_wavelength_step,
                std::vector<double>(d.abs_length_table + i * nw, d.abs_length_table + (i + 1) * nw), d.table_store_as_16bit != 0)));
            m->SetScatteringLength(i, I3CLSimFunctionConstPtr(new I3CLSimFunctionFromTable(d.table_start_wavelength, d.table_wavelength_step,
                std::vector<double>(d.sca_length_table + i * nw, d.sca_length_table + (i + 1) * nw), d.table_store_as_16bit != 0)));
        }
        m->SetPhaseRefractiveIndex(i, phase);
        if (d.group_index_kind != CLSIMHIP_REFINDEX_DISPERSION) m->SetGroupRefractiveIndexOverride(i, group);
    }
    if (d.scatter_kind == CLSIMHIP_SCATTER_MIXED)
        m->SetScatteringCosAngleDistribution(I3CLSimRandomValueConstPtr(new I3CLSimRandomValueMixed(
            d.liu_fraction, I3CLSimRandomValueConstPtr(new I3CLSimRandomValueSimplifiedLiu(d.mean_cosine)),
            I3CLSimRandomValueConstPtr(new I3CLSimRandomValueHenyeyGreenstein(d.mean_cosine)))));
    else if (d.scatter_kind == CLSIMHIP_SCATTER_HG)
        m->SetScatteringCosAngleDistribution(I3CLSimRandomValueConstPtr(new I3CLSimRandomValueHenyeyGreenstein(d.mean_cosine)));
    else
        m->SetScatteringCosAngleDistribution(I3CLSimRandomValueConstPtr(new I3CLSimRandomValueSimplifiedLiu(d.mean_cosine)));
    if (d.has_anisotropy)
        m->SetDirectionalAbsorptionLengthCorrection(I3CLSimScalarFieldConstPtr(new I3CLSimScalarFieldAnisotropyAbsLenScaling(d.aniso_azimuth, d.aniso_k1, d.aniso_k2)));
    else
        m->SetDirectionalAbsorptionLengthCorrection(I3CLSimScalarFieldConstPtr(new I3CLSimScalarFieldConstant(1.)));
    for (int k = 0; k < 2; ++k) {
        const bool has = k ? d.has_post_transform : d.has_pre_transform;
        I3CLSimVectorTransformConstPtr t;
        if (has) {
            I3Matrix mat(3, 3);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) mat(i, j) = (k ? d.post_matrix : d.pre_matrix)[3 * i + j];
            t.reset(new I3CLSimVectorTransformMatrix(mat, (k ? d.post_renormalize : d.pre_renormalize) != 0));
        } else {
            t.reset(new I3CLSimVectorTransformConstant());
        }
        if (k) m->SetPostScatterDirectionTransform(t); else m->SetPreScatterDirectionTransform(t);
    }
    if (d.has_tilt) {
        I3Matrix corr(d.tilt_num_distances, d.tilt_num_z);
        for (int i = 0; i < d.tilt_num_distances; ++i) for (int j = 0; j < d.tilt_num_z; ++j) corr(i, j) = d.tilt_z_corrections[i * d.tilt_num_z + j];
        m->SetIceTiltZShift(I3CLSimScalarFieldConstPtr(new I3CLSimScalarFieldIceTiltZShift(
            std::vector<double>(d.tilt_distances, d.tilt_distances + d.tilt_num_distances),
            std::vector<double>(d.tilt_z_coordinates, d.tilt_z_coordinates + d.tilt_num_z), corr, d.tilt_azimuth)));
    } else {
        m->SetIceTiltZShift(I3CLSimScalarFieldConstPtr(new I3CLSimScalarFieldConstant(0.)));
    }
    m->SetForcedMinWlen(d.min_wavelength);
    m->SetForcedMaxWlen(d.max_wavelength);
    return m;
}

bool SameArray(const double *a, const double *b, size_t n) { return (n == 0) || (a && b && std::memcmp(a, b, n * sizeof(double)) == 0); }

// every field of the description that arrived equals the one the objects were built from
int CompareDescriptions(const clsimhip_medium_desc &a, const clsimhip_medium_desc &b)
{
    REQUIRE(a.num_layers == b.num_layers && a.layers_z_start == b.layers_z_start && a.layers_height == b.layers_height);
    REQUIRE(a.min_wavelength == b.min_wavelength && a.max_wavelength == b.max_wavelength);
    REQUIRE(a.lengths_kind == b.lengths_kind);
    const size_t n = static_cast<size_t>(a.num_layers);
    if (a.lengths_kind == CLSIMHIP_LENGTHS_ICECUBE) {
        REQUIRE(a.alpha == b.alpha && a.kappa == b.kappa && a.A == b.A && a.B == b.B && a.D == b.D && a.E == b.E);
        REQUIRE(SameArray(a.a_dust400, b.a_dust400, n) && SameArray(a.delta_tau, b.delta_tau, n) && SameArray(a.b400, b.b400, n));
    } else if (a.lengths_kind == CLSIMHIP_LENGTHS_CONSTANT) {
        REQUIRE(SameArray(a.abs_length, b.abs_length, n) && SameArray(a.sca_length, b.sca_length, n));
    } else {
        REQUIRE(a.table_num_wavelengths == b.table_num_wavelengths && a.table_start_wavelength == b.table_start_wavelength &&
                a.table_wavelength_step == b.table_wavelength_step && a.table_store_as_16bit == b.table_store_as_16bit);
        REQUIRE(SameArray(a.abs_length_table, b.abs_length_table, n * a.table_num_wavelengths));
        REQUIRE(SameArray(a.sca_length_table, b.sca_length_table, n * a.table_num_wavelengths));
    }
    REQUIRE(a.phase_index_kind == b.phase_index_kind && a.group_index_kind == b.group_index_kind);
    if (a.phase_index_kind == CLSIMHIP_REFINDEX_ICECUBE) REQUIRE(std::memcmp(a.n, b.n, sizeof a.n) == 0);
    else REQUIRE(a.phase_index_table.n == b.phase_index_table.n && a.phase_index_table.start == b.phase_index_table.start &&
                 a.phase_index_table.step == b.phase_index_table.step && SameArray(a.phase_index_table.values, b.phase_index_table.values, a.phase_index_table.n));
    if (a.group_index_kind == CLSIMHIP_REFINDEX_ICECUBE) REQUIRE(std::memcmp(a.g, b.g, sizeof a.g) == 0);
    else REQUIRE(a.group_index_table.n == b.group_index_table.n && a.group_index_table.start == b.group_index_table.start &&
                 a.group_index_table.step == b.group_index_table.step && SameArray(a.group_index_table.values, b.group_index_table.values, a.group_index_table.n));
    REQUIRE(a.scatter_kind == b.scatter_kind && a.mean_cosine == b.mean_cosine);
    if (a.scatter_kind == CLSIMHIP_SCATTER_MIXED) REQUIRE(a.liu_fraction == b.liu_fraction);
    REQUIRE(a.has_anisotropy == b.has_anisotropy);
    if (a.has_anisotropy) REQUIRE(a.aniso_azimuth == b.aniso_azimuth && a.aniso_k1 == b.aniso_k1 && a.aniso_k2 == b.aniso_k2);
    REQUIRE(a.has_pre_transform == b.has_pre_transform && a.has_post_transform == b.has_post_transform);
    if (a.has_pre_transform) REQUIRE(a.pre_renormalize == b.pre_renormalize && std::memcmp(a.pre_matrix, b.pre_matrix, sizeof a.pre_matrix) == 0);
    if (a.has_post_transform) REQUIRE(a.post_renormalize == b.post_renormalize && std::memcmp(a.post_matrix, b.post_matrix, sizeof a.post_matrix) == 0);
    REQUIRE(a.has_tilt == b.has_tilt);
    if (a.has_tilt) {
        REQUIRE(a.tilt_num_distances == b.tilt_num_distances && a.tilt_num_z == b.tilt_num_z && a.tilt_azimuth == b.tilt_azimuth);
        REQUIRE(SameArray(a.tilt_distances, b.tilt_distances, a.tilt_num_distances) && SameArray(a.tilt_z_coordinates, b.tilt_z_coordinates, a.tilt_num_z));
        REQUIRE(SameArray(a.tilt_z_corrections, b.tilt_z_corrections, static_cast<size_t>(a.tilt_num_distances) * a.tilt_num_z));
    }
    return 0;
}

int RoundTrip(clsimhip_medium *source, const char *label)
{
    clsimhip_medium_desc before, after;
    REQUIRE(clsimhip_medium_describe(source, &before) == CLSIMHIP_OK);
    const I3CLSimMediumPropertiesPtr objects = MediumFromDescription(before);
    clsimhip_glue::MediumHolder through;
    clsimhip_glue::MakeHIPMedium(*objects, through);
    REQUIRE(through.m != nullptr);
    REQUIRE(clsimhip_medium_describe(through.m, &after) == CLSIMHIP_OK);
    if (CompareDescriptions(before, after) != 0) { std::printf("medium round trip differs: %s\n", label); return 1; }
    std::printf("medium round trip ok: %s (%d layers, lengths kind %d, tilt %d, anisotropy %d)\n", label, before.num_layers, before.lengths_kind,
                before.has_tilt, before.has_anisotropy);
    return 0;
}

template <class F>
bool Fatal(F &&f, const char *text)
{
    try { f(); } catch (const std::runtime_error &e) {
        if (std::strstr(e.what(), text)) return true;
        std::printf("unexpected message: %s\n", e.what());
        return false;
    }
    std::printf("no log_fatal / exception containing '%s'\n", text);
    return false;
}

struct Inputs {
    I3CLSimFunctionConstPtr bias;
    std::vector<I3CLSimRandomValueConstPtr> generators;
};

// I3CLSimModuleHelper::makeCherenkovWavelengthGenerator + python/GetIceCubeDOMAcceptance.py through the library's helpers
int MakeSpectra(clsimhip_medium *medium, Inputs &in)
{
    std::vector<double> acc(43), y(43);
    double start = 0, step = 0, first = 0, spacing = 0;
    REQUIRE(clsimhip_icecube_dom_acceptance(0.16510, 1.0, acc.data(), &start, &step) == CLSIMHIP_OK);
    const clsimhip_function bias = {CLSIMHIP_FUNCTION_TABLE, 43, start, step, acc.data(), 0., nullptr};
    REQUIRE(clsimhip_make_cherenkov_wlen_generator(&bias, medium, y.data(), &first, &spacing) == CLSIMHIP_OK);
    in.bias.reset(new I3CLSimFunctionFromTable(start, step, acc));
    in.generators.assign(1, I3CLSimRandomValueConstPtr(new I3CLSimRandomValueInterpolatedDistribution(first, spacing, y)));
    return 0;
}

I3CLSimSimpleGeometryUserConfigurablePtr SingleString()
{
    I3CLSimSimpleGeometryUserConfigurablePtr g(new I3CLSimSimpleGeometryUserConfigurable(0.16510 * 5., 60));
    for (int k = 0; k < 60; ++k) {
        g->SetStringID(k, 1); g->SetDomID(k, k + 1); g->SetPosX(k, 20.); g->SetPosY(k, 20.); g->SetPosZ(k, 500. - 17. * k); g->SetSubdetector(k, "IceCube");
    }
    return g;
}

int Check(const char *ice_dir, const char *photonics_file)
{
    clsimhip_medium *ppc = nullptr;
    REQUIRE(clsimhip_medium_create_from_ppc(ice_dir, 1948.07, 1, &ppc) == CLSIMHIP_OK);
    if (RoundTrip(ppc, ice_dir) != 0) return 1;
    if (photonics_file) {
        clsimhip_medium *tab = nullptr;
        REQUIRE(clsimhip_medium_create_from_photonics(photonics_file, 1948.07, &tab) == CLSIMHIP_OK);
        if (RoundTrip(tab, photonics_file) != 0) return 1;
        {   // tabulated refractive indices without an override: FromTable has no derivative (I3CLSimFunctionFromTable.h:67)
            clsimhip_medium_desc td;
            REQUIRE(clsimhip_medium_describe(tab, &td) == CLSIMHIP_OK);
            const I3CLSimMediumPropertiesPtr m = MediumFromDescription(td);
            for (uint32_t i = 0; i < m->GetLayersNum(); ++i) m->SetGroupRefractiveIndexOverride(i, I3CLSimFunctionConstPtr());
            clsimhip_glue::MediumHolder h;
            REQUIRE(Fatal([&] { clsimhip_glue::MakeHIPMedium(*m, h); }, "has no derivative"));
        }
        clsimhip_medium_destroy(tab);
    }
    {   // homogeneous medium with constant lengths (BASELINE config C1)
        clsimhip_medium_desc d;
        REQUIRE(clsimhip_medium_describe(ppc, &d) == CLSIMHIP_OK);
        const double absLen = 100., scaLen = 25.;
        d.num_layers = 1; d.layers_z_start = -1000.; d.layers_height = 2000.;
        d.lengths_kind = CLSIMHIP_LENGTHS_CONSTANT; d.abs_length = &absLen; d.sca_length = &scaLen;
        d.has_tilt = 0; d.has_anisotropy = 0; d.has_pre_transform = 0; d.has_post_transform = 0;
        clsimhip_medium *c1 = nullptr;
        REQUIRE(clsimhip_medium_create(&d, &c1) == CLSIMHIP_OK);
        if (RoundTrip(c1, "homogeneous") != 0) return 1;
        clsimhip_medium_destroy(c1);
    }

    // classes without a HIP kernel are refused, not approximated
    clsimhip_medium_desc d;
    REQUIRE(clsimhip_medium_describe(ppc, &d) == CLSIMHIP_OK);
    {
        const I3CLSimMediumPropertiesPtr m = MediumFromDescription(d);
        m->SetAbsorptionLength(3, I3CLSimFunctionConstPtr(new I3CLSimFunctionConstant(50.)));
        clsimhip_glue::MediumHolder h;
        REQUIRE(Fatal([&] { clsimhip_glue::MakeHIPMedium(*m, h); }, "layer 3 does not use I3CLSimFunctionAbsLenIceCube"));
    }
    {
        const I3CLSimMediumPropertiesPtr m = MediumFromDescription(d);
        // (an override on some layers only: the group velocity would depend on the layer, propagation_kernel.c.cl:525-527)
        m->SetGroupRefractiveIndexOverride(0, I3CLSimFunctionConstPtr());
        clsimhip_glue::MediumHolder h;
        REQUIRE(Fatal([&] { clsimhip_glue::MakeHIPMedium(*m, h); }, "the group refractive index depends on the layer"));
    }
    {
        // no override on any layer: the group velocity comes from the phase index's dispersion
        // (I3CLSimHelperGenerateMediumPropertiesSource.cxx:274-300) -- CLSIMHIP_REFINDEX_DISPERSION, every other number as before
        const I3CLSimMediumPropertiesPtr m = MediumFromDescription(d);
        for (uint32_t i = 0; i < m->GetLayersNum(); ++i) m->SetGroupRefractiveIndexOverride(i, I3CLSimFunctionConstPtr());
        clsimhip_glue::MediumHolder h;
        clsimhip_glue::MakeHIPMedium(*m, h);
        clsimhip_medium_desc got;
        REQUIRE(clsimhip_medium_describe(h.m, &got) == CLSIMHIP_OK);
        REQUIRE(got.group_index_kind == CLSIMHIP_REFINDEX_DISPERSION && got.phase_index_kind == CLSIMHIP_REFINDEX_ICECUBE);
        for (int k = 0; k < 5; ++k) REQUIRE(got.n[k] == d.n[k]);
        std::printf("medium without a group index override: group velocity from the dispersion\n");
    }
    {
        const I3CLSimMediumPropertiesPtr m = MediumFromDescription(d);
        m->SetScatteringCosAngleDistribution(I3CLSimRandomValueConstPtr(new I3CLSimRandomValueConstant(0.9)));
        clsimhip_glue::MediumHolder h;
        REQUIRE(Fatal([&] { clsimhip_glue::MakeHIPMedium(*m, h); }, "scattering angle distribution"));
    }
    {
        const I3CLSimMediumPropertiesPtr m = MediumFromDescription(d);
        m->SetIceTiltZShift(I3CLSimScalarFieldConstPtr(new I3CLSimScalarFieldConstant(1.5)));
        clsimhip_glue::MediumHolder h;
        REQUIRE(Fatal([&] { clsimhip_glue::MakeHIPMedium(*m, h); }, "constant ice tilt shift other than 0"));
    }

    // the converter through its real signatures, up to Compile() (no GPU needed until Initialize)
    Inputs in;
    if (MakeSpectra(ppc, in) != 0) return 1;
    I3CLSimStepToPhotonConverterHIP conv(I3RandomServicePtr(new TestRandomService(1)));
    I3CLSimStepToPhotonConverter &iface = conv;                 // the abstract interface IceTray modules hold
    REQUIRE(Fatal([&] { conv.Compile(); }, "WlenGenerators"));
    REQUIRE(Fatal([&] { iface.EnqueueSteps(I3CLSimStepSeriesConstPtr(), 0); }, "not initialized"));
    REQUIRE(Fatal([&] { iface.SetMediumProperties(I3CLSimMediumPropertiesConstPtr()); }, "(null)"));
    conv.SetDevice(0);
    iface.SetWlenGenerators(in.generators);
    iface.SetWlenBias(in.bias);
    iface.SetMediumProperties(MediumFromDescription(d));
    iface.SetGeometry(SingleString());
    conv.SetEnableDoubleBuffering(false);
    conv.SetDoublePrecision(false);
    conv.SetStopDetectedPhotons(true);
    conv.SetSaveAllPhotons(false);
    conv.SetSaveAllPhotonsPrescale(0.01);
    conv.SetFixedNumberOfAbsorptionLengths(std::numeric_limits<double>::quiet_NaN());
    conv.SetDOMPancakeFactor(5.);
    conv.SetPhotonHistoryEntries(0);
    conv.Compile();
    const std::size_t wg = conv.GetMaxWorkgroupSize();
    conv.SetWorkgroupSize(wg);
    REQUIRE(iface.GetWorkgroupSize() == wg);
    conv.SetMaxNumWorkitems(4 * wg);
    REQUIRE(iface.GetMaxNumWorkitems() == 4 * wg && !iface.IsInitialized());
    // what Compile() built from the objects equals what the ctypes path builds from the same ice directory
    double aDust[512];
    REQUIRE(clsimhip_get_table(conv.Handle(), "aDust400", aDust, 512) == d.num_layers);
    for (int i = 0; i < d.num_layers; ++i) REQUIRE(static_cast<float>(aDust[i]) != 0.f);
    REQUIRE(conv.GetStopDetectedPhotons() && !conv.GetSaveAllPhotons() && conv.GetDOMPancakeFactor() == 5. && conv.GetPhotonHistoryEntries() == 0);
    REQUIRE(conv.GetNumKernelCalls() == 0 && conv.GetTotalNumPhotonsGenerated() == 0);
    {   // an emission spectrum with its own wavelengths (a flasher LED): InterpolatedDistribution(x, y), as makeWavelengthGenerator
        // builds it (I3CLSimModuleHelper.cxx:142-147), reaches the library with its abscissae
        const std::vector<double> x = {350e-9, 361.139e-9, 365.918e-9, 373.362e-9, 400e-9, 455e-9};
        const std::vector<double> y = {0.1, 0.4, 0.9, 1.0, 0.5, 0.0};
        const I3CLSimRandomValueInterpolatedDistribution led(x, y);
        const clsimhip_glue::RandomValueHolder h = clsimhip_glue::MakeHIPWlenGenerator(led, 1);
        REQUIRE(h.r.kind == CLSIMHIP_RANDOM_INTERPOLATED_X && h.r.n == 6 && h.r.x == h.x.data() && h.r.y == h.y.data());
        for (std::size_t i = 0; i < x.size(); ++i) REQUIRE(h.x[i] == x[i] && h.y[i] == y[i]);
        std::vector<I3CLSimRandomValueConstPtr> two = in.generators;
        two.push_back(I3CLSimRandomValueConstPtr(new I3CLSimRandomValueInterpolatedDistribution(x, y)));
        I3CLSimStepToPhotonConverterHIP flasher(I3RandomServicePtr(new TestRandomService(1)));
        I3CLSimStepToPhotonConverter &fi = flasher;
        fi.SetWlenGenerators(two); fi.SetWlenBias(in.bias); fi.SetMediumProperties(MediumFromDescription(d)); fi.SetGeometry(SingleString());
        flasher.Compile();
        double xs[8];
        REQUIRE(clsimhip_get_table(flasher.Handle(), "_generateWavelength_1distXValues", xs, 8) == 6);
        REQUIRE(static_cast<float>(xs[1]) == static_cast<float>(361.139e-9));
    }
    {   // a generator class without a kernel
        std::vector<I3CLSimRandomValueConstPtr> bad(1, I3CLSimRandomValueConstPtr(new I3CLSimRandomValueHenyeyGreenstein(0.9)));
        REQUIRE(Fatal([&] { iface.SetWlenGenerators(bad); }, "without a HIP implementation"));
    }
    clsimhip_medium_destroy(ppc);
    std::printf("icetray adapter ok: interface signatures, glue, refusals, Compile()\n");
    return 0;
}

int Run(const char *ice_dir, const char *geometry_file, const char *steps_file, const char *photons_file, unsigned workitems)
{
    clsimhip_medium *ppc = nullptr;
    REQUIRE(clsimhip_medium_create_from_ppc(ice_dir, 1948.07, 1, &ppc) == CLSIMHIP_OK);
    clsimhip_medium_desc d;
    REQUIRE(clsimhip_medium_describe(ppc, &d) == CLSIMHIP_OK);
    Inputs in;
    if (MakeSpectra(ppc, in) != 0) return 1;
    // geometry file: one DOM per line "string dom x y z subdetector"
    std::vector<int32_t> sid; std::vector<uint32_t> did; std::vector<double> x, y, z; std::vector<std::string> sub;
    {
        std::ifstream f(geometry_file);
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ss(line);
            int s; unsigned dm; double px, py, pz; std::string name;
            if (ss >> s >> dm >> px >> py >> pz >> name) { sid.push_back(s); did.push_back(dm); x.push_back(px); y.push_back(py); z.push_back(pz); sub.push_back(name); }
        }
    }
    REQUIRE(!sid.empty());
    I3CLSimSimpleGeometryUserConfigurablePtr geometry(new I3CLSimSimpleGeometryUserConfigurable(0.16510 * 5., sid.size()));
    for (std::size_t k = 0; k < sid.size(); ++k) {
        geometry->SetStringID(k, sid[k]); geometry->SetDomID(k, did[k]); geometry->SetPosX(k, x[k]); geometry->SetPosY(k, y[k]); geometry->SetPosZ(k, z[k]);
        geometry->SetSubdetector(k, sub[k]);
    }
    boost::shared_ptr<I3CLSimStepSeries> steps(new I3CLSimStepSeries());
    {
        std::ifstream f(steps_file, std::ios::binary);
        I3CLSimStep s;
        while (f.read(reinterpret_cast<char *>(&s), sizeof s)) steps->push_back(s);
    }
    REQUIRE(!steps->empty());
    I3CLSimStepToPhotonConverterPtr conv = I3CLSimModuleHelper::initializeHIP(
        0, workitems, I3RandomServicePtr(new TestRandomService(2024)), geometry, MediumFromDescription(d), in.bias, in.generators,
        false, false, true, false, 0.01, std::numeric_limits<double>::quiet_NaN(), 5., 0, 0);
    REQUIRE(conv->IsInitialized());
    conv->EnqueueSteps(steps, 4711);
    const I3CLSimStepToPhotonConverter::ConversionResult_t r = conv->GetConversionResult();
    REQUIRE(r.identifier == 4711 && r.photons && !r.photonHistories);
    {
        std::ofstream f(photons_file, std::ios::binary);
        f.write(reinterpret_cast<const char *>(r.photons->data()), static_cast<std::streamsize>(r.photons->size() * sizeof(I3CLSimPhoton)));
    }
    const std::map<std::string, double> st = conv->GetStatistics();
    std::printf("identifier %u photons %zu generated %.0f\n", r.identifier, r.photons->size(), st.at("TotalNumPhotonsGenerated"));
    clsimhip_medium_destroy(ppc);
    return 0;
}

// ---- the producer side: I3CLSimLightSourceToStepConverterHIP through the reference's interface ----
I3Particle MakeParticle(I3Particle::ParticleType type, double energy_gev, double length_m)
{
    I3Particle p;
    p.SetType(type);
    p.SetEnergy(energy_gev * I3Units::GeV);
    p.SetPos(I3Position(1. * I3Units::m, -2. * I3Units::m, 3. * I3Units::m));
    p.SetDir(I3Direction(0., 0., -1.));
    p.SetTime(5. * I3Units::ns);
    p.SetLength(length_m * I3Units::m);
    return p;
}

// no GPU: configuration, messages, an empty barrier (one granule of no-op steps)
int LightSourceCheck(const char *ice_dir)
{
    clsimhip_medium *ppc = nullptr;
    REQUIRE(clsimhip_medium_create_from_ppc(ice_dir, 1948.07, 1, &ppc) == CLSIMHIP_OK);
    clsimhip_medium_desc d;
    REQUIRE(clsimhip_medium_describe(ppc, &d) == CLSIMHIP_OK);
    Inputs in;
    if (MakeSpectra(ppc, in) != 0) return 1;
    I3CLSimLightSourceToStepConverterHIP conv(0);
    I3CLSimLightSourceToStepConverter &iface = conv;                 // everything below goes through the reference's interface
    REQUIRE(!iface.IsInitialized());
    REQUIRE(Fatal([&] { iface.EnqueueBarrier(); }, "is not initialized!"));
    REQUIRE(Fatal([&] { iface.SetMaxBunchSize(0); }, "MaxBunchSize of 0 is invalid!"));
    REQUIRE(Fatal([&] { iface.SetBunchSizeGranularity(0); }, "BunchSizeGranularity of 0 is invalid!"));
    REQUIRE(Fatal([&] { iface.Initialize(); }, "WlenBias not set!"));
    iface.SetWlenBias(in.bias);
    REQUIRE(Fatal([&] { iface.Initialize(); }, "MediumProperties not set!"));
    iface.SetMediumProperties(MediumFromDescription(d));
    iface.SetRandomService(I3RandomServicePtr(new TestRandomService(1)));
    iface.SetBunchSizeGranularity(64);
    iface.SetMaxBunchSize(100);
    REQUIRE(Fatal([&] { iface.Initialize(); }, "not a multiple"));
    iface.SetMaxBunchSize(256);
    iface.Initialize();
    REQUIRE(iface.IsInitialized());
    REQUIRE(Fatal([&] { iface.SetMaxBunchSize(512); }, "already initialized!"));
    const double yield = conv.GetMeanPhotonsPerMeter(0);
    REQUIRE(yield > 2000. && yield < 3000.);
    {
        I3CLSimFlasherPulse pulse;
        REQUIRE(Fatal([&] { iface.EnqueueLightSource(I3CLSimLightSource(pulse), 1); }, "only works on particles"));
    }
    bool reset = true;
    REQUIRE(!iface.GetConversionResultWithBarrierInfo(reset, 1e6 * I3Units::ns) && !reset);       // nothing enqueued: timeout, null
    iface.EnqueueBarrier();
    REQUIRE(iface.BarrierActive());
    REQUIRE(Fatal([&] { iface.EnqueueBarrier(); }, "A barrier is already enqueued!"));
    REQUIRE(Fatal([&] { iface.EnqueueLightSource(I3CLSimLightSource(MakeParticle(I3Particle::EMinus, 1., NAN)), 2); }, "A barrier is enqueued!"));
    const I3CLSimStepSeriesConstPtr steps = iface.GetConversionResultWithBarrierInfo(reset);
    REQUIRE(steps && reset && steps->size() == 64 && !iface.BarrierActive());
    for (const I3CLSimStep &s : *steps) REQUIRE(s.GetNumPhotons() == 0 && s.GetWeight() == 0.f && s.GetBeta() == 1.f);
    clsimhip_medium_destroy(ppc);
    std::printf("light source adapter ok: interface signatures, messages, barrier, %.1f photons per metre\n", yield);
    return 0;
}

// GPU: three particles and a barrier; every bunch is appended to `steps_file`, one line per bunch on stdout
int LightSourceRun(const char *ice_dir, const char *steps_file)
{
    clsimhip_medium *ppc = nullptr;
    REQUIRE(clsimhip_medium_create_from_ppc(ice_dir, 1948.07, 1, &ppc) == CLSIMHIP_OK);
    clsimhip_medium_desc d;
    REQUIRE(clsimhip_medium_describe(ppc, &d) == CLSIMHIP_OK);
    Inputs in;
    if (MakeSpectra(ppc, in) != 0) return 1;
    I3CLSimLightSourceToStepConverterHIP conv(0);
    conv.SetSeed(5);
    conv.SetWlenBias(in.bias);
    conv.SetMediumProperties(MediumFromDescription(d));
    conv.SetBunchSizeGranularity(256);
    conv.SetMaxBunchSize(2048);
    conv.Initialize();
    conv.EnqueueLightSource(I3CLSimLightSource(MakeParticle(I3Particle::EMinus, 30., NAN)), 11);
    conv.EnqueueLightSource(I3CLSimLightSource(MakeParticle(I3Particle::MuMinus, 100., 120.)), 12);
    conv.EnqueueLightSource(I3CLSimLightSource(MakeParticle(I3Particle::Hadrons, 50., NAN)), 13);
    conv.EnqueueBarrier();
    std::ofstream f(steps_file, std::ios::binary);
    for (;;) {
        bool reset = false;
        std::vector<uint32_t> finished;
        const I3CLSimStepSeriesConstPtr steps = conv.GetConversionResultWithBarrierInfoAndMarkers(reset, finished, 60e9 * I3Units::ns);
        REQUIRE(steps);
        f.write(reinterpret_cast<const char *>(steps->data()), static_cast<std::streamsize>(steps->size() * sizeof(I3CLSimStep)));
        std::printf("bunch %zu finished", steps->size());
        for (uint32_t id : finished) std::printf(" %u", id);
        std::printf(" reset %d\n", reset ? 1 : 0);
        if (reset) break;
    }
    REQUIRE(!conv.BarrierActive() && !conv.MoreStepsAvailable());
    clsimhip_medium_destroy(ppc);
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    try {
        if (argc >= 3 && std::strcmp(argv[1], "check") == 0) return Check(argv[2], argc > 3 ? argv[3] : nullptr);
        if (argc >= 3 && std::strcmp(argv[1], "lightsource_check") == 0) return LightSourceCheck(argv[2]);
        if (argc >= 4 && std::strcmp(argv[1], "lightsource_run") == 0) return LightSourceRun(argv[2], argv[3]);
        if (argc >= 7 && std::strcmp(argv[1], "run") == 0) return Run(argv[2], argv[3], argv[4], argv[5], static_cast<unsigned>(std::atoi(argv[6])));
    } catch (const std::exception &e) {
        std::printf("FAILED: uncaught %s\n", e.what());
        return 1;
    }
    std::printf("usage: icetray_adapter_test check <ice dir> [<photonics file>] | run <ice dir> <geometry> <steps> <photons out> <workitems>\n");
    return 2;
}
