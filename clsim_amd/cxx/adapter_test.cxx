// Exercises the C++ adapter the way I3CLSimModuleHelper::initializeOpenCL drives the reference
// converter (ModuleHelper.cxx:303-372).  `adapter_test` alone checks configuration, Compile() and the
// error behaviour; `adapter_test run` additionally needs a GPU, propagates one bunch of 1024 steps
// (C1: homogeneous ice, single string) and prints the number of detected photons.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "I3CLSimStepToPhotonConverterHIP.h"

#define EXPECT_THROW(stmt, text)                                                                   \
    do {                                                                                           \
        bool thrown = false;                                                                       \
        try { stmt; } catch (const I3CLSimStepToPhotonConverter_exception &e) {                     \
            thrown = std::strstr(e.what(), text) != nullptr;                                        \
            if (!thrown) std::printf("unexpected message: %s\n", e.what());                        \
        }                                                                                          \
        if (!thrown) { std::printf("FAILED: %s did not throw '%s'\n", #stmt, text); return 1; }    \
    } while (0)

static int exercise(bool run, I3CLSimStepToPhotonConverterHIP::ConversionResultView *survivor)
{
    I3CLSimStepToPhotonConverterHIP conv(0);
    EXPECT_THROW(conv.Compile(), "WlenGenerators");
    EXPECT_THROW(conv.EnqueueSteps(I3CLSimStepSeriesConstPtr(), 0), "not initialized");

    // medium: one layer, constant absorption / scattering length (BASELINE config C1)
    clsimhip_medium_desc d;
    std::memset(&d, 0, sizeof d);
    const double absLen = 100., scaLen = 25.;
    d.num_layers = 1; d.layers_z_start = -1000.; d.layers_height = 2000.;
    d.min_wavelength = 265e-9; d.max_wavelength = 675e-9;
    d.lengths_kind = CLSIMHIP_LENGTHS_CONSTANT; d.abs_length = &absLen; d.sca_length = &scaLen;
    const double n[5] = {1.55749, -1.57988, 3.99993, -4.68271, 2.09354}, g[5] = {1.227106, -0.954648, 1.42568, -0.711832, 0.0};
    for (int i = 0; i < 5; ++i) { d.n[i] = n[i]; d.g[i] = g[i]; }
    d.scatter_kind = CLSIMHIP_SCATTER_MIXED; d.liu_fraction = 0.45; d.mean_cosine = 0.9;
    clsimhip_medium *medium = nullptr;
    if (clsimhip_medium_create(&d, &medium) != CLSIMHIP_OK) { std::printf("medium: %s\n", clsimhip_last_error(nullptr)); return 1; }

    std::vector<double> acc(43), y(43);
    double start = 0, step = 0, first = 0, spacing = 0;
    clsimhip_icecube_dom_acceptance(0.16510, 1.0, acc.data(), &start, &step);
    clsimhip_function bias = {CLSIMHIP_FUNCTION_TABLE, 43, start, step, acc.data(), 0., nullptr};
    clsimhip_make_cherenkov_wlen_generator(&bias, medium, y.data(), &first, &spacing);
    clsimhip_random_value gen = {CLSIMHIP_RANDOM_INTERPOLATED, 43, first, spacing, y.data(), 0., nullptr};

    std::vector<int32_t> sid; std::vector<uint32_t> did; std::vector<double> x, yy, z; std::vector<std::string> sub;
    for (int k = 0; k < 60; ++k) { sid.push_back(1); did.push_back(k + 1); x.push_back(20.); yy.push_back(20.); z.push_back(500. - 17. * k); sub.push_back("IceCube"); }

    conv.SetWlenGenerators(std::vector<clsimhip_random_value>(1, gen));
    conv.SetWlenBias(bias);
    conv.SetMediumProperties(medium);
    conv.SetGeometry(sid, did, x, yy, z, sub, 0.16510 * 5.);
    conv.SetEnableDoubleBuffering(false);
    conv.SetDoublePrecision(false);
    conv.SetStopDetectedPhotons(true);
    conv.SetSaveAllPhotons(false);
    conv.SetDOMPancakeFactor(5.);
    conv.SetPhotonHistoryEntries(0);
    conv.SetTuning("slices", 8);
    if (conv.GetTuning("slices") != 8 || conv.GetTuning("kernel") != 0) { std::printf("FAILED: tuning round trip\n"); return 1; }
    EXPECT_THROW(conv.SetTuning("no_such_key", 1), "no tuning key");
    conv.SetTuning("slices", 0);
    conv.Compile();
    const std::size_t wg = conv.GetMaxWorkgroupSize();
    conv.SetWorkgroupSize(wg);
    conv.SetMaxNumWorkitems(1024);
    std::printf("configured: workgroup %zu\n", wg);
    clsimhip_medium_destroy(medium);
    if (!run) { std::printf("adapter ok (no GPU run requested)\n"); return 0; }

    conv.Initialize();
    EXPECT_THROW(conv.SetDOMPancakeFactor(2.), "already initialized");
    std::shared_ptr<I3CLSimStepSeries> steps(new I3CLSimStepSeries(1024));
    for (size_t i = 0; i < steps->size(); ++i) {
        I3CLSimStep &s = (*steps)[i];
        std::memset(&s, 0, sizeof s);
        s.theta = static_cast<float>(std::acos(1. - 2. * ((i * 37) % 1024) / 1024.));
        s.phi = static_cast<float>(6.283185307 * ((i * 101) % 1024) / 1024.);
        s.length = 0.001f; s.beta = 1.f; s.num_photons = (i < 1000) ? 200 : 0; s.weight = 1.f; s.identifier = static_cast<uint32_t>(i);
    }
    EXPECT_THROW(conv.EnqueueSteps(std::shared_ptr<I3CLSimStepSeries>(new I3CLSimStepSeries(100)), 1), "multiple of the workgroup size");
    conv.EnqueueSteps(steps, 42);
    I3CLSimStepToPhotonConverter::ConversionResult_t r = conv.GetConversionResult();
    const std::map<std::string, double> st = conv.GetStatistics();
    std::printf("identifier %u photons %zu generated %.0f\n", r.identifier, r.photons->size(), st.at("TotalNumPhotonsGenerated"));
    for (const I3CLSimPhoton &p : *r.photons)
        if (p.string_id != 1 || p.om_id < 1 || p.om_id > 60) { std::printf("FAILED: bad IDs\n"); return 1; }
    // the same bunch once more through the in-place view: other streams states, so other photons -- checked for shape, IDs and the
    // buffer's return to the pool (a second view after the first was dropped must succeed)
    for (int k = 0; k < 2; ++k) {
        conv.EnqueueSteps(steps, 43 + k);
        I3CLSimStepToPhotonConverterHIP::ConversionResultView v = conv.GetConversionResultInPlace();
        if (v.identifier != uint32_t(43 + k) || v.size == 0 || !v.hold) { std::printf("FAILED: in-place view\n"); return 1; }
        for (const I3CLSimPhoton &p : v)
            if (p.string_id != 1 || p.om_id < 1 || p.om_id > 60) { std::printf("FAILED: bad IDs in the view\n"); return 1; }
        std::printf("view %d: identifier %u photons %zu\n", k, v.identifier, v.size);
    }
    // one more view, handed to the caller: it outlives `conv` (a view keeps the library's converter alive)
    conv.EnqueueSteps(steps, 99);
    *survivor = conv.GetConversionResultInPlace();
    return (r.identifier == 42 && !r.photons->empty() && conv.GetStatistics().at("TotalNumPhotonsGenerated") == 800000.) ? 0 : 1;
}

int main(int argc, char **argv)
{
    const bool run = argc > 1 && std::strcmp(argv[1], "run") == 0;
    I3CLSimStepToPhotonConverterHIP::ConversionResultView survivor;
    const int rc = exercise(run, &survivor);
    if (rc != 0 || !run) return rc;
    // the adapter is gone; the view's records are still there and its buffer can still be given back
    if (survivor.identifier != 99u || survivor.size == 0 || !survivor.hold) { std::printf("FAILED: the surviving view\n"); return 1; }
    for (const I3CLSimPhoton &p : survivor)
        if (p.string_id != 1 || p.om_id < 1 || p.om_id > 60) { std::printf("FAILED: bad IDs in the surviving view\n"); return 1; }
    std::printf("view that outlived its adapter: identifier %u photons %zu\n", survivor.identifier, survivor.size);
    survivor = I3CLSimStepToPhotonConverterHIP::ConversionResultView();        // releases the buffer, then the converter
    return 0;
}
