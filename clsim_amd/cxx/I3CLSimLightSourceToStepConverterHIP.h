// C++ adapter for the producer side: the reference's light-source converter interface
// (public/clsim/I3CLSimLightSourceToStepConverter.h:61-198) implemented on the C ABI's feeder (clsimhip_feeder_*), the PPC
// front end (clsimhip_ppc_*) and the GPU step producer.  It takes the place of
//   I3CLSimLightSourceToStepConverterAsync with I3CLSimLightSourceToStepConverterPPC as its only parameterisation
//   (private/clsim/I3CLSimLightSourceToStepConverterAsync.cxx, ...PPC.cxx)
// for particles: same setters, same queue / barrier semantics, same exception messages.  Flasher pulses and Geant4
// propagators are not handled by this class (flashers: clsimhip_flasher_enqueue + clsimhip_generate_flasher_steps).
//
// IceTray build only (-DCLSIMHIP_WITH_ICETRAY); this repository compiles and runs it against tests/stubs/
// (tests/test_icetray_adapter.py).  The random service is not consulted: the library draws photon numbers from its own
// counter-based generator seeded through SetSeed() (sim-services / GSL are not part of the reference
// tree), so results do not depend on the order light sources are enqueued in.
#pragma once
#ifndef CLSIMHIP_WITH_ICETRAY
#error "I3CLSimLightSourceToStepConverterHIP.h is the IceTray-side adapter: compile with -DCLSIMHIP_WITH_ICETRAY"
#endif
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include <clsim/I3CLSimLightSourceToStepConverter.h>
#include <icetray/I3Units.h>

#include "../../include/clsimhip.h"
#include "I3CLSimStepToPhotonConverterHIPGlue.h"

class I3CLSimLightSourceToStepConverterHIP : public I3CLSimLightSourceToStepConverter {
public:
    // photonsPerStep, highPhotonsPerStep, useHighPhotonsPerStepStartingFromNumPhotons: I3CLSimLightSourceToStepConverterPPC (:51-58);
    // maxQueueItems: I3CLSimLightSourceToStepConverterAsync (:58-66)
    explicit I3CLSimLightSourceToStepConverterHIP(int device = 0, uint32_t photonsPerStep = 200, uint32_t highPhotonsPerStep = 2000,
                                                  double useHighPhotonsPerStepStartingFromNumPhotons = 1.0e9, uint32_t maxQueueItems = 10)
        : device_(device), maxQueueItems_(maxQueueItems), granularity_(512), maxBunchSize_(512000), seed_(0), ppc_(0), feeder_(0)
    {
        if (photonsPerStep == 0 || highPhotonsPerStep == 0) throw I3CLSimLightSourceToStepConverter_exception("photonsPerStep may not be <= 0!");
        std::memset(&config_, 0, sizeof config_);
        config_.photons_per_step = photonsPerStep;
        config_.high_photons_per_step = highPhotonsPerStep;
        config_.use_high_photons_per_step_from = useHighPhotonsPerStepStartingFromNumPhotons;
        config_.use_cascade_extension = 1;
    }
    ~I3CLSimLightSourceToStepConverterHIP() override
    {
        if (feeder_) clsimhip_feeder_destroy(feeder_);
        if (ppc_) clsimhip_ppc_destroy(ppc_);
    }
    void SetUseCascadeExtension(bool v) { not_initialized(); config_.use_cascade_extension = v ? 1 : 0; }
    void SetSeed(uint64_t seed) { not_initialized(); seed_ = seed; }

    void SetBunchSizeGranularity(uint64_t num) override
    {
        not_initialized();
        if (num <= 0) throw I3CLSimLightSourceToStepConverter_exception("BunchSizeGranularity of 0 is invalid!");
        granularity_ = num;
    }
    void SetMaxBunchSize(uint64_t num) override
    {
        not_initialized();
        if (num <= 0) throw I3CLSimLightSourceToStepConverter_exception("MaxBunchSize of 0 is invalid!");
        maxBunchSize_ = num;
    }
    void SetRandomService(I3RandomServicePtr random) override { not_initialized(); randomService_ = random; }
    void SetWlenBias(I3CLSimFunctionConstPtr wlenBias) override { not_initialized(); wlenBias_ = wlenBias; }
    void SetMediumProperties(I3CLSimMediumPropertiesConstPtr mediumProperties) override { not_initialized(); mediumProperties_ = mediumProperties; }

    void Initialize() override
    {
        if (feeder_) throw I3CLSimLightSourceToStepConverter_exception("I3CLSimLightSourceToStepConverterHIP already initialized!");
        if (!wlenBias_) throw I3CLSimLightSourceToStepConverter_exception("WlenBias not set!");
        if (!mediumProperties_) throw I3CLSimLightSourceToStepConverter_exception("MediumProperties not set!");
        if (maxBunchSize_ % granularity_ != 0) throw I3CLSimLightSourceToStepConverter_exception("MaxBunchSize is not a multiple of BunchSizeGranularity!");
        clsimhip_glue::MediumHolder medium;
        clsimhip_glue::MakeHIPMedium(*mediumProperties_, medium);
        const clsimhip_glue::FunctionHolder bias = clsimhip_glue::MakeHIPFunction(*wlenBias_, "the wavelength bias");
        config_.medium_density = mediumProperties_->GetMediumDensity() / (I3Units::g / I3Units::cm3);
        config_.seed = seed_;
        check(clsimhip_ppc_create(medium.m, &bias.f, &config_, &ppc_));
        check(clsimhip_feeder_create(ppc_, device_, seed_, static_cast<size_t>(maxBunchSize_), static_cast<size_t>(granularity_), maxQueueItems_, &feeder_));
    }
    bool IsInitialized() const override { return feeder_ != 0; }

    // I3CLSimLightSourceToStepConverterPPC::EnqueueLightSource (:188-200) in front of the feeder's queue
    void EnqueueLightSource(const I3CLSimLightSource &lightSource, uint32_t identifier) override
    {
        initialized();
        if (lightSource.GetType() != I3CLSimLightSource::Particle)
            throw I3CLSimLightSourceToStepConverter_exception("The I3CLSimLightSourceToStepConverterPPC parameterization only works on particles.");
        const I3Particle &p = lightSource.GetParticle();
        clsimhip_particle c;
        std::memset(&c, 0, sizeof c);
        c.type = static_cast<int32_t>(p.GetType());
        c.shape = (p.GetShape() == I3Particle::CascadeSegment) ? CLSIMHIP_SHAPE_CASCADE_SEGMENT : CLSIMHIP_SHAPE_OTHER;
        c.x = p.GetPos().GetX() / I3Units::m; c.y = p.GetPos().GetY() / I3Units::m; c.z = p.GetPos().GetZ() / I3Units::m;
        c.time = p.GetTime() / I3Units::ns;
        c.dx = p.GetDir().GetX(); c.dy = p.GetDir().GetY(); c.dz = p.GetDir().GetZ();
        c.energy = p.GetEnergy() / I3Units::GeV;
        c.length = p.GetLength() / I3Units::m;
        c.identifier = identifier;
        check(clsimhip_feeder_enqueue_light_source(feeder_, &c));
    }
    void EnqueueBarrier() override { initialized(); check(clsimhip_feeder_enqueue_barrier(feeder_)); }
    bool BarrierActive() const override
    {
        initialized();
        int v = 0;
        check(clsimhip_feeder_barrier_active(feeder_, &v));
        return v != 0;
    }
    bool MoreStepsAvailable() const override
    {
        initialized();
        int v = 0;
        check(clsimhip_feeder_more_steps_available(feeder_, &v));
        return v != 0;
    }
    // GetConversionResultWithBarrierInfoAndMarkers: `finished` receives the identifiers of the light sources whose steps
    // have all been handed out; timeout in I3Units of time (NaN: wait for ever), a null pointer on timeout
    I3CLSimStepSeriesConstPtr GetConversionResultWithBarrierInfoAndMarkers(bool &barrierWasReset, std::vector<uint32_t> &finished, double timeout = NAN)
    {
        initialized();
        int got = 0, reset = 0;
        const clsimhip_step *steps = 0;
        const uint32_t *fin = 0;
        size_t n = 0, nfin = 0;
        const double timeout_ms = std::isnan(timeout) ? -1. : timeout / I3Units::ns * 1e-6;
        check(clsimhip_feeder_get_conversion_result(feeder_, timeout_ms, &got, &steps, &n, &fin, &nfin, &reset));
        barrierWasReset = false;
        finished.clear();
        if (!got) return I3CLSimStepSeriesConstPtr();
        I3CLSimStepSeriesPtr out(new I3CLSimStepSeries());
        out->resize(n);
        static_assert(sizeof(I3CLSimStep) == sizeof(clsimhip_step), "I3CLSimStep is the 48-byte record of the C ABI");
        if (n) std::memcpy(static_cast<void *>(&(*out)[0]), steps, n * sizeof(clsimhip_step));
        finished.assign(fin, fin + nfin);
        barrierWasReset = reset != 0;
        check(clsimhip_feeder_release_result(feeder_, steps));
        return out;
    }
    I3CLSimStepSeriesConstPtr GetConversionResultWithBarrierInfo(bool &barrierWasReset, double timeout = NAN) override
    {
        std::vector<uint32_t> finished;
        return GetConversionResultWithBarrierInfoAndMarkers(barrierWasReset, finished, timeout);
    }
    // meanPhotonsPerMeterInLayer_ (PPC.cxx:113-131)
    double GetMeanPhotonsPerMeter(uint32_t layer = 0) const
    {
        initialized();
        double v = 0;
        check(clsimhip_ppc_photons_per_meter(ppc_, static_cast<int>(layer), &v));
        return v;
    }

private:
    void not_initialized() const
    {
        if (feeder_) throw I3CLSimLightSourceToStepConverter_exception("I3CLSimLightSourceToStepConverterHIP already initialized!");
    }
    void initialized() const
    {
        if (!feeder_) throw I3CLSimLightSourceToStepConverter_exception("I3CLSimLightSourceToStepConverterHIP is not initialized!");
    }
    static void check(int rc)
    {
        if (rc != CLSIMHIP_OK) {
            const char *msg = clsimhip_last_error(0);
            throw I3CLSimLightSourceToStepConverter_exception(msg ? msg : "libclsimhip call failed");
        }
    }
    int device_;
    uint32_t maxQueueItems_;
    uint64_t granularity_, maxBunchSize_, seed_;
    clsimhip_ppc_config config_;
    I3RandomServicePtr randomService_;
    I3CLSimFunctionConstPtr wlenBias_;
    I3CLSimMediumPropertiesConstPtr mediumProperties_;
    clsimhip_ppc_converter *ppc_;
    clsimhip_feeder *feeder_;
};
