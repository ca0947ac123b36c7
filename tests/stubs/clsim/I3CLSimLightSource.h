// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimLightSource.h:47-83
#pragma once
#include <vector>
#include <clsim/I3CLSimFlasherPulse.h>
#include <dataclasses/physics/I3Particle.h>
#include <icetray/I3PointerTypedefs.h>
class I3CLSimLightSource {
public:
    enum LightSourceType { Unknown = 0, Particle = 1, Flasher = 2 };
    I3CLSimLightSource(const I3Particle &particle) : lightSourceType_(Particle), particle_(particle) {}
    I3CLSimLightSource(const I3CLSimFlasherPulse &flasher) : lightSourceType_(Flasher), flasher_(flasher) {}
    LightSourceType GetType() const { return lightSourceType_; }
    const I3Particle &GetParticle() const { return particle_; }
    const I3CLSimFlasherPulse &GetFlasherPulse() const { return flasher_; }
private:
    LightSourceType lightSourceType_;
    I3Particle particle_;
    I3CLSimFlasherPulse flasher_;
};
typedef std::vector<I3CLSimLightSource> I3CLSimLightSourceSeries;
I3_POINTER_TYPEDEFS(I3CLSimLightSource);
