// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimFlasherPulse.h:60-110: the getters of a flasher pulse
#pragma once
#include <dataclasses/physics/I3Particle.h>
class I3CLSimFlasherPulse {
public:
    enum FlasherPulseType { Unknown = 0, LED340nm = 1, LED370nm = 2, LED405nm = 3, LED450nm = 4, LED505nm = 5, SC1 = 6, SC2 = 7 };
    I3CLSimFlasherPulse() : flasherPulseType_(Unknown), time_(0), numberOfPhotonsNoBias_(0), pulseWidth_(0), angularEmissionSigmaPolar_(0),
                            angularEmissionSigmaAzimuthal_(0) {}
    FlasherPulseType GetType() const { return flasherPulseType_; }
    const I3Position &GetPos() const { return pos_; }
    const I3Direction &GetDir() const { return dir_; }
    double GetTime() const { return time_; }
    double GetNumberOfPhotonsNoBias() const { return numberOfPhotonsNoBias_; }
    double GetPulseWidth() const { return pulseWidth_; }
    double GetAngularEmissionSigmaPolar() const { return angularEmissionSigmaPolar_; }
    double GetAngularEmissionSigmaAzimuthal() const { return angularEmissionSigmaAzimuthal_; }
    void SetType(FlasherPulseType t) { flasherPulseType_ = t; }
    void SetPos(const I3Position &p) { pos_ = p; }
    void SetDir(const I3Direction &d) { dir_ = d; }
    void SetTime(double t) { time_ = t; }
    void SetNumberOfPhotonsNoBias(double n) { numberOfPhotonsNoBias_ = n; }
    void SetPulseWidth(double w) { pulseWidth_ = w; }
    void SetAngularEmissionSigmaPolar(double s) { angularEmissionSigmaPolar_ = s; }
    void SetAngularEmissionSigmaAzimuthal(double s) { angularEmissionSigmaAzimuthal_ = s; }
private:
    FlasherPulseType flasherPulseType_;
    I3Position pos_;
    I3Direction dir_;
    double time_, numberOfPhotonsNoBias_, pulseWidth_, angularEmissionSigmaPolar_, angularEmissionSigmaAzimuthal_;
};
