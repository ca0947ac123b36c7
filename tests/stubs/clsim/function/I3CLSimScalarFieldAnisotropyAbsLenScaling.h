// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimScalarFieldAnisotropyAbsLenScaling.h:45-100: no
// getters; private members anisotropyDirAzimuth_, magnitudeAlongDir_, magnitudePerpToDir_ (:87-89)
#pragma once
#include <cmath>
#include <clsim/function/I3CLSimScalarField.h>
struct I3CLSimScalarFieldAnisotropyAbsLenScaling : public I3CLSimScalarField {
    I3CLSimScalarFieldAnisotropyAbsLenScaling(double anisotropyDirAzimuth, double magnitudeAlongDir, double magnitudePerpToDir)
        : anisotropyDirAzimuth_(anisotropyDirAzimuth), magnitudeAlongDir_(magnitudeAlongDir), magnitudePerpToDir_(magnitudePerpToDir) {}
    virtual bool HasNativeImplementation() const { return true; }
    virtual double GetValue(double, double, double) const { return NAN; }
    virtual std::string GetOpenCLFunction(const std::string &) const { return std::string(); }
    virtual bool CompareTo(const I3CLSimScalarField &) const { return false; }
private:
    double anisotropyDirAzimuth_;
    double magnitudeAlongDir_;
    double magnitudePerpToDir_;
};
I3_POINTER_TYPEDEFS(I3CLSimScalarFieldAnisotropyAbsLenScaling);
