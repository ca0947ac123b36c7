// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimScalarFieldIceTiltZShift.h:41-96: no getters; private
// members distancesFromOriginAlongTilt_, zCoordinates_, zCorrections_ (I3Matrix [distance][z]), directionOfTiltAzimuth_ (:82-85)
#pragma once
#include <cmath>
#include <vector>
#include <dataclasses/I3Matrix.h>
#include <clsim/function/I3CLSimScalarField.h>
struct I3CLSimScalarFieldIceTiltZShift : public I3CLSimScalarField {
    I3CLSimScalarFieldIceTiltZShift(const std::vector<double> &distancesFromOriginAlongTilt, const std::vector<double> &zCoordinates,
                                    const I3Matrix &zCorrections, double directionOfTiltAzimuth)
        : distancesFromOriginAlongTilt_(distancesFromOriginAlongTilt), zCoordinates_(zCoordinates), zCorrections_(zCorrections),
          directionOfTiltAzimuth_(directionOfTiltAzimuth), firstZCoordinate_(zCoordinates.empty() ? NAN : zCoordinates[0]),
          zCoordinateSpacing_(zCoordinates.size() > 1 ? zCoordinates[1] - zCoordinates[0] : NAN) {}
    virtual bool HasNativeImplementation() const { return true; }
    virtual double GetValue(double, double, double) const { return NAN; }
    virtual std::string GetOpenCLFunction(const std::string &) const { return std::string(); }
    virtual bool CompareTo(const I3CLSimScalarField &) const { return false; }
private:
    I3CLSimScalarFieldIceTiltZShift();
    std::vector<double> distancesFromOriginAlongTilt_;
    std::vector<double> zCoordinates_;
    I3Matrix zCorrections_;
    double directionOfTiltAzimuth_;
    double firstZCoordinate_;
    double zCoordinateSpacing_;
};
I3_POINTER_TYPEDEFS(I3CLSimScalarFieldIceTiltZShift);
