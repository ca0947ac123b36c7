// stand-in (tests/stubs/README.md) for dataclasses/I3Direction.h: Cartesian getters and the theta/phi members
// I3CLSimStep.h's inline accessors mention (IceCube convention: theta, phi point back to the source)
#pragma once
#include <cmath>
#include <icetray/I3PointerTypedefs.h>
struct I3Direction {
    I3Direction(double x = 0, double y = 0, double z = 1) : x_(x), y_(y), z_(z) {}
    double GetX() const { return x_; }
    double GetY() const { return y_; }
    double GetZ() const { return z_; }
    double CalcTheta() const { return std::acos(-z_ / std::sqrt(x_ * x_ + y_ * y_ + z_ * z_)); }
    double CalcPhi() const { double p = std::atan2(-y_, -x_); return p < 0 ? p + 2 * M_PI : p; }
    void SetThetaPhi(double theta, double phi)
    {
        x_ = -std::sin(theta) * std::cos(phi); y_ = -std::sin(theta) * std::sin(phi); z_ = -std::cos(theta);
    }
private:
    double x_, y_, z_;
};
I3_POINTER_TYPEDEFS(I3Direction);
