// stand-in (tests/stubs/README.md) for dataclasses/I3Position.h: Cartesian getters only
#pragma once
#include <icetray/I3PointerTypedefs.h>
struct I3Position {
    I3Position(double x = 0, double y = 0, double z = 0) : x_(x), y_(y), z_(z) {}
    double GetX() const { return x_; }
    double GetY() const { return y_; }
    double GetZ() const { return z_; }
private:
    double x_, y_, z_;
};
I3_POINTER_TYPEDEFS(I3Position);
