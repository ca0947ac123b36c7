// stand-in (tests/stubs/README.md) for public/clsim/random_value/I3CLSimRandomValueConstant.h:40-82 (value_ :72, no getter)
#pragma once
#include <clsim/random_value/I3CLSimRandomValue.h>
struct I3CLSimRandomValueConstant : public I3CLSimRandomValue {
    I3CLSimRandomValueConstant() : value_(NAN) {}
    I3CLSimRandomValueConstant(double value) : value_(value) {}
    I3STUB_RANDOM_VALUE_BOILERPLATE
private:
    double value_;
};
I3_POINTER_TYPEDEFS(I3CLSimRandomValueConstant);
