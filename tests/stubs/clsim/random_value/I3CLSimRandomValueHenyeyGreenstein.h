// stand-in (tests/stubs/README.md) for public/clsim/random_value/I3CLSimRandomValueHenyeyGreenstein.h:43-82 (meanCosine_ :72)
#pragma once
#include <clsim/random_value/I3CLSimRandomValue.h>
struct I3CLSimRandomValueHenyeyGreenstein : public I3CLSimRandomValue {
    I3CLSimRandomValueHenyeyGreenstein(double meanCosine) : meanCosine_(meanCosine) {}
    I3STUB_RANDOM_VALUE_BOILERPLATE
private:
    I3CLSimRandomValueHenyeyGreenstein();
    double meanCosine_;
};
I3_POINTER_TYPEDEFS(I3CLSimRandomValueHenyeyGreenstein);
