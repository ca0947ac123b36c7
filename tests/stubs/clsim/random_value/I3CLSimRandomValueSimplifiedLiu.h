// stand-in (tests/stubs/README.md) for public/clsim/random_value/I3CLSimRandomValueSimplifiedLiu.h:46-82 (meanCosine_ :72)
#pragma once
#include <clsim/random_value/I3CLSimRandomValue.h>
struct I3CLSimRandomValueSimplifiedLiu : public I3CLSimRandomValue {
    I3CLSimRandomValueSimplifiedLiu(double meanCosine) : meanCosine_(meanCosine) {}
    I3STUB_RANDOM_VALUE_BOILERPLATE
private:
    I3CLSimRandomValueSimplifiedLiu();
    double meanCosine_;
};
I3_POINTER_TYPEDEFS(I3CLSimRandomValueSimplifiedLiu);
