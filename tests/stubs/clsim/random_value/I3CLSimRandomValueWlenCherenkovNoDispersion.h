// stand-in (tests/stubs/README.md) for public/clsim/random_value/I3CLSimRandomValueWlenCherenkovNoDispersion.h:38-80
// (fromWlen_, toWlen_ :69-70, no getters)
#pragma once
#include <clsim/random_value/I3CLSimRandomValue.h>
struct I3CLSimRandomValueWlenCherenkovNoDispersion : public I3CLSimRandomValue {
    I3CLSimRandomValueWlenCherenkovNoDispersion(double fromWlen, double toWlen) : fromWlen_(fromWlen), toWlen_(toWlen) {}
    I3STUB_RANDOM_VALUE_BOILERPLATE
private:
    I3CLSimRandomValueWlenCherenkovNoDispersion();
    double fromWlen_;
    double toWlen_;
};
I3_POINTER_TYPEDEFS(I3CLSimRandomValueWlenCherenkovNoDispersion);
