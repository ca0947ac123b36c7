// stand-in (tests/stubs/README.md) for public/clsim/random_value/I3CLSimRandomValueMixed.h:38-78: no getters; private
// members fractionOfFirstDistribution_, firstDistribution_, secondDistribution_ (:65-67)
#pragma once
#include <clsim/random_value/I3CLSimRandomValue.h>
struct I3CLSimRandomValueMixed : public I3CLSimRandomValue {
    I3CLSimRandomValueMixed(double fractionOfFirstDistribution, I3CLSimRandomValueConstPtr firstDistribution, I3CLSimRandomValueConstPtr secondDistribution)
        : fractionOfFirstDistribution_(fractionOfFirstDistribution), firstDistribution_(firstDistribution), secondDistribution_(secondDistribution) {}
    I3STUB_RANDOM_VALUE_BOILERPLATE
private:
    I3CLSimRandomValueMixed();
    double fractionOfFirstDistribution_;
    I3CLSimRandomValueConstPtr firstDistribution_;
    I3CLSimRandomValueConstPtr secondDistribution_;
};
I3_POINTER_TYPEDEFS(I3CLSimRandomValueMixed);
