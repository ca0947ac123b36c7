// stand-in (tests/stubs/README.md) for public/clsim/random_value/I3CLSimRandomValueInterpolatedDistribution.h:43-92: no
// getters; private members x_, y_, constantXSpacing_, firstX_ (:80-83; NaN spacing = tabulated x values)
#pragma once
#include <clsim/random_value/I3CLSimRandomValue.h>
struct I3CLSimRandomValueInterpolatedDistribution : public I3CLSimRandomValue {
    I3CLSimRandomValueInterpolatedDistribution(const std::vector<double> &x, const std::vector<double> &y)
        : x_(x), y_(y), constantXSpacing_(NAN), firstX_(NAN) {}
    I3CLSimRandomValueInterpolatedDistribution(double xFirst, double xSpacing, const std::vector<double> &y)
        : y_(y), constantXSpacing_(xSpacing), firstX_(xFirst) {}
    I3STUB_RANDOM_VALUE_BOILERPLATE
private:
    I3CLSimRandomValueInterpolatedDistribution();
    std::vector<double> data_acu_, data_beta_;
    std::vector<double> x_, y_;
    double constantXSpacing_;
    double firstX_;
};
I3_POINTER_TYPEDEFS(I3CLSimRandomValueInterpolatedDistribution);
