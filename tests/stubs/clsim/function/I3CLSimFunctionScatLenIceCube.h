// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimFunctionScatLenIceCube.h:40-103 (getters :88-89)
#pragma once
#include <limits>
#include <clsim/function/I3CLSimFunction.h>
struct I3CLSimFunctionScatLenIceCube : public I3CLSimFunction {
    I3CLSimFunctionScatLenIceCube(double alpha, double b400) : alpha_(alpha), b400_(b400) {}
    I3STUB_FUNCTION_BOILERPLATE
    virtual double GetMinWlen() const { return -std::numeric_limits<double>::infinity(); }
    virtual double GetMaxWlen() const { return std::numeric_limits<double>::infinity(); }
    virtual bool CompareTo(const I3CLSimFunction &other) const
    {
        const I3CLSimFunctionScatLenIceCube *o = dynamic_cast<const I3CLSimFunctionScatLenIceCube *>(&other);
        return o && o->alpha_ == alpha_ && o->b400_ == b400_;
    }
    double GetAlpha() const { return alpha_; }
    double GetB400() const { return b400_; }
private:
    I3CLSimFunctionScatLenIceCube();
    double alpha_, b400_;
};
I3_POINTER_TYPEDEFS(I3CLSimFunctionScatLenIceCube);
