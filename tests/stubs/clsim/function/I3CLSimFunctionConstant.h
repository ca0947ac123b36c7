// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimFunctionConstant.h:40-104 (no getter; value_ :102)
#pragma once
#include <limits>
#include <clsim/function/I3CLSimFunction.h>
struct I3CLSimFunctionConstant : public I3CLSimFunction {
    I3CLSimFunctionConstant(double value) : value_(value) {}
    I3STUB_FUNCTION_BOILERPLATE
    virtual double GetMinWlen() const { return -std::numeric_limits<double>::infinity(); }
    virtual double GetMaxWlen() const { return std::numeric_limits<double>::infinity(); }
    virtual bool CompareTo(const I3CLSimFunction &other) const
    {
        const I3CLSimFunctionConstant *o = dynamic_cast<const I3CLSimFunctionConstant *>(&other);
        return o && o->value_ == value_;
    }
private:
    I3CLSimFunctionConstant();
    double value_;
};
I3_POINTER_TYPEDEFS(I3CLSimFunctionConstant);
