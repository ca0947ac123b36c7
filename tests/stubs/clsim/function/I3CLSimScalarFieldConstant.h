// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimScalarFieldConstant.h:40-82 (value_ :79, no getter)
#pragma once
#include <clsim/function/I3CLSimScalarField.h>
struct I3CLSimScalarFieldConstant : public I3CLSimScalarField {
    I3CLSimScalarFieldConstant(double value) : value_(value) {}
    virtual bool HasNativeImplementation() const { return true; }
    virtual double GetValue(double, double, double) const { return value_; }
    virtual std::string GetOpenCLFunction(const std::string &) const { return std::string(); }
    virtual bool CompareTo(const I3CLSimScalarField &other) const
    {
        const I3CLSimScalarFieldConstant *o = dynamic_cast<const I3CLSimScalarFieldConstant *>(&other);
        return o && o->value_ == value_;
    }
private:
    I3CLSimScalarFieldConstant();
    double value_;
};
I3_POINTER_TYPEDEFS(I3CLSimScalarFieldConstant);
