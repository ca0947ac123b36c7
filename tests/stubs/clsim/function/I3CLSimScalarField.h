// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimScalarField.h:44-84
#pragma once
#include <string>
#include <icetray/I3FrameObject.h>
struct I3CLSimScalarField : public I3FrameObject {
    virtual ~I3CLSimScalarField() {}
    virtual bool HasNativeImplementation() const = 0;
    virtual double GetValue(double x, double y, double z) const = 0;
    virtual std::string GetOpenCLFunction(const std::string &functionName) const = 0;
    virtual bool CompareTo(const I3CLSimScalarField &other) const = 0;
};
I3_POINTER_TYPEDEFS(I3CLSimScalarField);
