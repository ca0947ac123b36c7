// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimFunction.h:41-97
#pragma once
#include <cmath>
#include <string>
#include <icetray/I3FrameObject.h>
struct I3CLSimFunction : public I3FrameObject {
    virtual ~I3CLSimFunction() {}
    virtual bool HasNativeImplementation() const = 0;
    virtual bool HasDerivative() const = 0;
    virtual double GetValue(double wlen) const = 0;
    virtual double GetDerivative(double) const { return NAN; }
    virtual double GetMinWlen() const = 0;
    virtual double GetMaxWlen() const = 0;
    virtual std::string GetOpenCLFunction(const std::string &functionName) const = 0;
    virtual std::string GetOpenCLFunctionDerivative(const std::string &) const { return std::string(); }
    virtual bool CompareTo(const I3CLSimFunction &other) const = 0;
};
I3_POINTER_TYPEDEFS(I3CLSimFunction);
// what the stand-ins answer where the glue never asks (no reference arithmetic is restated here)
#define I3STUB_FUNCTION_BOILERPLATE                                                                 \
    virtual bool HasNativeImplementation() const { return true; }                                   \
    virtual bool HasDerivative() const { return false; }                                            \
    virtual double GetValue(double) const { return NAN; }                                            \
    virtual std::string GetOpenCLFunction(const std::string &) const { return std::string(); }
