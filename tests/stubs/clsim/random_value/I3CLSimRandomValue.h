// stand-in (tests/stubs/README.md) for public/clsim/random_value/I3CLSimRandomValue.h:43-99
#pragma once
#include <cmath>
#include <string>
#include <vector>
#include <icetray/I3FrameObject.h>
#include <phys-services/I3RandomService.h>
struct I3CLSimRandomValue : public I3FrameObject {
    virtual ~I3CLSimRandomValue() {}
    virtual double SampleFromDistribution(const I3RandomServicePtr &random, const std::vector<double> &parameters) const = 0;
    virtual std::size_t NumberOfParameters() const = 0;
    virtual bool OpenCLFunctionWillOnlyUseASingleRandomNumber() const = 0;
    virtual std::string GetOpenCLFunction(const std::string &functionName, const std::string &functionArgs, const std::string &functionArgsToCall,
                                          const std::string &uniformRandomCall_co, const std::string &uniformRandomCall_oc) const = 0;
    virtual bool CompareTo(const I3CLSimRandomValue &other) const = 0;
};
I3_POINTER_TYPEDEFS(I3CLSimRandomValue);
#define I3STUB_RANDOM_VALUE_BOILERPLATE                                                                                          \
    virtual double SampleFromDistribution(const I3RandomServicePtr &, const std::vector<double> &) const { return NAN; }          \
    virtual std::size_t NumberOfParameters() const { return 0; }                                                                  \
    virtual bool OpenCLFunctionWillOnlyUseASingleRandomNumber() const { return true; }                                            \
    virtual std::string GetOpenCLFunction(const std::string &, const std::string &, const std::string &, const std::string &,     \
                                          const std::string &) const { return std::string(); }                                   \
    virtual bool CompareTo(const I3CLSimRandomValue &) const { return false; }
