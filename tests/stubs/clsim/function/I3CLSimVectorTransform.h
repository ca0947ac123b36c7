// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimVectorTransform.h:44-77
#pragma once
#include <string>
#include <vector>
#include <icetray/I3FrameObject.h>
struct I3CLSimVectorTransform : public I3FrameObject {
    virtual ~I3CLSimVectorTransform() {}
    virtual bool HasNativeImplementation() const = 0;
    virtual std::vector<double> ApplyTransform(const std::vector<double> &vec) const = 0;
    virtual std::string GetOpenCLFunction(const std::string &functionName) const = 0;
    virtual bool CompareTo(const I3CLSimVectorTransform &other) const = 0;
};
I3_POINTER_TYPEDEFS(I3CLSimVectorTransform);
