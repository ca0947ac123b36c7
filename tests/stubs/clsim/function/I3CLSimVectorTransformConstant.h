// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimVectorTransformConstant.h:40-72 (the identity)
#pragma once
#include <clsim/function/I3CLSimVectorTransform.h>
struct I3CLSimVectorTransformConstant : public I3CLSimVectorTransform {
    I3CLSimVectorTransformConstant() {}
    virtual bool HasNativeImplementation() const { return true; }
    virtual std::vector<double> ApplyTransform(const std::vector<double> &vec) const { return vec; }
    virtual std::string GetOpenCLFunction(const std::string &) const { return std::string(); }
    virtual bool CompareTo(const I3CLSimVectorTransform &other) const { return dynamic_cast<const I3CLSimVectorTransformConstant *>(&other) != 0; }
};
I3_POINTER_TYPEDEFS(I3CLSimVectorTransformConstant);
