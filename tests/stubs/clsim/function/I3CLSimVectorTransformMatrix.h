// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimVectorTransformMatrix.h:41-80: no getters; private
// members matrix_ (3x3 I3Matrix), renormalize_ (:67-68)
#pragma once
#include <stdexcept>
#include <dataclasses/I3Matrix.h>
#include <clsim/function/I3CLSimVectorTransform.h>
struct I3CLSimVectorTransformMatrix : public I3CLSimVectorTransform {
    I3CLSimVectorTransformMatrix(const I3Matrix &matrix, bool renormalize = false) : matrix_(matrix), renormalize_(renormalize)
    {
        if (matrix_.size1() != 3 || matrix_.size2() != 3) throw std::runtime_error("matrix must be 3x3!");
    }
    virtual bool HasNativeImplementation() const { return true; }
    virtual std::vector<double> ApplyTransform(const std::vector<double> &vec) const { return vec; }
    virtual std::string GetOpenCLFunction(const std::string &) const { return std::string(); }
    virtual bool CompareTo(const I3CLSimVectorTransform &) const { return false; }
private:
    I3CLSimVectorTransformMatrix();
    I3Matrix matrix_;
    bool renormalize_;
};
I3_POINTER_TYPEDEFS(I3CLSimVectorTransformMatrix);
