// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimFunctionRefIndexIceCube.h:40-142: no getters;
// private members mode_, n0_..n4_, g0_..g4_ (:126-136)
#pragma once
#include <limits>
#include <string>
#include <clsim/function/I3CLSimFunction.h>
struct I3CLSimFunctionRefIndexIceCube : public I3CLSimFunction {
    I3CLSimFunctionRefIndexIceCube(std::string mode, double n0, double n1, double n2, double n3, double n4,
                                   double g0, double g1, double g2, double g3, double g4)
        : mode_(mode), n0_(n0), n1_(n1), n2_(n2), n3_(n3), n4_(n4), g0_(g0), g1_(g1), g2_(g2), g3_(g3), g4_(g4) {}
    virtual bool HasNativeImplementation() const { return true; }
    virtual bool HasDerivative() const { return true; }
    virtual double GetValue(double) const { return NAN; }
    virtual std::string GetOpenCLFunction(const std::string &) const { return std::string(); }
    virtual double GetMinWlen() const { return -std::numeric_limits<double>::infinity(); }
    virtual double GetMaxWlen() const { return std::numeric_limits<double>::infinity(); }
    virtual bool CompareTo(const I3CLSimFunction &other) const
    {
        const I3CLSimFunctionRefIndexIceCube *o = dynamic_cast<const I3CLSimFunctionRefIndexIceCube *>(&other);
        return o && o->mode_ == mode_ && o->n0_ == n0_ && o->n1_ == n1_ && o->n2_ == n2_ && o->n3_ == n3_ && o->n4_ == n4_ &&
               o->g0_ == g0_ && o->g1_ == g1_ && o->g2_ == g2_ && o->g3_ == g3_ && o->g4_ == g4_;
    }
private:
    std::string mode_;
    double n0_, n1_, n2_, n3_, n4_, g0_, g1_, g2_, g3_, g4_;
};
I3_POINTER_TYPEDEFS(I3CLSimFunctionRefIndexIceCube);
