// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimFunctionAbsLenIceCube.h:40-115 (getters :93-99)
#pragma once
#include <limits>
#include <clsim/function/I3CLSimFunction.h>
struct I3CLSimFunctionAbsLenIceCube : public I3CLSimFunction {
    I3CLSimFunctionAbsLenIceCube(double kappa, double A, double B, double D, double E, double aDust400, double deltaTau)
        : kappa_(kappa), A_(A), B_(B), D_(D), E_(E), aDust400_(aDust400), deltaTau_(deltaTau) {}
    I3STUB_FUNCTION_BOILERPLATE
    virtual double GetMinWlen() const { return -std::numeric_limits<double>::infinity(); }
    virtual double GetMaxWlen() const { return std::numeric_limits<double>::infinity(); }
    virtual bool CompareTo(const I3CLSimFunction &other) const
    {
        const I3CLSimFunctionAbsLenIceCube *o = dynamic_cast<const I3CLSimFunctionAbsLenIceCube *>(&other);
        return o && o->kappa_ == kappa_ && o->A_ == A_ && o->B_ == B_ && o->D_ == D_ && o->E_ == E_ && o->aDust400_ == aDust400_ && o->deltaTau_ == deltaTau_;
    }
    double GetKappa() const { return kappa_; }
    double GetA() const { return A_; }
    double GetB() const { return B_; }
    double GetD() const { return D_; }
    double GetE() const { return E_; }
    double GetADust400() const { return aDust400_; }
    double GetDeltaTau() const { return deltaTau_; }
private:
    I3CLSimFunctionAbsLenIceCube();
    double kappa_, A_, B_, D_, E_, aDust400_, deltaTau_;
};
I3_POINTER_TYPEDEFS(I3CLSimFunctionAbsLenIceCube);
