// stand-in (tests/stubs/README.md) for public/clsim/function/I3CLSimFunctionFromTable.h:41-143 (getters :87-112;
// private storeDataAsHalfPrecision_ :135 has none)
#pragma once
#include <stdexcept>
#include <vector>
#include <clsim/function/I3CLSimFunction.h>
struct I3CLSimFunctionFromTable : public I3CLSimFunction {
    I3CLSimFunctionFromTable(double startWlen, double wlenStep, const std::vector<double> &values, bool storeDataAsHalfPrecision = false)
        : startWlen_(startWlen), wlenStep_(wlenStep), values_(values), equalSpacingMode_(true), storeDataAsHalfPrecision_(storeDataAsHalfPrecision)
    {
        if (values_.size() < 2) throw std::runtime_error("values must contain at least 2 elements!");
    }
    I3CLSimFunctionFromTable(const std::vector<double> &wlens, const std::vector<double> &values, bool storeDataAsHalfPrecision = false)
        : startWlen_(NAN), wlenStep_(NAN), wlens_(wlens), values_(values), equalSpacingMode_(false), storeDataAsHalfPrecision_(storeDataAsHalfPrecision) {}
    I3STUB_FUNCTION_BOILERPLATE
    virtual double GetMinWlen() const { return equalSpacingMode_ ? startWlen_ : wlens_.front(); }
    virtual double GetMaxWlen() const { return equalSpacingMode_ ? startWlen_ + wlenStep_ * static_cast<double>(values_.size() - 1) : wlens_.back(); }
    virtual bool CompareTo(const I3CLSimFunction &other) const
    {
        const I3CLSimFunctionFromTable *o = dynamic_cast<const I3CLSimFunctionFromTable *>(&other);
        return o && o->startWlen_ == startWlen_ && o->wlenStep_ == wlenStep_ && o->values_ == values_ && o->wlens_ == wlens_;
    }
    double GetFirstWavelength() const { return startWlen_; }
    double GetWavelengthStepping() const { return wlenStep_; }
    std::size_t GetNumEntries() const { return values_.size(); }
    double GetEntryValue(std::size_t i) const { return values_[i]; }
    double GetEntryWavelength(std::size_t i) const { return wlens_[i]; }
    bool GetInEqualSpacingMode() const { return equalSpacingMode_; }
private:
    I3CLSimFunctionFromTable();
    double startWlen_, wlenStep_;
    std::vector<double> wlens_, values_;
    bool equalSpacingMode_;
    bool storeDataAsHalfPrecision_;
};
I3_POINTER_TYPEDEFS(I3CLSimFunctionFromTable);
