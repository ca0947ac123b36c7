// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimLightSourceToStepConverter.h:53-198: the abstract interface with
// its real virtual signatures, in the reference's order
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include <boost/noncopyable.hpp>
#include <clsim/I3CLSimLightSource.h>
#include <clsim/I3CLSimMediumProperties.h>
#include <clsim/I3CLSimStep.h>
#include <clsim/function/I3CLSimFunction.h>
#include <phys-services/I3RandomService.h>
struct I3CLSimLightSourceParameterization {};          // public/clsim/I3CLSimLightSourceParameterization.h (not read by the adapter)
typedef std::vector<I3CLSimLightSourceParameterization> I3CLSimLightSourceParameterizationSeries;
class I3CLSimLightSourceToStepConverter_exception : public std::runtime_error {
public:
    I3CLSimLightSourceToStepConverter_exception(const std::string &msg) : std::runtime_error(msg) {}
};
struct I3CLSimLightSourceToStepConverter : private boost::noncopyable {
public:
    I3CLSimLightSourceToStepConverter() {}
    virtual ~I3CLSimLightSourceToStepConverter() {}
    virtual void SetBunchSizeGranularity(uint64_t num) = 0;
    virtual void SetMaxBunchSize(uint64_t num) = 0;
    virtual void SetRandomService(I3RandomServicePtr random) = 0;
    virtual void SetWlenBias(I3CLSimFunctionConstPtr wlenBias) = 0;
    virtual void SetMediumProperties(I3CLSimMediumPropertiesConstPtr mediumProperties) = 0;
    virtual void SetLightSourceParameterizationSeries(const I3CLSimLightSourceParameterizationSeries &s) { parameterizationSeries = s; }
    virtual const I3CLSimLightSourceParameterizationSeries &GetLightSourceParameterizationSeries() const { return parameterizationSeries; }
    virtual void Initialize() = 0;
    virtual bool IsInitialized() const = 0;
    virtual void EnqueueLightSource(const I3CLSimLightSource &lightSource, uint32_t identifier) = 0;
    virtual void EnqueueBarrier() = 0;
    virtual bool BarrierActive() const = 0;
    virtual bool MoreStepsAvailable() const = 0;
    virtual I3CLSimStepSeriesConstPtr GetConversionResultWithBarrierInfo(bool &barrierWasReset, double timeout = NAN) = 0;
    virtual I3CLSimStepSeriesConstPtr GetConversionResult(double timeout = NAN)
    {
        bool dummy;
        return GetConversionResultWithBarrierInfo(dummy, timeout);
    }
protected:
    I3CLSimLightSourceParameterizationSeries parameterizationSeries;
};
I3_POINTER_TYPEDEFS(I3CLSimLightSourceToStepConverter);
