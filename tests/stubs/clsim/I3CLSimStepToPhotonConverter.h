// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimStepToPhotonConverter.h:57-192: the abstract interface
// with its real virtual signatures, in the reference's order
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include <boost/noncopyable.hpp>
#include <icetray/I3PointerTypedefs.h>
#include <clsim/I3CLSimStep.h>
#include <clsim/I3CLSimPhoton.h>
#include <clsim/I3CLSimPhotonHistory.h>
#include <clsim/I3CLSimMediumProperties.h>
#include <clsim/I3CLSimSimpleGeometry.h>
#include <clsim/random_value/I3CLSimRandomValue.h>
#include <clsim/function/I3CLSimFunction.h>

class I3CLSimStepToPhotonConverter_exception : public std::runtime_error {
public:
    virtual ~I3CLSimStepToPhotonConverter_exception() throw() {}
    I3CLSimStepToPhotonConverter_exception(const std::string &msg) : std::runtime_error(msg) {}
};

struct I3CLSimStepToPhotonConverter : private boost::noncopyable {
public:
    struct ConversionResult_t {
        ConversionResult_t() : identifier(0) {}
        ConversionResult_t(uint32_t identifier_, I3CLSimPhotonSeriesPtr photons_ = I3CLSimPhotonSeriesPtr(),
                           I3CLSimPhotonHistorySeriesPtr photonHistories_ = I3CLSimPhotonHistorySeriesPtr())
            : identifier(identifier_), photons(photons_), photonHistories(photonHistories_) {}
        uint32_t identifier;
        I3CLSimPhotonSeriesPtr photons;
        I3CLSimPhotonHistorySeriesPtr photonHistories;
    };
    // no virtual destructor: the reference has it commented out (I3CLSimStepToPhotonConverter.h:88)
    virtual void SetWlenGenerators(const std::vector<I3CLSimRandomValueConstPtr> &wlenGenerators) = 0;
    virtual void SetWlenBias(I3CLSimFunctionConstPtr wlenBias) = 0;
    virtual void SetMediumProperties(I3CLSimMediumPropertiesConstPtr mediumProperties) = 0;
    virtual void SetGeometry(I3CLSimSimpleGeometryConstPtr geometry) = 0;
    virtual void Initialize() = 0;
    virtual bool IsInitialized() const = 0;
    virtual void EnqueueSteps(I3CLSimStepSeriesConstPtr steps, uint32_t identifier) = 0;
    virtual std::size_t GetWorkgroupSize() const = 0;
    virtual std::size_t GetMaxNumWorkitems() const = 0;
    virtual std::size_t QueueSize() const = 0;
    virtual bool MorePhotonsAvailable() const = 0;
    virtual ConversionResult_t GetConversionResult() = 0;
    virtual std::map<std::string, double> GetStatistics() const { return std::map<std::string, double>(); }
};
I3_POINTER_TYPEDEFS(I3CLSimStepToPhotonConverter);
