// TEST SCAFFOLDING (tests/stubs/README.md), build container only.
//
// tests/test_reference_headers.py compiles clsim_amd/cxx/icetray_adapter_test.cxx with -I/root/reference/public FIRST
// on the include path, so that every <clsim/...> header the adapter and its glue include is the reference's own.  The
// reference's class implementations (private/clsim/**/*.cxx) need boost and icetray's archive library and cannot be
// built here; this file gives the members that those headers declare out of line the smallest definition that lets the
// test link: constructors that store their arguments in the members the real headers name (so a wrong member name or
// type is a compile error here), empty destructors, and virtuals the glue never calls returning NaN / "" / false.
// No reference arithmetic is restated here.  It is compiled ONLY against the real headers.
#include <cmath>
#include <limits>
#include <ostream>
#include <stdexcept>

#include <clsim/I3CLSimStep.h>
#include <clsim/I3CLSimPhoton.h>
#include <clsim/I3CLSimPhotonHistory.h>
#include <clsim/I3CLSimMediumProperties.h>
#include <clsim/I3CLSimSimpleGeometryUserConfigurable.h>
#include <clsim/I3CLSimLightSource.h>
#include <clsim/I3CLSimFlasherPulse.h>
#include <clsim/I3CLSimLightSourceToStepConverter.h>
#include <clsim/function/I3CLSimFunctionConstant.h>
#include <clsim/function/I3CLSimFunctionFromTable.h>
#include <clsim/function/I3CLSimFunctionAbsLenIceCube.h>
#include <clsim/function/I3CLSimFunctionScatLenIceCube.h>
#include <clsim/function/I3CLSimFunctionRefIndexIceCube.h>
#include <clsim/function/I3CLSimScalarFieldConstant.h>
#include <clsim/function/I3CLSimScalarFieldIceTiltZShift.h>
#include <clsim/function/I3CLSimScalarFieldAnisotropyAbsLenScaling.h>
#include <clsim/function/I3CLSimVectorTransformConstant.h>
#include <clsim/function/I3CLSimVectorTransformMatrix.h>
#include <clsim/random_value/I3CLSimRandomValueConstant.h>
#include <clsim/random_value/I3CLSimRandomValueMixed.h>
#include <clsim/random_value/I3CLSimRandomValueHenyeyGreenstein.h>
#include <clsim/random_value/I3CLSimRandomValueSimplifiedLiu.h>
#include <clsim/random_value/I3CLSimRandomValueInterpolatedDistribution.h>
#include <clsim/random_value/I3CLSimRandomValueWlenCherenkovNoDispersion.h>

// ---- records ----
I3CLSimStep::~I3CLSimStep() {}
I3CLSimPhoton::~I3CLSimPhoton() {}
I3CLSimPhotonHistory::~I3CLSimPhotonHistory() {}

// ---- wavelength functions ----
#define FN_VIRTUALS(C)                                                                          \
    C::~C() {}                                                                                  \
    double C::GetValue(double) const { return NAN; }                                            \
    std::string C::GetOpenCLFunction(const std::string &) const { return std::string(); }      \
    bool C::CompareTo(const I3CLSimFunction &other) const { return this == &other; }

I3CLSimFunction::I3CLSimFunction() {}
I3CLSimFunction::~I3CLSimFunction() {}

I3CLSimFunctionConstant::I3CLSimFunctionConstant(double value) : value_(value) {}
FN_VIRTUALS(I3CLSimFunctionConstant)
bool I3CLSimFunctionConstant::HasNativeImplementation() const { return true; }
bool I3CLSimFunctionConstant::HasDerivative() const { return true; }
double I3CLSimFunctionConstant::GetDerivative(double) const { return 0.; }
double I3CLSimFunctionConstant::GetMinWlen() const { return -std::numeric_limits<double>::infinity(); }
double I3CLSimFunctionConstant::GetMaxWlen() const { return std::numeric_limits<double>::infinity(); }
std::string I3CLSimFunctionConstant::GetOpenCLFunctionDerivative(const std::string &) const { return std::string(); }

const bool I3CLSimFunctionFromTable::default_storeDataAsHalfPrecision = false;
I3CLSimFunctionFromTable::I3CLSimFunctionFromTable(const std::vector<double> &wlens, const std::vector<double> &values, bool storeDataAsHalfPrecision)
    : startWlen_(NAN), wlenStep_(NAN), wlens_(wlens), values_(values), equalSpacingMode_(false), storeDataAsHalfPrecision_(storeDataAsHalfPrecision) {}
I3CLSimFunctionFromTable::I3CLSimFunctionFromTable(double startWlen, double wlenStep, const std::vector<double> &values, bool storeDataAsHalfPrecision)
    : startWlen_(startWlen), wlenStep_(wlenStep), values_(values), equalSpacingMode_(true), storeDataAsHalfPrecision_(storeDataAsHalfPrecision)
{
    if (values_.size() < 2) throw std::runtime_error("values must contain at least 2 elements!");
}
FN_VIRTUALS(I3CLSimFunctionFromTable)
double I3CLSimFunctionFromTable::GetMinWlen() const { return equalSpacingMode_ ? startWlen_ : wlens_.front(); }
double I3CLSimFunctionFromTable::GetMaxWlen() const
{
    return equalSpacingMode_ ? startWlen_ + wlenStep_ * static_cast<double>(values_.size() - 1) : wlens_.back();
}

I3CLSimFunctionAbsLenIceCube::I3CLSimFunctionAbsLenIceCube(double kappa, double A, double B, double D, double E, double aDust400, double deltaTau)
    : kappa_(kappa), A_(A), B_(B), D_(D), E_(E), aDust400_(aDust400), deltaTau_(deltaTau) {}
FN_VIRTUALS(I3CLSimFunctionAbsLenIceCube)

I3CLSimFunctionScatLenIceCube::I3CLSimFunctionScatLenIceCube(double alpha, double b400) : alpha_(alpha), b400_(b400) {}
FN_VIRTUALS(I3CLSimFunctionScatLenIceCube)

const std::string I3CLSimFunctionRefIndexIceCube::default_mode = "phase";
const double I3CLSimFunctionRefIndexIceCube::default_n0 = NAN, I3CLSimFunctionRefIndexIceCube::default_n1 = NAN,
             I3CLSimFunctionRefIndexIceCube::default_n2 = NAN, I3CLSimFunctionRefIndexIceCube::default_n3 = NAN,
             I3CLSimFunctionRefIndexIceCube::default_n4 = NAN, I3CLSimFunctionRefIndexIceCube::default_g0 = NAN,
             I3CLSimFunctionRefIndexIceCube::default_g1 = NAN, I3CLSimFunctionRefIndexIceCube::default_g2 = NAN,
             I3CLSimFunctionRefIndexIceCube::default_g3 = NAN, I3CLSimFunctionRefIndexIceCube::default_g4 = NAN;
I3CLSimFunctionRefIndexIceCube::I3CLSimFunctionRefIndexIceCube(std::string mode, double n0, double n1, double n2, double n3, double n4,
                                                               double g0, double g1, double g2, double g3, double g4)
    : mode_(mode), n0_(n0), n1_(n1), n2_(n2), n3_(n3), n4_(n4), g0_(g0), g1_(g1), g2_(g2), g3_(g3), g4_(g4) {}
FN_VIRTUALS(I3CLSimFunctionRefIndexIceCube)
double I3CLSimFunctionRefIndexIceCube::GetDerivative(double) const { return NAN; }
std::string I3CLSimFunctionRefIndexIceCube::GetOpenCLFunctionDerivative(const std::string &) const { return std::string(); }

// ---- scalar fields ----
#define FIELD_VIRTUALS(C)                                                                       \
    C::~C() {}                                                                                  \
    bool C::HasNativeImplementation() const { return true; }                                    \
    double C::GetValue(double, double, double) const { return NAN; }                            \
    std::string C::GetOpenCLFunction(const std::string &) const { return std::string(); }      \
    bool C::CompareTo(const I3CLSimScalarField &other) const { return this == &other; }

I3CLSimScalarField::I3CLSimScalarField() {}
I3CLSimScalarField::~I3CLSimScalarField() {}

I3CLSimScalarFieldConstant::I3CLSimScalarFieldConstant(double value) : value_(value) {}
FIELD_VIRTUALS(I3CLSimScalarFieldConstant)

const double I3CLSimScalarFieldIceTiltZShift::default_directionOfTiltAzimuth = NAN;
I3CLSimScalarFieldIceTiltZShift::I3CLSimScalarFieldIceTiltZShift(const std::vector<double> &distancesFromOriginAlongTilt,
                                                                 const std::vector<double> &zCoordinates, const I3Matrix &zCorrections,
                                                                 double directionOfTiltAzimuth)
    : distancesFromOriginAlongTilt_(distancesFromOriginAlongTilt), zCoordinates_(zCoordinates), zCorrections_(zCorrections),
      directionOfTiltAzimuth_(directionOfTiltAzimuth), firstZCoordinate_(zCoordinates.empty() ? NAN : zCoordinates[0]),
      zCoordinateSpacing_(zCoordinates.size() > 1 ? zCoordinates[1] - zCoordinates[0] : NAN) {}
FIELD_VIRTUALS(I3CLSimScalarFieldIceTiltZShift)

const double I3CLSimScalarFieldAnisotropyAbsLenScaling::default_anisotropyDirAzimuth = NAN;
const double I3CLSimScalarFieldAnisotropyAbsLenScaling::default_magnitudeAlongDir = NAN;
const double I3CLSimScalarFieldAnisotropyAbsLenScaling::default_magnitudePerpToDir = NAN;
I3CLSimScalarFieldAnisotropyAbsLenScaling::I3CLSimScalarFieldAnisotropyAbsLenScaling(double anisotropyDirAzimuth, double magnitudeAlongDir,
                                                                                     double magnitudePerpToDir)
    : anisotropyDirAzimuth_(anisotropyDirAzimuth), magnitudeAlongDir_(magnitudeAlongDir), magnitudePerpToDir_(magnitudePerpToDir) {}
FIELD_VIRTUALS(I3CLSimScalarFieldAnisotropyAbsLenScaling)

// ---- direction transforms ----
#define XFORM_VIRTUALS(C)                                                                                   \
    C::~C() {}                                                                                              \
    bool C::HasNativeImplementation() const { return true; }                                                \
    std::vector<double> C::ApplyTransform(const std::vector<double> &vec) const { return vec; }             \
    std::string C::GetOpenCLFunction(const std::string &) const { return std::string(); }                  \
    bool C::CompareTo(const I3CLSimVectorTransform &other) const { return this == &other; }

I3CLSimVectorTransform::I3CLSimVectorTransform() {}
I3CLSimVectorTransform::~I3CLSimVectorTransform() {}
I3CLSimVectorTransformConstant::I3CLSimVectorTransformConstant() {}
XFORM_VIRTUALS(I3CLSimVectorTransformConstant)
I3CLSimVectorTransformMatrix::I3CLSimVectorTransformMatrix(const I3Matrix &matrix, bool renormalize) : matrix_(matrix), renormalize_(renormalize) {}
XFORM_VIRTUALS(I3CLSimVectorTransformMatrix)

// ---- random values ----
#define RV_VIRTUALS(C)                                                                                                      \
    C::~C() {}                                                                                                              \
    std::size_t C::NumberOfParameters() const { return 0; }                                                                 \
    double C::SampleFromDistribution(const I3RandomServicePtr &, const std::vector<double> &) const { return NAN; }         \
    std::string C::GetOpenCLFunction(const std::string &, const std::string &, const std::string &, const std::string &,    \
                                     const std::string &) const { return std::string(); }                                  \
    bool C::CompareTo(const I3CLSimRandomValue &other) const { return this == &other; }

I3CLSimRandomValue::I3CLSimRandomValue() {}
I3CLSimRandomValue::~I3CLSimRandomValue() {}

I3CLSimRandomValueConstant::I3CLSimRandomValueConstant() : value_(NAN) {}
I3CLSimRandomValueConstant::I3CLSimRandomValueConstant(double value) : value_(value) {}
RV_VIRTUALS(I3CLSimRandomValueConstant)

I3CLSimRandomValueMixed::I3CLSimRandomValueMixed(double fractionOfFirstDistribution, I3CLSimRandomValueConstPtr firstDistribution,
                                                 I3CLSimRandomValueConstPtr secondDistribution)
    : fractionOfFirstDistribution_(fractionOfFirstDistribution), firstDistribution_(firstDistribution), secondDistribution_(secondDistribution) {}
RV_VIRTUALS(I3CLSimRandomValueMixed)
bool I3CLSimRandomValueMixed::OpenCLFunctionWillOnlyUseASingleRandomNumber() const { return true; }

I3CLSimRandomValueHenyeyGreenstein::I3CLSimRandomValueHenyeyGreenstein(double meanCosine) : meanCosine_(meanCosine) {}
RV_VIRTUALS(I3CLSimRandomValueHenyeyGreenstein)
I3CLSimRandomValueSimplifiedLiu::I3CLSimRandomValueSimplifiedLiu(double meanCosine) : meanCosine_(meanCosine) {}
RV_VIRTUALS(I3CLSimRandomValueSimplifiedLiu)

I3CLSimRandomValueInterpolatedDistribution::I3CLSimRandomValueInterpolatedDistribution(const std::vector<double> &x, const std::vector<double> &y)
    : x_(x), y_(y), constantXSpacing_(NAN), firstX_(NAN) {}
I3CLSimRandomValueInterpolatedDistribution::I3CLSimRandomValueInterpolatedDistribution(double xFirst, double xSpacing, const std::vector<double> &y)
    : y_(y), constantXSpacing_(xSpacing), firstX_(xFirst) {}
RV_VIRTUALS(I3CLSimRandomValueInterpolatedDistribution)

I3CLSimRandomValueWlenCherenkovNoDispersion::I3CLSimRandomValueWlenCherenkovNoDispersion(double fromWlen, double toWlen)
    : fromWlen_(fromWlen), toWlen_(toWlen) {}
RV_VIRTUALS(I3CLSimRandomValueWlenCherenkovNoDispersion)

// ---- medium properties (setters / getters as private/clsim/I3CLSimMediumProperties.cxx:85-260 behaves: one slot per layer) ----
const double I3CLSimMediumProperties::default_mediumDensity = NAN;
const uint32_t I3CLSimMediumProperties::default_layersNum = 1;
const double I3CLSimMediumProperties::default_layersZStart = NAN;
const double I3CLSimMediumProperties::default_layersHeight = NAN;
const double I3CLSimMediumProperties::default_rockZCoordinate = NAN;
const double I3CLSimMediumProperties::default_airZCoordinate = NAN;
I3CLSimMediumProperties::I3CLSimMediumProperties(double mediumDensity, uint32_t layersNum, double layersZStart, double layersHeight,
                                                 double rockZCoordinate, double airZCoordinate)
    : mediumDensity_(mediumDensity), layersNum_(layersNum), layersZStart_(layersZStart), layersHeight_(layersHeight),
      rockZCoordinate_(rockZCoordinate), airZCoordinate_(airZCoordinate), forcedMinWlen_(-std::numeric_limits<double>::infinity()),
      forcedMaxWlen_(std::numeric_limits<double>::infinity()), absorptionLength_(layersNum), scatteringLength_(layersNum),
      phaseRefractiveIndex_(layersNum), groupRefractiveIndexOverride_(layersNum) {}
I3CLSimMediumProperties::~I3CLSimMediumProperties() {}
bool I3CLSimMediumProperties::IsReady() const
{
    for (uint32_t i = 0; i < layersNum_; ++i)
        if (!absorptionLength_[i] || !scatteringLength_[i] || !phaseRefractiveIndex_[i]) return false;
    return static_cast<bool>(scatteringCosAngleDist_);
}
const std::vector<I3CLSimFunctionConstPtr> &I3CLSimMediumProperties::GetAbsorptionLengths() const { return absorptionLength_; }
const std::vector<I3CLSimFunctionConstPtr> &I3CLSimMediumProperties::GetScatteringLengths() const { return scatteringLength_; }
const std::vector<I3CLSimFunctionConstPtr> &I3CLSimMediumProperties::GetPhaseRefractiveIndices() const { return phaseRefractiveIndex_; }
const std::vector<I3CLSimFunctionConstPtr> &I3CLSimMediumProperties::GetGroupRefractiveIndicesOverride() const { return groupRefractiveIndexOverride_; }
I3CLSimFunctionConstPtr I3CLSimMediumProperties::GetAbsorptionLength(uint32_t layer) const { return absorptionLength_.at(layer); }
I3CLSimFunctionConstPtr I3CLSimMediumProperties::GetScatteringLength(uint32_t layer) const { return scatteringLength_.at(layer); }
I3CLSimFunctionConstPtr I3CLSimMediumProperties::GetPhaseRefractiveIndex(uint32_t layer) const { return phaseRefractiveIndex_.at(layer); }
I3CLSimFunctionConstPtr I3CLSimMediumProperties::GetGroupRefractiveIndexOverride(uint32_t layer) const { return groupRefractiveIndexOverride_.at(layer); }
I3CLSimRandomValueConstPtr I3CLSimMediumProperties::GetScatteringCosAngleDistribution() const { return scatteringCosAngleDist_; }
I3CLSimScalarFieldConstPtr I3CLSimMediumProperties::GetDirectionalAbsorptionLengthCorrection() const { return directionalAbsorptionLengthCorrection_; }
I3CLSimVectorTransformConstPtr I3CLSimMediumProperties::GetPreScatterDirectionTransform() const { return preScatterDirectionTransform_; }
I3CLSimVectorTransformConstPtr I3CLSimMediumProperties::GetPostScatterDirectionTransform() const { return postScatterDirectionTransform_; }
I3CLSimScalarFieldConstPtr I3CLSimMediumProperties::GetIceTiltZShift() const { return iceTiltZShift_; }
void I3CLSimMediumProperties::SetAbsorptionLength(uint32_t layer, I3CLSimFunctionConstPtr ptr) { absorptionLength_.at(layer) = ptr; }
void I3CLSimMediumProperties::SetScatteringLength(uint32_t layer, I3CLSimFunctionConstPtr ptr) { scatteringLength_.at(layer) = ptr; }
void I3CLSimMediumProperties::SetPhaseRefractiveIndex(uint32_t layer, I3CLSimFunctionConstPtr ptr) { phaseRefractiveIndex_.at(layer) = ptr; }
void I3CLSimMediumProperties::SetGroupRefractiveIndexOverride(uint32_t layer, I3CLSimFunctionConstPtr ptr) { groupRefractiveIndexOverride_.at(layer) = ptr; }
void I3CLSimMediumProperties::SetScatteringCosAngleDistribution(I3CLSimRandomValueConstPtr ptr) { scatteringCosAngleDist_ = ptr; }
void I3CLSimMediumProperties::SetDirectionalAbsorptionLengthCorrection(I3CLSimScalarFieldConstPtr ptr) { directionalAbsorptionLengthCorrection_ = ptr; }
void I3CLSimMediumProperties::SetPreScatterDirectionTransform(I3CLSimVectorTransformConstPtr ptr) { preScatterDirectionTransform_ = ptr; }
void I3CLSimMediumProperties::SetPostScatterDirectionTransform(I3CLSimVectorTransformConstPtr ptr) { postScatterDirectionTransform_ = ptr; }
void I3CLSimMediumProperties::SetIceTiltZShift(I3CLSimScalarFieldConstPtr ptr) { iceTiltZShift_ = ptr; }
double I3CLSimMediumProperties::GetMinWavelength() const
{
    if (!IsReady()) return NAN;
    double mini = forcedMinWlen_;
    for (uint32_t i = 0; i < layersNum_; ++i) {
        const I3CLSimFunctionConstPtr f[4] = {absorptionLength_[i], scatteringLength_[i], phaseRefractiveIndex_[i], groupRefractiveIndexOverride_[i]};
        for (int k = 0; k < 4; ++k) if (f[k] && f[k]->GetMinWlen() > mini) mini = f[k]->GetMinWlen();
    }
    return mini;
}
double I3CLSimMediumProperties::GetMaxWavelength() const
{
    if (!IsReady()) return NAN;
    double maxi = forcedMaxWlen_;
    for (uint32_t i = 0; i < layersNum_; ++i) {
        const I3CLSimFunctionConstPtr f[4] = {absorptionLength_[i], scatteringLength_[i], phaseRefractiveIndex_[i], groupRefractiveIndexOverride_[i]};
        for (int k = 0; k < 4; ++k) if (f[k] && f[k]->GetMaxWlen() < maxi) maxi = f[k]->GetMaxWlen();
    }
    return maxi;
}

// ---- geometry ----
I3CLSimSimpleGeometryUserConfigurable::I3CLSimSimpleGeometryUserConfigurable(double OMRadius, std::size_t numOMs)
    : OMRadius_(OMRadius), numOMs_(numOMs), stringIDs_(numOMs, 0), domIDs_(numOMs, 0), posX_(numOMs, NAN), posY_(numOMs, NAN),
      posZ_(numOMs, NAN), subdetectors_(numOMs, "") {}
I3CLSimSimpleGeometryUserConfigurable::~I3CLSimSimpleGeometryUserConfigurable() {}

// ---- light sources ----
I3CLSimFlasherPulse::I3CLSimFlasherPulse()
    : flasherPulseType_(Unknown), time_(NAN), numberOfPhotonsNoBias_(NAN), pulseWidth_(NAN), angularEmissionSigmaPolar_(NAN),
      angularEmissionSigmaAzimuthal_(NAN) {}
I3CLSimFlasherPulse::~I3CLSimFlasherPulse() {}
I3CLSimLightSource::I3CLSimLightSource(const I3Particle &particle) : lightSourceType_(Particle), particle_(particle) {}
I3CLSimLightSource::I3CLSimLightSource(const I3CLSimFlasherPulse &flasher) : lightSourceType_(Flasher), flasher_(flasher) {}
I3CLSimLightSource::I3CLSimLightSource(const I3CLSimLightSource &o) : lightSourceType_(o.lightSourceType_), particle_(o.particle_), flasher_(o.flasher_) {}
I3CLSimLightSource::~I3CLSimLightSource() {}
const I3Particle &I3CLSimLightSource::GetParticle() const
{
    if (lightSourceType_ != Particle) throw std::runtime_error("light source is not a particle");
    return particle_;
}
const I3CLSimFlasherPulse &I3CLSimLightSource::GetFlasherPulse() const
{
    if (lightSourceType_ != Flasher) throw std::runtime_error("light source is not a flasher pulse");
    return flasher_;
}

// ---- producer-side interface: its three non-pure members ----
I3CLSimLightSourceToStepConverter::I3CLSimLightSourceToStepConverter() {}
I3CLSimLightSourceToStepConverter::~I3CLSimLightSourceToStepConverter() {}
void I3CLSimLightSourceToStepConverter::SetLightSourceParameterizationSeries(const I3CLSimLightSourceParameterizationSeries &parameterizationSeries_)
{
    parameterizationSeries = parameterizationSeries_;
}
const I3CLSimLightSourceParameterizationSeries &I3CLSimLightSourceToStepConverter::GetLightSourceParameterizationSeries() const
{
    return parameterizationSeries;
}
I3CLSimStepSeriesConstPtr I3CLSimLightSourceToStepConverter::GetConversionResult(double timeout)
{
    bool dummy;
    return GetConversionResultWithBarrierInfo(dummy, timeout);
}
I3CLSimLightSourceParameterization::I3CLSimLightSourceParameterization()
    : forParticleType(I3Particle::unknown), fromEnergy(0.), toEnergy(0.), needsLength(false), catchAll(false), flasherMode(false),
      forFlasherPulseType(I3CLSimFlasherPulse::Unknown) {}
I3CLSimLightSourceParameterization::~I3CLSimLightSourceParameterization() {}
