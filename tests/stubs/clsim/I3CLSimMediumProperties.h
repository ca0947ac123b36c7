// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimMediumProperties.h:54-200 and the getters implemented in
// private/clsim/I3CLSimMediumProperties.cxx:85-155 (GetMin/MaxWavelength)
#pragma once
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>
#include <icetray/I3FrameObject.h>
#include <clsim/function/I3CLSimFunction.h>
#include <clsim/function/I3CLSimScalarField.h>
#include <clsim/function/I3CLSimVectorTransform.h>
#include <clsim/random_value/I3CLSimRandomValue.h>
class I3CLSimMediumProperties : public I3FrameObject {
public:
    I3CLSimMediumProperties(double mediumDensity, uint32_t layersNum, double layersZStart, double layersHeight,
                            double rockZCoordinate, double airZCoordinate)
        : mediumDensity_(mediumDensity), layersNum_(layersNum), layersZStart_(layersZStart), layersHeight_(layersHeight),
          rockZCoordinate_(rockZCoordinate), airZCoordinate_(airZCoordinate), forcedMinWlen_(-INFINITY), forcedMaxWlen_(INFINITY),
          efficiency_(1.), absorptionLength_(layersNum), scatteringLength_(layersNum), phaseRefractiveIndex_(layersNum),
          groupRefractiveIndexOverride_(layersNum) {}
    bool IsReady() const
    {
        for (uint32_t i = 0; i < layersNum_; ++i)
            if (!absorptionLength_[i] || !scatteringLength_[i] || !phaseRefractiveIndex_[i]) return false;
        return static_cast<bool>(scatteringCosAngleDist_);
    }
    const std::vector<I3CLSimFunctionConstPtr> &GetAbsorptionLengths() const { return absorptionLength_; }
    const std::vector<I3CLSimFunctionConstPtr> &GetScatteringLengths() const { return scatteringLength_; }
    const std::vector<I3CLSimFunctionConstPtr> &GetPhaseRefractiveIndices() const { return phaseRefractiveIndex_; }
    const std::vector<I3CLSimFunctionConstPtr> &GetGroupRefractiveIndicesOverride() const { return groupRefractiveIndexOverride_; }
    I3CLSimFunctionConstPtr GetAbsorptionLength(uint32_t layer) const { return absorptionLength_.at(layer); }
    I3CLSimFunctionConstPtr GetScatteringLength(uint32_t layer) const { return scatteringLength_.at(layer); }
    I3CLSimFunctionConstPtr GetPhaseRefractiveIndex(uint32_t layer) const { return phaseRefractiveIndex_.at(layer); }
    I3CLSimFunctionConstPtr GetGroupRefractiveIndexOverride(uint32_t layer) const { return groupRefractiveIndexOverride_.at(layer); }
    I3CLSimRandomValueConstPtr GetScatteringCosAngleDistribution() const { return scatteringCosAngleDist_; }
    I3CLSimScalarFieldConstPtr GetDirectionalAbsorptionLengthCorrection() const { return directionalAbsorptionLengthCorrection_; }
    I3CLSimVectorTransformConstPtr GetPreScatterDirectionTransform() const { return preScatterDirectionTransform_; }
    I3CLSimVectorTransformConstPtr GetPostScatterDirectionTransform() const { return postScatterDirectionTransform_; }
    I3CLSimScalarFieldConstPtr GetIceTiltZShift() const { return iceTiltZShift_; }
    void SetAbsorptionLength(uint32_t layer, I3CLSimFunctionConstPtr ptr) { absorptionLength_.at(layer) = ptr; }
    void SetScatteringLength(uint32_t layer, I3CLSimFunctionConstPtr ptr) { scatteringLength_.at(layer) = ptr; }
    void SetPhaseRefractiveIndex(uint32_t layer, I3CLSimFunctionConstPtr ptr) { phaseRefractiveIndex_.at(layer) = ptr; }
    void SetGroupRefractiveIndexOverride(uint32_t layer, I3CLSimFunctionConstPtr ptr) { groupRefractiveIndexOverride_.at(layer) = ptr; }
    void SetScatteringCosAngleDistribution(I3CLSimRandomValueConstPtr ptr) { scatteringCosAngleDist_ = ptr; }
    void SetDirectionalAbsorptionLengthCorrection(I3CLSimScalarFieldConstPtr ptr) { directionalAbsorptionLengthCorrection_ = ptr; }
    void SetPreScatterDirectionTransform(I3CLSimVectorTransformConstPtr ptr) { preScatterDirectionTransform_ = ptr; }
    void SetPostScatterDirectionTransform(I3CLSimVectorTransformConstPtr ptr) { postScatterDirectionTransform_ = ptr; }
    void SetIceTiltZShift(I3CLSimScalarFieldConstPtr ptr) { iceTiltZShift_ = ptr; }
    double GetMinWavelength() const
    {
        if (!IsReady()) return NAN;
        double mini = forcedMinWlen_;
        for (uint32_t i = 0; i < layersNum_; ++i) {
            const I3CLSimFunctionConstPtr f[4] = {absorptionLength_[i], scatteringLength_[i], phaseRefractiveIndex_[i], groupRefractiveIndexOverride_[i]};
            for (int k = 0; k < 4; ++k) if (f[k] && f[k]->GetMinWlen() > mini) mini = f[k]->GetMinWlen();
        }
        return mini;
    }
    double GetMaxWavelength() const
    {
        if (!IsReady()) return NAN;
        double maxi = forcedMaxWlen_;
        for (uint32_t i = 0; i < layersNum_; ++i) {
            const I3CLSimFunctionConstPtr f[4] = {absorptionLength_[i], scatteringLength_[i], phaseRefractiveIndex_[i], groupRefractiveIndexOverride_[i]};
            for (int k = 0; k < 4; ++k) if (f[k] && f[k]->GetMaxWlen() < maxi) maxi = f[k]->GetMaxWlen();
        }
        return maxi;
    }
    double GetMediumDensity() const { return mediumDensity_; }
    uint32_t GetLayersNum() const { return layersNum_; }
    double GetLayersZStart() const { return layersZStart_; }
    double GetLayersHeight() const { return layersHeight_; }
    double GetRockZCoord() const { return rockZCoordinate_; }
    double GetAirZCoord() const { return airZCoordinate_; }
    double GetForcedMinWlen() const { return forcedMinWlen_; }
    double GetForcedMaxWlen() const { return forcedMaxWlen_; }
    void SetForcedMinWlen(double val) { forcedMinWlen_ = val; }
    void SetForcedMaxWlen(double val) { forcedMaxWlen_ = val; }
    double GetEfficiency() const { return efficiency_; }
    void SetEfficiency(double val) { efficiency_ = val; }
private:
    double mediumDensity_;
    uint32_t layersNum_;
    double layersZStart_, layersHeight_, rockZCoordinate_, airZCoordinate_, forcedMinWlen_, forcedMaxWlen_, efficiency_;
    std::vector<I3CLSimFunctionConstPtr> absorptionLength_, scatteringLength_, phaseRefractiveIndex_, groupRefractiveIndexOverride_;
    I3CLSimRandomValueConstPtr scatteringCosAngleDist_;
    I3CLSimScalarFieldConstPtr directionalAbsorptionLengthCorrection_;
    I3CLSimVectorTransformConstPtr preScatterDirectionTransform_, postScatterDirectionTransform_;
    I3CLSimScalarFieldConstPtr iceTiltZShift_;
};
I3_POINTER_TYPEDEFS(I3CLSimMediumProperties);
