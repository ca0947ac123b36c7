// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimStep.h:68-155: the 48-byte step record and the accessors
// the adapter test uses.  Field order = the reference's (posAndTime, dirAndLengthAndBeta, numPhotons, weight,
// identifier, sourceType, dummy1, dummy2).
#pragma once
#include <cstdint>
#include <dataclasses/I3Vector.h>
struct I3CLSimStep {
    float GetPosX() const { return posAndTime[0]; }
    float GetPosY() const { return posAndTime[1]; }
    float GetPosZ() const { return posAndTime[2]; }
    float GetTime() const { return posAndTime[3]; }
    float GetDirTheta() const { return dirAndLengthAndBeta[0]; }
    float GetDirPhi() const { return dirAndLengthAndBeta[1]; }
    float GetLength() const { return dirAndLengthAndBeta[2]; }
    float GetBeta() const { return dirAndLengthAndBeta[3]; }
    uint32_t GetNumPhotons() const { return numPhotons; }
    float GetWeight() const { return weight; }
    uint32_t GetID() const { return identifier; }
    uint8_t GetSourceType() const { return sourceType; }
    void SetPosX(const float &v) { posAndTime[0] = v; }
    void SetPosY(const float &v) { posAndTime[1] = v; }
    void SetPosZ(const float &v) { posAndTime[2] = v; }
    void SetTime(const float &v) { posAndTime[3] = v; }
    void SetDirTheta(const float &v) { dirAndLengthAndBeta[0] = v; }
    void SetDirPhi(const float &v) { dirAndLengthAndBeta[1] = v; }
    void SetLength(const float &v) { dirAndLengthAndBeta[2] = v; }
    void SetBeta(const float &v) { dirAndLengthAndBeta[3] = v; }
    void SetNumPhotons(const uint32_t &v) { numPhotons = v; }
    void SetWeight(const float &v) { weight = v; }
    void SetID(const uint32_t &v) { identifier = v; }
    void SetSourceType(const uint8_t &v) { sourceType = v; }
    float posAndTime[4];
    float dirAndLengthAndBeta[4];
    uint32_t numPhotons;
    float weight;
    uint32_t identifier;
    uint8_t sourceType, dummy1;
    uint16_t dummy2;
};
static_assert(sizeof(I3CLSimStep) == 48, "I3CLSimStep is a 48-byte blob");
typedef I3Vector<I3CLSimStep> I3CLSimStepSeries;
I3_POINTER_TYPEDEFS(I3CLSimStep);
I3_POINTER_TYPEDEFS(I3CLSimStepSeries);
