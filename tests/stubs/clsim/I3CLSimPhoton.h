// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimPhoton.h:60-213: the 80-byte photon record
#pragma once
#include <cstdint>
#include <dataclasses/I3Vector.h>
struct I3CLSimPhoton {
    float GetPosX() const { return posAndTime[0]; }
    float GetPosY() const { return posAndTime[1]; }
    float GetPosZ() const { return posAndTime[2]; }
    float GetTime() const { return posAndTime[3]; }
    float GetDirTheta() const { return dir[0]; }
    float GetDirPhi() const { return dir[1]; }
    float GetWavelength() const { return wavelength; }
    float GetCherenkovDist() const { return cherenkovDist; }
    uint32_t GetNumScatters() const { return numScatters; }
    float GetWeight() const { return weight; }
    uint32_t GetID() const { return identifier; }
    int16_t GetStringID() const { return stringID; }
    uint16_t GetOMID() const { return omID; }
    float GetStartPosX() const { return startPosAndTime[0]; }
    float GetStartPosY() const { return startPosAndTime[1]; }
    float GetStartPosZ() const { return startPosAndTime[2]; }
    float GetStartTime() const { return startPosAndTime[3]; }
    float GetStartDirTheta() const { return startDir[0]; }
    float GetStartDirPhi() const { return startDir[1]; }
    float GetGroupVelocity() const { return groupVelocity; }
    float GetDistInAbsLens() const { return distInAbsLens; }
    float posAndTime[4];
    float dir[2];
    float wavelength, cherenkovDist;
    uint32_t numScatters;
    float weight;
    uint32_t identifier;
    int16_t stringID;
    uint16_t omID;
    float startPosAndTime[4];
    float startDir[2];
    float groupVelocity, distInAbsLens;
};
static_assert(sizeof(I3CLSimPhoton) == 80, "I3CLSimPhoton is an 80-byte blob");
typedef I3Vector<I3CLSimPhoton> I3CLSimPhotonSeries;
I3_POINTER_TYPEDEFS(I3CLSimPhoton);
I3_POINTER_TYPEDEFS(I3CLSimPhotonSeries);
