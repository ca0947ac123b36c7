// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimPhotonHistory.h:39-70
#pragma once
#include <cstddef>
#include <vector>
#include <dataclasses/I3Vector.h>
struct I3CLSimPhotonHistory {
    I3CLSimPhotonHistory() {}
    std::size_t size() const { return posX_.size(); }
    float GetX(std::size_t i) const { return posX_[i]; }
    float GetY(std::size_t i) const { return posY_[i]; }
    float GetZ(std::size_t i) const { return posZ_[i]; }
    float GetDistanceInAbsorptionLengths(std::size_t i) const { return distanceInAbsorptionLengths_[i]; }
    void push_back(float x, float y, float z, float abslens)
    {
        posX_.push_back(x); posY_.push_back(y); posZ_.push_back(z); distanceInAbsorptionLengths_.push_back(abslens);
    }
private:
    std::vector<float> posX_, posY_, posZ_, distanceInAbsorptionLengths_;
};
typedef I3Vector<I3CLSimPhotonHistory> I3CLSimPhotonHistorySeries;
I3_POINTER_TYPEDEFS(I3CLSimPhotonHistory);
I3_POINTER_TYPEDEFS(I3CLSimPhotonHistorySeries);
