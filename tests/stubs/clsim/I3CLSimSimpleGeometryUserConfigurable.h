// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimSimpleGeometryUserConfigurable.h
#pragma once
#include <clsim/I3CLSimSimpleGeometry.h>
class I3CLSimSimpleGeometryUserConfigurable : public I3CLSimSimpleGeometry {
public:
    I3CLSimSimpleGeometryUserConfigurable(double OMRadius, std::size_t numOMs)
        : OMRadius_(OMRadius), numOMs_(numOMs), stringIDs_(numOMs), domIDs_(numOMs), posX_(numOMs), posY_(numOMs), posZ_(numOMs), subdetectors_(numOMs) {}
    virtual std::size_t size() const { return numOMs_; }
    virtual double GetOMRadius() const { return OMRadius_; }
    virtual const std::vector<int32_t> &GetStringIDVector() const { return stringIDs_; }
    virtual const std::vector<uint32_t> &GetDomIDVector() const { return domIDs_; }
    virtual const std::vector<double> &GetPosXVector() const { return posX_; }
    virtual const std::vector<double> &GetPosYVector() const { return posY_; }
    virtual const std::vector<double> &GetPosZVector() const { return posZ_; }
    virtual const std::vector<std::string> &GetSubdetectorVector() const { return subdetectors_; }
    virtual int32_t GetStringID(std::size_t pos) const { return stringIDs_.at(pos); }
    virtual uint32_t GetDomID(std::size_t pos) const { return domIDs_.at(pos); }
    virtual double GetPosX(std::size_t pos) const { return posX_.at(pos); }
    virtual double GetPosY(std::size_t pos) const { return posY_.at(pos); }
    virtual double GetPosZ(std::size_t pos) const { return posZ_.at(pos); }
    virtual std::string GetSubdetector(std::size_t pos) const { return subdetectors_.at(pos); }
    virtual void SetStringID(std::size_t pos, int32_t val) { stringIDs_.at(pos) = val; }
    virtual void SetDomID(std::size_t pos, uint32_t val) { domIDs_.at(pos) = val; }
    virtual void SetPosX(std::size_t pos, double val) { posX_.at(pos) = val; }
    virtual void SetPosY(std::size_t pos, double val) { posY_.at(pos) = val; }
    virtual void SetPosZ(std::size_t pos, double val) { posZ_.at(pos) = val; }
    virtual void SetSubdetector(std::size_t pos, const std::string &val) { subdetectors_.at(pos) = val; }
private:
    double OMRadius_;
    std::size_t numOMs_;
    std::vector<int32_t> stringIDs_;
    std::vector<uint32_t> domIDs_;
    std::vector<double> posX_, posY_, posZ_;
    std::vector<std::string> subdetectors_;
};
I3_POINTER_TYPEDEFS(I3CLSimSimpleGeometryUserConfigurable);
