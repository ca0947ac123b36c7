// stand-in (tests/stubs/README.md) for public/clsim/I3CLSimSimpleGeometry.h:39-63 (abstract) and
// I3CLSimSimpleGeometryUserConfigurable.h (a concrete geometry the test fills)
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <icetray/I3PointerTypedefs.h>
class I3CLSimSimpleGeometry {
public:
    // no virtual destructor in the reference (I3CLSimSimpleGeometry.h:44-63): geometries live in shared_ptrs of the concrete type
    virtual std::size_t size() const = 0;
    virtual double GetOMRadius() const = 0;
    virtual const std::vector<int32_t> &GetStringIDVector() const = 0;
    virtual const std::vector<uint32_t> &GetDomIDVector() const = 0;
    virtual const std::vector<double> &GetPosXVector() const = 0;
    virtual const std::vector<double> &GetPosYVector() const = 0;
    virtual const std::vector<double> &GetPosZVector() const = 0;
    virtual const std::vector<std::string> &GetSubdetectorVector() const = 0;
    virtual int32_t GetStringID(std::size_t pos) const = 0;
    virtual uint32_t GetDomID(std::size_t pos) const = 0;
    virtual double GetPosX(std::size_t pos) const = 0;
    virtual double GetPosY(std::size_t pos) const = 0;
    virtual double GetPosZ(std::size_t pos) const = 0;
    virtual std::string GetSubdetector(std::size_t pos) const = 0;
};
I3_POINTER_TYPEDEFS(I3CLSimSimpleGeometry);
