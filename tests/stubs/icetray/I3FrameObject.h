// stand-in (tests/stubs/README.md)
#pragma once
#include <icetray/I3PointerTypedefs.h>
#include <icetray/serialization.h>
class I3FrameObject {
public:
    virtual ~I3FrameObject() {}
};
I3_POINTER_TYPEDEFS(I3FrameObject);
