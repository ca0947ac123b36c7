// stand-in (tests/stubs/README.md): phys-services/I3RandomService.h -- the one member the converter uses to seed its
// streams (private/opencl/mwcrng_init.h:108-112 calls Integer(0xffffffff))
#pragma once
#include <icetray/I3PointerTypedefs.h>
class I3RandomService {
public:
    virtual ~I3RandomService() {}
    virtual unsigned int Integer(unsigned int imax) = 0;
};
I3_POINTER_TYPEDEFS(I3RandomService);
