// stand-in (tests/stubs/README.md): only the pointer type appears in the signatures the glue sees
#pragma once
#include <icetray/I3PointerTypedefs.h>
class I3RandomService;
I3_POINTER_TYPEDEFS(I3RandomService);
