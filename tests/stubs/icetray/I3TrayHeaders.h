// stand-in (tests/stubs/README.md) for icetray/I3TrayHeaders.h: what the clsim public headers rely on it for
#pragma once
#include <cstddef>
#include <stdint.h>
#include <iosfwd>
#include <cmath>
#include <icetray/I3Logging.h>
#include <icetray/I3PointerTypedefs.h>
#include <icetray/I3FrameObject.h>
#include <icetray/serialization.h>
