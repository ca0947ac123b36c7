// The instantiations of prop_pool_kernel without STOP_PHOTONS_ON_DETECTION (prop_pool_kernel.hip: KEEP = true,
// SetStopDetectedPhotons(false) -- the reference class's default, OpenCL.cxx:86) as a translation unit of their own: compiled in
// parallel with the others and with the same code generation (Makefile: POOL_CODEGEN).
#define CLSIMHIP_POOL_KEEP_UNIT 1
#include "prop_pool_kernel.hip"
