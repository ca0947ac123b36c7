// The instantiations of prop_kernel without STOP_PHOTONS_ON_DETECTION (prop_kernel.hip: TAB = 3, SetStopDetectedPhotons(false))
// as a translation unit of their own: compiled in parallel with the others, with the propagation kernels' code generation (Makefile:
// KERNEL_CODEGEN) and four waves per SIMD (the search that saves every DOM on the way holds a hit sink and its masks on top of the photon).
#define CLSIMHIP_TAB_UNIT 1
#define CLSIMHIP_KEEP_UNIT 1
#include "prop_kernel.hip"
