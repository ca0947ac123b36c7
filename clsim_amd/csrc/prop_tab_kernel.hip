// The TABULATE instantiations of prop_kernel (the table maker, prop_kernel.hip: TAB = 1, 2) as a translation unit of their
// own: they are compiled with the compiler's default code generation -- the table maker's limit is its fp64 atomics, and it
// loses 3 % under the settings that the propagation instantiations gain 8-12 % from (Makefile: KERNEL_CODEGEN) -- and in
// parallel with them.
#define CLSIMHIP_TAB_UNIT 1
// (threads per workgroup of the table maker's kernels; the propagation kernels' 256 unless the build says otherwise)
#ifdef CLSIMHIP_TAB_BLOCK
#define CLSIMHIP_BLOCK CLSIMHIP_TAB_BLOCK
#endif
#include "prop_kernel.hip"
