// Read access to private data members of classes whose headers must stay untouched.
//
// Most of clsim's configuration classes keep their parameters private and offer no getters (at the reference revision:
// I3CLSimFunctionRefIndexIceCube, I3CLSimScalarFieldIceTiltZShift, I3CLSimScalarFieldAnisotropyAbsLenScaling,
// I3CLSimVectorTransformMatrix, I3CLSimRandomValueMixed / HenyeyGreenstein / SimplifiedLiu / InterpolatedDistribution /
// Constant / WlenCherenkovNoDispersion, I3CLSimFunctionConstant, I3CLSimScalarFieldConstant); their only public output
// is generated OpenCL source text.  A drop-in converter that does not patch those headers needs the numbers anyway.
// C++ allows naming a private member in an explicit template instantiation ([temp.spec]/6: "the usual access checking
// rules do not apply to names used to specify explicit instantiations"), which is what this uses:
//
//     CLSIMHIP_PRIVATE_MEMBER(n0, I3CLSimFunctionRefIndexIceCube, double, n0_)     // at namespace scope
//     double v = obj.*member(clsimhip_private::n0());          // the friend is found by argument-dependent lookup
//
// The member NAMES are those of the reference headers (cited next to each use in the glue); a maintainer who prefers
// getters adds them upstream and defines CLSIMHIP_HAVE_PARAMETER_GETTERS (INTEGRATION.md lists them).
#pragma once

namespace clsimhip_private {
template <class Tag, typename Tag::type Member>
struct Bind {
    friend typename Tag::type member(Tag) { return Member; }
};
}

#define CLSIMHIP_PRIVATE_MEMBER(tag, Class, Type, name)                         \
    namespace clsimhip_private {                                                \
    struct tag {                                                                \
        typedef Type Class::*type;                                              \
        friend type member(tag);                                                \
    };                                                                          \
    template struct Bind<tag, &Class::name>;                                    \
    }
