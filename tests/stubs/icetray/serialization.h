// stand-in (tests/stubs/README.md) for icetray/serialization.h: the names the clsim headers mention in declarations
// (friend access class, archive classes, versioning / split-member macros).  Nothing is serialised in these tests.
#pragma once
#include <icetray/I3PointerTypedefs.h>
namespace icecube {
namespace serialization {
class access;
template <class T> struct nvp;
}
namespace archive {
class portable_binary_iarchive;
class portable_binary_oarchive;
class xml_iarchive;
class xml_oarchive;
}
}
#define I3_SERIALIZATION_SPLIT_MEMBER() template <class Archive> void serialize(Archive &ar, unsigned version)
#define I3_CLASS_VERSION(C, v) static_assert(sizeof(C) > 0 && (v) >= 0, "class version")
#define I3_SERIALIZABLE(C)
#define I3_SPLIT_SERIALIZABLE(C)
