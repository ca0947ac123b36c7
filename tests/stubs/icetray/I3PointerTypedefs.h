// stand-in (tests/stubs/README.md): icetray/I3PointerTypedefs.h
#pragma once
#include <boost/shared_ptr.hpp>
#define I3_POINTER_TYPEDEFS(C) typedef boost::shared_ptr<C> C##Ptr; typedef boost::shared_ptr<const C> C##ConstPtr
#define I3_FORWARD_DECLARATION(C) class C; I3_POINTER_TYPEDEFS(C)
