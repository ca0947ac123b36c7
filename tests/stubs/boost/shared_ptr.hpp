// stand-in (tests/stubs/README.md): boost smart pointers as aliases of the std ones
#pragma once
#include <memory>
namespace boost {
using std::shared_ptr;
using std::dynamic_pointer_cast;
using std::const_pointer_cast;
using std::make_shared;
}
