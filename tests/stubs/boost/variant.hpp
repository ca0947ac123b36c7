// stand-in (tests/stubs/README.md): boost::variant as std::variant (the producer interface's parameter type only)
#pragma once
#include <variant>
namespace boost {
template <class... T> using variant = std::variant<T...>;
}
