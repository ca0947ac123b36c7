// stand-in (tests/stubs/README.md): log_fatal ends the job in IceTray; here it throws so that tests can see it
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>
namespace i3stub {
template <class... A> inline std::string format(const char *fmt, A... a)
{
    char buf[1024];
    std::snprintf(buf, sizeof buf, fmt, a...);
    return buf;
}
inline std::string format(const char *fmt) { return fmt; }
}
#define log_fatal(...) throw std::runtime_error(std::string("log_fatal: ") + i3stub::format(__VA_ARGS__))
#define log_error(...) std::fprintf(stderr, "%s\n", i3stub::format(__VA_ARGS__).c_str())
#define log_warn(...) std::fprintf(stderr, "%s\n", i3stub::format(__VA_ARGS__).c_str())
#define log_info(...) ((void)0)
#define log_debug(...) ((void)0)
#define log_trace(...) ((void)0)
