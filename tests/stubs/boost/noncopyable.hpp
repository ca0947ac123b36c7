// stand-in (tests/stubs/README.md)
#pragma once
namespace boost {
class noncopyable {
protected:
    noncopyable() {}
    ~noncopyable() {}
private:
    noncopyable(const noncopyable &);
    noncopyable &operator=(const noncopyable &);
};
}
