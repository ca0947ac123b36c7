// stand-in (tests/stubs/README.md): dataclasses/I3Vector.h -- a std::vector that is also a frame object
#pragma once
#include <vector>
#include <icetray/I3FrameObject.h>
template <typename T>
struct I3Vector : public std::vector<T>, public I3FrameObject {
    I3Vector() {}
    explicit I3Vector(typename std::vector<T>::size_type s) : std::vector<T>(s) {}
    I3Vector(typename std::vector<T>::size_type s, const T &v) : std::vector<T>(s, v) {}
    template <typename It> I3Vector(It first, It last) : std::vector<T>(first, last) {}
    template <class Archive> void serialize(Archive &ar, unsigned version);
};
