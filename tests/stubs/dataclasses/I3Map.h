// stand-in (tests/stubs/README.md) for dataclasses/I3Map.h
#pragma once
#include <map>
#include <icetray/I3FrameObject.h>
template <typename K, typename V>
struct I3Map : public std::map<K, V>, public I3FrameObject {
    template <class Archive> void serialize(Archive &ar, unsigned version);
};
struct OMKey;
struct ModuleKey;
