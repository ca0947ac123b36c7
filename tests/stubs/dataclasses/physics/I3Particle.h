// stand-in (tests/stubs/README.md) for dataclasses/physics/I3Particle.h: the getters the light-source adapter reads and the
// enumerators it maps (PDG codes and IceCube's own codes, as dataclasses defines them)
#pragma once
#include <cmath>
#include <cstdint>
#include <dataclasses/I3Position.h>
#include <dataclasses/I3Direction.h>
class I3Particle {
public:
    enum ParticleType { unknown = 0, Gamma = 22, EPlus = -11, EMinus = 11, MuPlus = -13, MuMinus = 13, TauPlus = -15, TauMinus = 15, Pi0 = 111,
                        PiPlus = 211, PiMinus = -211, K0_Long = 130, KPlus = 321, KMinus = -321, K0_Short = 310, PPlus = 2212, PMinus = -2212,
                        Neutron = 2112, Brems = -2000001001, DeltaE = -2000001002, PairProd = -2000001003, NuclInt = -2000001004,
                        Hadrons = -2000001006 };
    enum ParticleShape { Null = 0, Primary = 10, TopShower = 20, Cascade = 30, CascadeSegment = 31, InfiniteTrack = 40, StartingTrack = 50,
                         StoppingTrack = 60, ContainedTrack = 70, MCTrack = 80, Dark = 90 };
    I3Particle() : type_(unknown), shape_(Null), time_(0), energy_(0), length_(NAN) {}
    ParticleType GetType() const { return type_; }
    ParticleShape GetShape() const { return shape_; }
    const I3Position &GetPos() const { return pos_; }
    const I3Direction &GetDir() const { return dir_; }
    double GetTime() const { return time_; }
    double GetEnergy() const { return energy_; }
    double GetLength() const { return length_; }
    void SetType(ParticleType t) { type_ = t; }
    void SetShape(ParticleShape s) { shape_ = s; }
    void SetPos(const I3Position &p) { pos_ = p; }
    void SetDir(const I3Direction &d) { dir_ = d; }
    void SetTime(double t) { time_ = t; }
    void SetEnergy(double e) { energy_ = e; }
    void SetLength(double l) { length_ = l; }
private:
    ParticleType type_;
    ParticleShape shape_;
    I3Position pos_;
    I3Direction dir_;
    double time_, energy_, length_;
};
