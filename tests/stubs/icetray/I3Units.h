// stand-in (tests/stubs/README.md) for icetray/I3Units.h: the units the light-source adapter converts (length in metres,
// energy in GeV, time in ns, mass in grams are the base units of IceTray)
#pragma once
namespace I3Units {
static const double meter = 1., m = meter, cm = 1.e-2 * meter, cm3 = cm * cm * cm;
static const double gram = 1., g = gram;
static const double GeV = 1., TeV = 1.e3 * GeV;
static const double ns = 1., nanometer = 1.e-9 * meter;
}
