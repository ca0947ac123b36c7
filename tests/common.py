"""Builds the same configuration on both sides: the oracle (numpy builders +
C restatement) and the product (libclsimhip.so through its Python mirror)."""
import os

import numpy as np

from clsim_amd import converter as CV
from clsim_amd import synthetic as S
from oracle import builders as B
from oracle import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ICE = os.path.join(ROOT, "clsim_amd", "data", "ice")
PHOTONICS = {"photonics_mie": os.path.join(ICE, "photonics_spice_mie", "Ice_table.mie.i3coords.cos090.08Apr2011.txt"),
             "photonics_wham": os.path.join(ICE, "photonics_wham", "Ice_table.wham.i3coords.cos090.11jul2011.txt")}
FLASHER_WLEN = 405e-9

_streams = {}


def streams(n, seed=12345):
    """(x, a) for n RNG streams: multipliers from the product's generator are
    checked against the oracle's in tests/test_rng.py; here the oracle's are used
    for small n and the product's for large n."""
    key = (n, seed)
    if key not in _streams:
        a = B.mwc_multipliers(n) if n <= 4096 else CV.mwc_multipliers(n)
        x = B.seed_streams(a, seed) if n <= 4096 else CV.seed_streams(a, seed)
        _streams[key] = (x, a)
    return _streams[key]


def config(name):
    """name: 'c1' homogeneous/single string, 'mie' SPICE-Mie/IC86, 'lea' SPICE-Lea/IC86,
    'flasher' SPICE-Lea/IC86 + 405 nm generator; '<name>_regular': the same with a detector whose DOMs sit exactly on
    their string axes -- the strings then share two DOM position templates (GeometrySource.cxx:449-495); '<name>_60': 60 of
    the 86 strings (52 standard + the 8 dense ones: both subdetectors) -- the largest kind of detector for which the
    reference's search WITHOUT STOP_PHOTONS_ON_DETECTION stays inside its bit mask (oracle/clsim_oracle.c:
    checkForCollision_OnString_keep)."""
    if name.endswith("_60"):
        cfg = config(name[:-len("_60")])
        g = cfg["geom"]
        ids = np.asarray(g["string_ids"])
        keep = (ids <= 52) | (ids >= 79)
        geom = {k: (np.asarray(v)[keep] if (hasattr(v, "__len__") and not isinstance(v, str) and len(v) == len(ids)) else v) for k, v in g.items()}
        assert len(np.unique(geom["string_ids"])) == 60
        return dict(cfg, name=name, geom=geom)
    if name.endswith("_regular"):
        cfg = config(name[:-len("_regular")])
        return dict(cfg, name=name, geom=S.ic86_geometry(jitter=0.0))
    if name.endswith("_dispersion"):
        # the same medium WITHOUT a group refractive index override: the group velocity then comes from the phase index and its
        # derivative (I3CLSimHelperGenerateMediumPropertiesSource.cxx:274-300); CLSIMHIP_REFINDEX_DISPERSION in the C ABI
        import ctypes as C
        from clsim_amd import _lib
        cfg = config(name[:-len("_dispersion")])
        d = _lib.MediumDesc()
        assert _lib.load().clsimhip_medium_describe(cfg["med_p"]._h, C.byref(d)) == 0
        d.group_index_kind = 2
        h = C.c_void_p()
        assert _lib.load().clsimhip_medium_create(C.byref(d), C.byref(h)) == 0
        return dict(cfg, name=name, med_o=dict(cfg["med_o"], group_from_dispersion=True), med_p=CV.I3CLSimMediumProperties(h, keep=cfg["med_p"]))
    if name == "c1":
        geom = S.single_string_geometry()
        med_o = B.homogeneous_medium()
        med_p = CV.MakeHomogeneousMediumProperties()
    elif name.startswith("clear"):
        # one layer of ice nobody has seen: scattering length 600 m, absorption length 2 km -- segments that pass dozens of
        # DOMs and several strings (the search without STOP_PHOTONS_ON_DETECTION: many hits per photon, its bit masks at work)
        geom = S.ic86_geometry()
        med_o = B.homogeneous_medium(abs_len=2000.0, sca_len=600.0)
        med_p = CV.MakeHomogeneousMediumProperties(absLen=2000.0, scaLen=600.0)
    elif name.startswith("photonics"):
        geom = S.ic86_geometry()
        med_o = B.load_photonics_ice(PHOTONICS[name])
        med_p = CV.MakeIceCubeMediumPropertiesPhotonics(PHOTONICS[name])
    else:
        geom = S.ic86_geometry()
        d = os.path.join(ICE, "spice_mie" if name == "mie" else "spice_lea")
        med_o = B.load_ppc_ice(d)
        med_p = CV.MakeIceCubeMediumProperties(iceDataDirectory=d)
    # 'flasher': the second generator is a 405 nm delta peak; 'flasher_led405': the LED's measured emission spectrum, a table with
    # its own wavelengths (GetIceCubeFlasherSpectrum.py -> makeWavelengthGenerator -> InterpolatedDistribution(x, y))
    return dict(name=name, geom=geom, med_o=med_o, med_p=med_p, flasher=name.startswith("flasher"),
                led=("LED405nm" if name.startswith("flasher_led405") else None))


def oracle_generators(cfg, bias):
    gens = [B.cherenkov_wlen_generator(bias, cfg["med_o"])]
    if cfg.get("led"):
        gens.append(B.make_wavelength_generator(B.flasher_spectrum(cfg["led"], CV.FLASHER_DATA), bias, cfg["med_o"]))
    elif cfg["flasher"]:
        gens.append(dict(kind="const", value=FLASHER_WLEN))
    return gens


def product_generators(cfg, bias):
    gens = [CV.makeCherenkovWavelengthGenerator(bias, cfg["med_p"])]
    if cfg.get("led"):
        gens.append(CV.makeWavelengthGenerator(CV.GetIceCubeFlasherSpectrum(cfg["led"]), bias, cfg["med_p"]))
    elif cfg["flasher"]:
        gens.append(CV.I3CLSimRandomValueConstant(FLASHER_WLEN))
    return gens


def oracle_tables(cfg, pancake=5.0, stop_detected=True):
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    bias = B.icecube_dom_acceptance()
    gens = oracle_generators(cfg, bias)
    return capi.make_tables(cfg["med_o"], geo, gens, bias, pancake=pancake, stop_detected=stop_detected)


def product_converter(cfg, max_items, pancake=5.0, initialize=True, device=0, seed=12345, double_buffering=False, stop_detected=True):
    bias = CV.GetIceCubeDOMAcceptance()
    gens = product_generators(cfg, bias)
    geom = CV.I3CLSimSimpleGeometry.from_dict(cfg["geom"])
    if not initialize:
        conv = CV.I3CLSimStepToPhotonConverterHIP(device)
        conv.SetWlenGenerators(gens); conv.SetWlenBias(bias); conv.SetMediumProperties(cfg["med_p"])
        conv.SetGeometry(geom); conv.SetDOMPancakeFactor(pancake)
        conv.SetStopDetectedPhotons(stop_detected)          # (the class's own default is false, OpenCL.cxx:86)
        return conv
    return CV.initializeHIP(device, geom, cfg["med_p"], bias, gens, pancakeFactor=pancake, enableDoubleBuffering=double_buffering,
                            stopDetectedPhotons=stop_detected, approximateNumberOfWorkItems=max_items, streams=streams(max_items, seed))


def steps_for(cfg, n, seed=3, pad_to=256):
    if cfg["name"] == "c1":
        return S.cascade_steps(n, seed=seed, vertex=(0.0, 0.0, 0.0), pad_to=pad_to)
    if cfg["flasher"]:
        g = cfg["geom"]
        k = 30 * 60 + 29      # a DOM in the middle of the detector
        return S.flasher_steps(n, seed=seed, position=(g["x"][k] + 12.0, g["y"][k], g["z"][k]), pad_to=pad_to)
    return S.cascade_steps(n, seed=seed, pad_to=pad_to)


def sort_photons(ph):
    return capi.sort_photons(ph)
