#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the reference tree.

Runs ONLY in the build container (needs /root/reference); the fixtures it
writes are committed, the reference never travels.  Everything numerical below
is executed by the REFERENCE's own Python code:

  * mwc_multipliers.npy      first column of resources/scripts/compareToPPCredux/
                             test_ice_models/lea/rnd.txt (16028 safeprime MWC multipliers)
  * anisotropy_scaling.npz   DimasAbsLenScalingFactor() of resources/tests/testScalarFields.py:52-91
                             (the reference's Python port of PPC's formula) on seeded unit vectors
  * spice_lea_transforms.npz evaluateVectorTransformationPPCPre/Post() of
                             resources/tests/testSpiceLeaTransforms.py:37-72
  * vector_transform.npz     calculateDotProducts() of resources/tests/testVectorTransforms.py:52-60 (the reference
                             test's numpy answer for I3CLSimVectorTransformMatrix) on a seeded random matrix
  * ice_<model>.json         what python/MakeIceCubeMediumProperties.py (+ util/GetIceTiltZShift.py,
                             util/GetSpiceLeaAnisotropyTransforms.py) passes to the clsim C++
                             constructors for resources/ice/<model>
  * dom_acceptance.json      python/GetIceCubeDOMAcceptance.py
  * ppc_wavelength_cdf.txt   resources/scripts/compareToPPCredux/test_ice_models/lea/wv.dat (PPC's cumulative
                             photon spectrum: DOM acceptance x Cherenkov yield, 265..675 nm), copied as data
  * ice_photonics_<m>.npz    what python/MakeIceCubeMediumPropertiesPhotonics.py passes to the clsim C++
                             constructors for resources/ice/photonics_<m>/*.txt (per-layer FromTable functions)

The reference's Python modules import `icecube` (IceTray), which does not exist
here.  The loader scripts only use it to CONSTRUCT result objects, so they are
executed with a recording stand-in for those constructors (class Recorder below:
it stores the arguments it is called with and does no arithmetic).  The two test
files are not importable at all (they open an OpenCL device at import time), so
only the pure-Python reference functions named above are extracted from their
syntax tree and executed.
"""
import ast
import importlib.util
import json
import math
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


class Recorder:
    """Stands in for an icecube.clsim class: remembers how it was built."""

    def __init__(self, *args, **kwargs):
        self.cls = type(self).__name__
        self.args = args
        self.kwargs = kwargs
        self.calls = []

    def __getattr__(self, name):
        if name.startswith("Set") or name.startswith("Add"):
            def rec(*a, **k):
                self.calls.append((name, a, k))
            return rec
        raise AttributeError(name)


def recorder_module(name):
    mod = types.ModuleType(name)
    cache = {}

    def getattr_(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        if attr not in cache:
            cache[attr] = type(attr, (Recorder,), {})
        return cache[attr]
    mod.__getattr__ = getattr_
    return mod


class I3Units:          # icetray I3Units: base units m, ns, GeV, radian
    m = meter = 1.0
    cm = 0.01
    mm = 0.001
    cm3 = 1e-6
    meter2 = 1.0
    nanometer = 1e-9
    micrometer = 1e-6
    ns = 1.0
    g = 1.0
    deg = math.pi / 180.0


def install_stubs():
    icecube = types.ModuleType("icecube")
    icecube.__path__ = []
    clsim = recorder_module("icecube.clsim")
    clsim.__path__ = []
    icetray = types.ModuleType("icecube.icetray")
    icetray.I3Units = I3Units
    dataclasses = types.ModuleType("icecube.dataclasses")
    dataclasses.I3Matrix = lambda a: np.array(a, dtype=np.float64)
    i3tray = types.ModuleType("I3Tray")
    i3tray.I3Units = I3Units
    icecube.clsim, icecube.icetray, icecube.dataclasses = clsim, icetray, dataclasses
    sys.modules.update({"icecube": icecube, "icecube.clsim": clsim, "icecube.icetray": icetray,
                        "icecube.dataclasses": dataclasses, "I3Tray": i3tray})
    return clsim


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def extract_functions(path, names, env):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names), (path, names)
    code = compile(ast.Module(body=body, type_ignores=[]), path, "exec")
    exec(code, env)
    return [env[n] for n in names]


def to_jsonable(v):
    if isinstance(v, Recorder):
        return {"class": v.cls, "args": [to_jsonable(a) for a in v.args],
                "kwargs": {k: to_jsonable(x) for k, x in v.kwargs.items()},
                "calls": [[n, [to_jsonable(a) for a in aa], {k: to_jsonable(x) for k, x in kk.items()}] for n, aa, kk in v.calls]}
    if isinstance(v, np.ndarray):
        return {"ndarray": v.tolist(), "shape": list(v.shape)}
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [to_jsonable(a) for a in v]
    return v


def main():
    # ---- 1. MWC multipliers -------------------------------------------------
    rnd = np.loadtxt(os.path.join(REF, "resources/scripts/compareToPPCredux/test_ice_models/lea/rnd.txt"),
                     usecols=0, dtype=np.uint64)
    np.save(os.path.join(OUT, "mwc_multipliers.npy"), rnd.astype(np.uint32))

    wv = np.loadtxt(os.path.join(REF, "resources/scripts/compareToPPCredux/test_ice_models/lea/wv.dat"))
    np.savetxt(os.path.join(OUT, "ppc_wavelength_cdf.txt"), wv, fmt="%.8f %.1f")

    # ---- 2. reference Python ports of PPC formulas --------------------------
    rng = np.random.Generator(np.random.PCG64(20260101))
    n = 2000
    zen = np.arccos(rng.uniform(0., 1., n) * 2. - 1.)
    azi = rng.uniform(0., 2. * math.pi, n)
    x, y, z = np.sin(zen) * np.cos(azi), np.sin(zen) * np.sin(azi), np.cos(zen)
    (dimas,) = extract_functions(os.path.join(REF, "resources/tests/testScalarFields.py"),
                                 ["DimasAbsLenScalingFactor"], {"numpy": np, "math": math})
    thx, k1, k2 = 216., 0.04, -0.08
    np.savez(os.path.join(OUT, "anisotropy_scaling.npz"), x=x, y=y, z=z, thx=thx, logk1=k1, logk2=k2,
             expected=dimas(x, y, z, thx, k1, k2))
    pre, post = extract_functions(os.path.join(REF, "resources/tests/testSpiceLeaTransforms.py"),
                                  ["evaluateVectorTransformationPPCPre", "evaluateVectorTransformationPPCPost"],
                                  {"numpy": np, "math": math})
    azx, azy = np.cos(thx * I3Units.deg), np.sin(thx * I3Units.deg)
    ek1, ek2 = np.exp(k1), np.exp(k2)
    kz = 1. / (ek1 * ek2)
    vec = np.array([x, y, z]).T
    np.savez(os.path.join(OUT, "spice_lea_transforms.npz"), vectors=vec, thx=thx, logk1=k1, logk2=k2,
             pre=np.array([pre(v, azx, azy, ek1, ek2, kz) for v in vec]),
             post=np.array([post(v, azx, azy, ek1, ek2, kz) for v in vec]))

    (dots,) = extract_functions(os.path.join(REF, "resources/tests/testVectorTransforms.py"), ["calculateDotProducts"],
                                {"numpy": np, "math": math})
    matrix = rng.uniform(-10., 10., (3, 3))                     # testVectorTransforms.py:16
    np.savez(os.path.join(OUT, "vector_transform.npz"), matrix=matrix, x=x, y=y, z=z,
             plain=dots(matrix, x, y, z, renormalize=False), renormalized=dots(matrix, x, y, z, renormalize=True))

    # ---- 3. the reference's ice / acceptance loaders -----------------------
    clsim = install_stubs()
    util = types.ModuleType("icecube.clsim.util")
    sys.modules["icecube.clsim.util"] = util
    clsim.util = util
    util.GetIceTiltZShift = load(os.path.join(REF, "python/util/GetIceTiltZShift.py"),
                                 "icecube.clsim.util.GetIceTiltZShift").GetIceTiltZShift
    util.GetSpiceLeaAnisotropyTransforms = load(os.path.join(REF, "python/util/GetSpiceLeaAnisotropyTransforms.py"),
                                                "icecube.clsim.util.GetSpiceLeaAnisotropyTransforms").GetSpiceLeaAnisotropyTransforms
    mk = load(os.path.join(REF, "python/MakeIceCubeMediumProperties.py"), "icecube.clsim.MakeIceCubeMediumProperties")
    for model in ("spice_mie", "spice_lea"):
        m = mk.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(REF, "resources/ice", model))
        with open(os.path.join(OUT, "ice_%s.json" % model), "w") as f:
            json.dump(to_jsonable(m), f)
    mkp = load(os.path.join(REF, "python/MakeIceCubeMediumPropertiesPhotonics.py"),
               "icecube.clsim.MakeIceCubeMediumPropertiesPhotonics")
    for tag, rel in (("spice_mie", "photonics_spice_mie/Ice_table.mie.i3coords.cos090.08Apr2011.txt"),
                     ("wham", "photonics_wham/Ice_table.wham.i3coords.cos090.11jul2011.txt")):
        m = mkp.MakeIceCubeMediumPropertiesPhotonics(tableFile=os.path.join(REF, "resources/ice", rel))
        n = m.kwargs["layersNum"]
        rec = dict(layersNum=n, layersZStart=m.kwargs["layersZStart"], layersHeight=m.kwargs["layersHeight"])
        tabs = {"SetAbsorptionLength": [None] * n, "SetScatteringLength": [None] * n,
                "SetPhaseRefractiveIndex": [None] * n, "SetGroupRefractiveIndexOverride": [None] * n}
        for name, a, k in m.calls:
            if name in tabs:
                f = a[1]
                assert f.cls == "I3CLSimFunctionFromTable"
                tabs[name][a[0]] = (f.args[0], f.args[1], np.array(f.args[2], dtype=np.float64),
                                    bool(f.kwargs.get("storeDataAsHalfPrecision", False)), id(f))
            elif name == "SetScatteringCosAngleDistribution":
                assert a[0].cls == "I3CLSimRandomValueHenyeyGreenstein"
                rec["meanCosine"] = a[0].kwargs["meanCosine"]
            elif name == "SetIceTiltZShift":
                assert a[0].cls == "I3CLSimScalarFieldConstant" and a[0].args == (0.,)
            elif name == "SetDirectionalAbsorptionLengthCorrection":
                assert a[0].cls == "I3CLSimScalarFieldConstant" and a[0].args == (1.,)
            elif name in ("SetPreScatterDirectionTransform", "SetPostScatterDirectionTransform"):
                assert a[0].cls == "I3CLSimVectorTransformConstant"
        for name, key in (("SetAbsorptionLength", "abs"), ("SetScatteringLength", "sca"),
                          ("SetPhaseRefractiveIndex", "phase"), ("SetGroupRefractiveIndexOverride", "group")):
            t = tabs[name]
            assert all(x is not None and x[0] == t[0][0] and x[1] == t[0][1] and x[3] == t[0][3] for x in t)
            rec[key + "_start"], rec[key + "_step"], rec[key + "_16bit"] = t[0][0], t[0][1], t[0][3]
            rec[key + "_same_object"] = len(set(x[4] for x in t)) == 1
            rec[key] = np.array([x[2] for x in t])
        np.savez_compressed(os.path.join(OUT, "ice_photonics_%s.npz" % tag), **rec)
    acc = load(os.path.join(REF, "python/GetIceCubeDOMAcceptance.py"), "icecube.clsim.GetIceCubeDOMAcceptance")
    a = acc.GetIceCubeDOMAcceptance()
    with open(os.path.join(OUT, "dom_acceptance.json"), "w") as f:
        json.dump(to_jsonable(a), f)
    print("fixtures written to", OUT)


if __name__ == "__main__":
    main()
