"""The IceTray build of the C++ adapter (-DCLSIMHIP_WITH_ICETRAY): I3CLSimStepToPhotonConverterHIP derived from the real
abstract interface, its four configuration setters with the signatures of
/root/reference/public/clsim/I3CLSimStepToPhotonConverter.h:91-124, translated by
clsim_amd/cxx/I3CLSimStepToPhotonConverterHIPGlue.h.  IceTray is not in this image: the build is against the stand-in
headers of tests/stubs/ (same class, constructor, getter and private-member names; tests/stubs/README.md).

CPU: I3CLSimMediumProperties objects built like python/MakeIceCubeMediumProperties.py does arrive in the library with
every number intact (SPICE-Lea: tilt, anisotropy, transforms; SPICE-Mie; photonics tables; homogeneous), unsupported
classes are refused, the canonical configuration sequence runs up to Compile().
GPU: I3CLSimModuleHelper::initializeHIP -> EnqueueSteps(I3CLSimStepSeriesConstPtr) -> GetConversionResult() yields the
same photons as the ctypes path on the same steps and streams."""
import os
import subprocess

import numpy as np
import pytest

from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = os.path.join(ROOT, "clsim_amd", "cxx")


def build(tmp_path):
    exe = str(tmp_path / "icetray_adapter_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-DCLSIMHIP_WITH_ICETRAY",
                           "-I" + os.path.join(ROOT, "tests", "stubs"), "-I" + CXX, "-o", exe, os.path.join(CXX, "icetray_adapter_test.cxx"),
                           "-L" + os.path.join(ROOT, "clsim_amd"), "-lclsimhip", "-Wl,-rpath," + os.path.join(ROOT, "clsim_amd")])
    return exe


@pytest.mark.parametrize("ice", ["spice_lea", "spice_mie"])
def test_glue_carries_every_medium_parameter(tmp_path, ice):
    args = [build(tmp_path), "check", os.path.join(common.ICE, ice)]
    if ice == "spice_lea":
        args.append(common.PHOTONICS["photonics_mie"])
    out = subprocess.check_output(args, text=True)
    assert out.count("medium round trip ok") == (3 if ice == "spice_lea" else 2)
    assert "icetray adapter ok" in out


def test_light_source_adapter_through_the_reference_interface(tmp_path):
    """I3CLSimLightSourceToStepConverterHIP (clsim_amd/cxx/): the reference's producer-side interface
    (public/clsim/I3CLSimLightSourceToStepConverter.h:61-198) on the feeder; no GPU: setters, messages, barrier"""
    out = subprocess.check_output([build(tmp_path), "lightsource_check", os.path.join(common.ICE, "spice_mie")], text=True)
    assert "light source adapter ok" in out


@pytest.mark.gpu
def test_light_source_adapter_yields_the_steps_of_the_python_feeder(tmp_path):
    """particles through the C++ interface (EnqueueLightSource(I3CLSimLightSource, id) ... GetConversionResultWithBarrierInfo)
    = the same particles through the ctypes feeder with the same seed: bunches, markers and steps bit for bit"""
    import threading
    from clsim_amd import converter as CV, step_store as SS
    from clsim_amd.synthetic import STEP_DTYPE
    steps_file = str(tmp_path / "steps.bin")
    out = subprocess.check_output([build(tmp_path), "lightsource_run", os.path.join(common.ICE, "spice_mie"), steps_file], text=True)
    lines = [l.split() for l in out.splitlines() if l.startswith("bunch")]
    cxx_steps = np.fromfile(steps_file, dtype=STEP_DTYPE)
    cfg = common.config("mie")
    ppc = CV.I3CLSimLightSourceToStepConverterPPC()
    ppc.SetWlenBias(CV.GetIceCubeDOMAcceptance()); ppc.SetMediumProperties(cfg["med_p"]); ppc.SetRandomSeed(5); ppc.Initialize()
    parts = np.zeros(3, dtype=CV.PARTICLE_DTYPE)
    parts["type"] = [CV.ParticleType.EMinus, CV.ParticleType.MuMinus, CV.ParticleType.Hadrons]
    parts["energy"] = [30.0, 100.0, 50.0]
    parts["length"] = [np.nan, 120.0, np.nan]
    parts["x"], parts["y"], parts["z"], parts["time"], parts["dz"] = 1.0, -2.0, 3.0, 5.0, -1.0
    parts["identifier"] = [11, 12, 13]
    f = SS.I3CLSimLightSourceToStepConverterAsync()
    f.SetMaxBunchSize(2048); f.SetBunchSizeGranularity(256); f.SetLightSourceParameterization(ppc, seed=5, device=0); f.Initialize()
    got = []

    def drain():
        while True:
            r = f.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=60000)
            assert r is not None
            got.append(r)
            if r[2]:
                return
    t = threading.Thread(target=drain)
    t.start()
    for p in parts:
        f.EnqueueLightSource(p)
    f.EnqueueBarrier()
    t.join(120)
    assert len(got) == len(lines)
    for (steps, finished, reset), line in zip(got, lines):
        i = line.index("finished"); j = line.index("reset")
        assert int(line[1]) == len(steps) and [int(v) for v in line[i + 1:j]] == finished and int(line[j + 1]) == int(reset)
    assert np.concatenate([s for s, _, _ in got]).tobytes() == cxx_steps.tobytes()
    assert int(cxx_steps["num"].sum()) > 100000


def _test_random_service(seed, count):
    """TestRandomService::Integer(0xffffffff) of icetray_adapter_test.cxx: splitmix64, high word % imax"""
    out = np.empty(count, dtype=np.uint64)
    s = seed
    mask = (1 << 64) - 1
    for i in range(count):
        s = (s + 0x9e3779b97f4a7c15) & mask
        z = s
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & mask
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & mask
        z ^= z >> 31
        out[i] = (z >> 32) % 0xffffffff
    return out


@pytest.mark.gpu
def test_icetray_interface_yields_the_photons_of_the_ctypes_path(tmp_path):
    from clsim_amd import converter as CV
    cfg = common.config("lea")
    n = 4096
    steps = common.steps_for(cfg, n, seed=31)
    g = cfg["geom"]
    geo_file, steps_file, out_file = (str(tmp_path / f) for f in ("geometry.txt", "steps.bin", "photons.bin"))
    with open(geo_file, "w") as f:
        for k in range(len(g["x"])):
            f.write("%d %d %.17g %.17g %.17g %s\n" % (g["string_ids"][k], g["dom_ids"][k], g["x"][k], g["y"][k], g["z"][k], g["subdetectors"][k]))
    steps.tofile(steps_file)
    out = subprocess.check_output([build(tmp_path), "run", os.path.join(common.ICE, "spice_lea"), geo_file, steps_file, out_file, str(n)], text=True)
    assert "identifier 4711" in out and "generated %d" % int(steps["num"].sum()) in out
    got = np.fromfile(out_file, dtype=CV.PHOTON_DTYPE)
    # the same streams: multipliers from the library, state words from the test's random service (mwcrng_init.h:104-112)
    a = CV.mwc_multipliers(n)
    draws = iter(_test_random_service(2024, 4 * n + 64))
    x = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        while x[i] == 0 or (int(x[i]) >> 32) >= int(a[i]) - 1 or (int(x[i]) & 0xffffffff) >= 0xffffffff:
            x[i] = (int(next(draws)) << 32) + int(next(draws))
    bias = CV.GetIceCubeDOMAcceptance()
    conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_dict(g), cfg["med_p"], bias, [CV.makeCherenkovWavelengthGenerator(bias, cfg["med_p"])],
                            pancakeFactor=5.0, approximateNumberOfWorkItems=n, streams=(x, a))
    conv.EnqueueSteps(steps, 1)
    _, want = conv.GetConversionResult()
    assert len(want) > 100 and len(got) == len(want)
    assert common.sort_photons(got).tobytes() == common.sort_photons(want).tobytes()
