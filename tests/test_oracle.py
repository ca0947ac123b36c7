"""The oracle itself (CPU only): RNG against a big-integer statement of the MWC
recurrence, the math spec against binary64 libm, host helpers, and the restated
kernel against size-independent properties plus a committed snapshot of its own
output (tests/golden/oracle_c1_hits.npz) that guards against accidental edits."""
import os

import numpy as np
import pytest

from clsim_amd import converter as CV
from clsim_amd import synthetic as S
from oracle import builders as B
from oracle import capi
from tests import common

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_mwc_recurrence_and_float_conversion(oracle_lib):
    """mwcrng_kernel.cl:12-20: x <- lo32(x)*a + hi32(x); u = float_rtz(lo32(x)) / 2^32."""
    a = int(B.mwc_multipliers(3)[2])
    x = int(B.seed_streams(np.array([a], dtype=np.uint32), seed=99)[0])
    got, x_after = capi.eval_rng(x, a, 1000)
    xs = x
    for i in range(1000):
        xs = (xs & 0xFFFFFFFF) * a + (xs >> 32)
        lo = xs & 0xFFFFFFFF
        drop = max(lo.bit_length() - 24, 0)
        exp = np.float32(((lo >> drop) << drop) / 4294967296.0)      # exact in binary64, then exact in binary32
        assert got[i] == exp and 0.0 <= got[i] < 1.0
    assert x_after == xs


def test_seed_streams_valid_and_equal_to_product():
    a = B.mwc_multipliers(2048)
    x = B.seed_streams(a, seed=12345)
    assert np.array_equal(x, CV.seed_streams(a, seed=12345))
    hi, lo = (x >> np.uint64(32)).astype(np.uint64), (x & np.uint64(0xFFFFFFFF))
    assert np.all(x != 0) and np.all(hi < a.astype(np.uint64) - 1) and np.all(lo < 0xFFFFFFFF)   # mwcrng_init.h:107
    assert not np.array_equal(x, B.seed_streams(a, seed=12346))


def test_float_literal_round_trip():
    assert B.float_literal(1e-9) == np.float32(1e-9) and B.float_literal(0.0) == 0.0
    # 11 significant digits, then correctly rounded to binary32 (ToFloatString.h:36-59)
    assert B.float_literal(0.299792458) == np.float32(0.299792458)
    v = 71.402900695801
    assert B.float_literal(v) == np.float32(float("%.10e" % v))
    assert B.float_literal(1.0 / 3.0) == np.float32(0.33333333333)


@pytest.mark.parametrize("what,lo,hi,ref,ulps", [
    # (round 5: log and sin / cos on [0, 2 pi] are the table forms -- oracle/mathcheck.c scans every float: 1.21 / 2.36 / 2.47 ulp;
    # OpenCL allows 3 / 4 / 4.  Outside [0, 2 pi] sin and cos are the Cephes forms as before.)
    (0, 6e-8, 1.0, np.log, 1.25), (0, 1.0, 1000.0, np.log, 1.0), (1, -30.0, 0.0, np.exp, 1.2), (2, 0.0, 6.2831855, np.sin, 2.4),
    (3, 0.0, 6.2831855, np.cos, 2.5), (2, -100.0, 0.0, np.sin, 1.6), (3, 6.2832, 100.0, np.cos, 1.6), (5, -1.0, 1.0, np.arccos, 0.6), (8, 0.0, 100.0, np.sqrt, 0.5001),
    (10, -1.0, 1.0, np.arccos, 2.0)])
def test_math_spec_accuracy(oracle_lib, what, lo, hi, ref, ulps):
    rng = np.random.Generator(np.random.PCG64(what))
    x = (lo + (hi - lo) * rng.random(200000)).astype(np.float32)
    got = capi.eval_math(what, x).astype(np.float64)
    want = ref(x.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got - want) / ulp) <= ulps


def test_powr_accuracy_on_the_ice_model_ranges(oracle_lib):
    rng = np.random.Generator(np.random.PCG64(7))
    for lo, hi, y in ((265.0, 675.0, -1.084106802940), (0.66, 1.69, -0.898608505726), (1e-9, 1.0, 0.0526315793)):
        x = (lo + (hi - lo) * rng.random(200000)).astype(np.float32)
        yy = np.full_like(x, np.float32(y))
        got = capi.eval_math(4, x, yy).astype(np.float64)
        want = np.power(x.astype(np.float64), np.float64(np.float32(y)))
        ulp = np.spacing(want.astype(np.float32)).astype(np.float64)
        assert np.max(np.abs(got - want) / ulp) <= 1.3
    assert capi.eval_math(4, np.array([0.0], np.float32), np.array([0.05], np.float32))[0] == 0.0


def test_ice_functions_against_host_formulas(oracle_lib):
    """Device-side restatement vs the reference's host GetValue() formulas
    (AbsLenIceCube.cxx:63-67, ScatLenIceCube.cxx:54-58, RefIndexIceCube.cxx:84-101) in double."""
    cfg = common.config("mie")
    T = common.oracle_tables(cfg)
    m = cfg["med_o"]
    wl = np.linspace(265e-9, 675e-9, 83).astype(np.float32)
    for layer in (0, 17, 85, 170):
        x = wl.astype(np.float64) / 1e-9
        absl = 1.0 / ((m["D"] * m["aDust400"][layer] + m["E"]) * x ** (-m["kappa"]) + m["A"] * np.exp(-m["B"] / x) * (1.0 + 0.01 * m["deltaTau"][layer]))
        scal = 1.0 / (m["b400"][layer] * (x / 400.0) ** (-m["alpha"]))
        assert np.allclose(capi.eval_medium(T, 0, wl, layer), absl, rtol=2e-6)
        assert np.allclose(capi.eval_medium(T, 1, wl, layer), scal, rtol=2e-6)
    xm = wl.astype(np.float64) / 1e-6
    n, g = m["n"], m["g"]
    nphase = n[0] + xm * (n[1] + xm * (n[2] + xm * (n[3] + xm * n[4])))
    ngroup = nphase * (g[0] + xm * (g[1] + xm * (g[2] + xm * (g[3] + xm * g[4]))))
    assert np.allclose(capi.eval_medium(T, 2, wl), nphase, rtol=1e-6)
    assert np.allclose(capi.eval_medium(T, 3, wl), 0.299792458 / ngroup, rtol=1e-6)


def test_group_velocity_from_the_dispersion_against_the_host_formula(oracle_lib):
    """A medium WITHOUT a group refractive index override: getGroupVelocity = c (1 + y lambda / n) / n with y = dn/dlambda of the phase
    index (I3CLSimHelperGenerateMediumPropertiesSource.cxx:274-300, I3CLSimFunctionRefIndexIceCube.cxx:107-112 GetDerivative), here in
    double; and what it is physically: within 1.1 % of the parameterised group index the IceCube media set as an override (the
    `g` polynomial is a fit to this very derivative form, resources/plots/medium_properties_IceCube.py:68-80), never equal to it."""
    T = common.oracle_tables(common.config("mie_dispersion"))
    T_override = common.oracle_tables(common.config("mie"))
    wl = np.linspace(265e-9, 675e-9, 411).astype(np.float32)
    w = wl.astype(np.float64)
    n = B.REFINDEX_N
    x = w / 1e-6
    n_phase = n[0] + x * (n[1] + x * (n[2] + x * (n[3] + x * n[4])))
    dn = (n[1] + x * (2. * n[2] + x * (3. * n[3] + x * 4. * n[4]))) / 1e-6
    v = capi.eval_medium(T, 3, wl)
    assert np.allclose(v, 0.299792458 * (1.0 + dn * w / n_phase) / n_phase, rtol=1e-6)
    assert np.array_equal(capi.eval_medium(T, 2, wl), capi.eval_medium(T_override, 2, wl))          # the phase index is the same function
    rel = v / capi.eval_medium(T_override, 3, wl) - 1.0
    assert 1e-4 < np.abs(rel).max() < 0.011 and np.all(v < 0.299792458 / 1.3)
    # the table maker needs an override (I3CLSimStepToTableConverter.cxx:103-104)
    with pytest.raises(ValueError, match="group refractive indices"):
        B.minimum_refractive_index(common.config("mie_dispersion")["med_o"])


def test_tilt_against_host_formula(oracle_lib):
    """resources/tests/testScalarFieldIceTiltZShift.py: device vs host GetValue(), <= 10 cm."""
    cfg = common.config("mie")
    T = common.oracle_tables(cfg)
    t = cfg["med_o"]["tilt"]
    rng = np.random.Generator(np.random.PCG64(3))
    xyz = (rng.random((20000, 3)) * 2400.0 - 1200.0).astype(np.float32)
    got = capi.eval_field(T, 0, xyz)
    lnx, lny = np.cos(t["azimuth"]), np.sin(t["azimuth"])
    first, dz = B.tilt_spacing(t["zcoords"])
    d, zc = t["distances"], t["zcorr"]
    exp = np.zeros(len(xyz))
    for i, (x, y, z) in enumerate(xyz.astype(np.float64)):
        zr = (z - first) / dz
        k = int(min(max(np.floor(zr), 0.0), len(t["zcoords"]) - 2))
        fa, fb = zr - k, (k + 1) - zr
        nr = lnx * x + lny * y
        for j in range(1, len(d)):
            if nr < d[j] or j == len(d) - 1:
                w = d[j] - d[j - 1]
                exp[i] = (zc[j][k + 1] * fa + zc[j][k] * fb) * ((nr - d[j - 1]) / w) + (zc[j - 1][k + 1] * fa + zc[j - 1][k] * fb) * ((d[j] - nr) / w)
                break
    assert np.max(np.abs(got - exp)) < 0.1
    assert np.max(np.abs(got - exp)) < 2e-2          # in fact float rounding only (extrapolated bins amplify it)


def _c1_run(threads=1):
    cfg = common.config("c1")
    steps = common.steps_for(cfg, 1000, seed=3, pad_to=512)
    x, a = common.streams(len(steps))
    T = common.oracle_tables(cfg)
    return cfg, T, steps, capi.propagate(T, steps, x, a, threads=threads)


def test_oracle_kernel_snapshot_and_threads(oracle_lib):
    cfg, T, steps, (ph, cnt, x_after, iters) = _c1_run(1)
    _, _, _, (ph_mt, cnt_mt, x_mt, iters_mt) = _c1_run(4)
    assert cnt == cnt_mt and iters == iters_mt and np.array_equal(x_after, x_mt)
    assert common.sort_photons(ph).tobytes() == common.sort_photons(ph_mt).tobytes()
    path = os.path.join(G, "oracle_c1_hits.npz")
    if not os.path.exists(path):                      # first run in the build container writes the snapshot
        np.savez_compressed(path, photons=common.sort_photons(ph), x_after=x_after, iterations=iters)
    snap = np.load(path)
    assert snap["photons"].tobytes() == common.sort_photons(ph).tobytes()
    assert np.array_equal(snap["x_after"], x_after) and int(snap["iterations"]) == iters


def test_oracle_kernel_physical_properties(oracle_lib):
    cfg = common.config("mie")
    steps = common.steps_for(cfg, 2048, seed=5)
    x, a = common.streams(len(steps))
    T = common.oracle_tables(cfg)
    ph, cnt, x_after, iters = capi.propagate(T, steps, x, a, threads=8)
    assert 50 < cnt < 0.01 * steps["num"].sum()
    assert np.all(x_after[steps["num"] > 0] != x[:len(steps)][steps["num"] > 0])
    assert 20 < iters / steps["num"].sum() < 40       # ~ absorption / scattering length
    assert np.all((ph["wavelength"] >= 260e-9) & (ph["wavelength"] <= 680e-9))
    assert np.all(ph["t"] > ph["st"]) and np.all(ph["cherenkovDist"] > 0)
    v = ph["groupVelocity"]
    assert np.all((v > 0.20) & (v < 0.23))            # c / n_group, n_group ~ 1.35
    # time of flight = path / group velocity (accumulated per segment in float)
    assert np.allclose(ph["t"] - ph["st"], ph["cherenkovDist"] / v, rtol=1e-4)
    # hit position is relative to the DOM centre; un-pancaked it lies within the oversized radius
    r = np.sqrt(ph["x"].astype(np.float64) ** 2 + ph["y"] ** 2 + ph["z"] ** 2)
    assert np.all(r <= 0.8255 * 1.0001) and np.all(r >= 0.8255 / 5 * 0.999)
    assert np.all(ph["stringID"] >= 0) and np.all(ph["stringID"] < 86) and np.all(ph["omID"] < 60)
    bias = B.icecube_dom_acceptance()
    w = np.array([1.0 / B.from_table_host(bias, float(wl)) for wl in ph["wavelength"]])
    assert np.allclose(ph["weight"], w, rtol=1e-5)
    ids = capi.replace_indices_with_ids(ph, T.geo)
    assert np.all((ids["stringID"] >= 1) & (ids["stringID"] <= 86)) and np.all((ids["omID"] >= 1) & (ids["omID"] <= 60))


def test_oracle_edge_cases(oracle_lib):
    """Empty / padded steps consume nothing but keep their stream; a vertical
    photon direction skips the DOM search (collision c.cl:511-512)."""
    cfg = common.config("c1")
    T = common.oracle_tables(cfg)
    steps = common.steps_for(cfg, 256, seed=1, pad_to=512)
    x, a = common.streams(512)
    ph, cnt, x_after, _ = capi.propagate(T, steps, x, a)
    assert np.array_equal(x_after[256:], x[256:512])              # padded steps: numPhotons = 0
    zero = steps.copy(); zero["num"] = 0
    ph0, cnt0, x0, it0 = capi.propagate(T, zero, x, a)
    assert cnt0 == 0 and it0 == 0 and np.array_equal(x0, x[:512])
    tiny = capi.propagate(T, steps[:512], x, a, max_hits=3)        # overflow: counter runs on, 3 stored
    assert tiny[1] == cnt and len(tiny[0]) == 3


def test_cube_root_of_the_power_axes_is_within_opencl_bounds(oracle_lib):
    """om_cbrt (tabulator power axes with power 3, Axis.cxx:164-165): <= 2 ulp like OpenCL's cbrt, odd, exact at 0"""
    rng = np.random.Generator(np.random.PCG64(9))
    x = np.concatenate([rng.uniform(-7e3, 7e3, 200000), np.exp(rng.uniform(np.log(1e-6), np.log(1e6), 200000))]).astype(np.float32)
    got = capi.eval_math(15, x).astype(np.float64)
    want = np.cbrt(x.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got - want) / ulp) <= 2.0
    assert capi.eval_math(15, np.array([0.0, 8.0, -27.0], dtype=np.float32)).tolist() == [0.0, 2.0, -3.0]


def test_threaded_driver_without_stop_keeps_every_record():
    """oracle_propagate_mt gives every step room for numPhotons records; without STOP_PHOTONS_ON_DETECTION a photon is recorded
    by every DOM on its way, so a step can need more -- the driver runs such a step again with more room.  Steps placed at DOMs
    (tests/test_search_filter_gpu.py: boundary_steps), threaded against the serial driver."""
    from tests.test_search_filter_gpu import boundary_steps
    cfg = common.config("mie")
    steps = boundary_steps(cfg, 8192, seed=5, reach=1.9, photons=4)           # few photons per step: several records per photon matter
    x, a = common.streams(len(steps))
    T = common.oracle_tables(cfg, stop_detected=False)
    ph_1, cnt_1, x_1, _ = capi.propagate(T, steps, x, a, threads=1)
    for rep in range(3):
        ph_n, cnt_n, x_n, _ = capi.propagate(T, steps, x, a, threads=8)
        assert cnt_n == cnt_1 and np.array_equal(x_n, x_1)
        assert common.sort_photons(ph_n).tobytes() == common.sort_photons(ph_1).tobytes()
    per_step = np.bincount(ph_1["id"], minlength=1)
    assert per_step.max() > 4                                                  # some step recorded more hits than it has photons


def test_product_and_oracle_math_tables_are_the_same_text():
    """tools/make_math_tables.py writes the constants of the table-driven log / sincos twice -- the product may not include from
    oracle/ -- and the two bodies must be identical (and regenerate identically)."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def body(path):
        t = open(os.path.join(root, path)).read()
        return re.sub(r"^#ifndef \w+\n#define \w+\n", "", t)
    a, b = body("clsim_amd/csrc/math_tables.h"), body("oracle/math_tables.h")
    assert a == b and "MT_LOG_TABLE" in a and "MT_SC_TABLE" in a
    # the generator in its --check mode: compares what it would write with the checked-in headers and touches nothing (the headers are
    # Makefile dependencies of every kernel)
    pytest.importorskip("mpmath")
    stamp = os.path.getmtime(os.path.join(root, "oracle", "math_tables.h"))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "make_math_tables.py"), "--check"], stdout=subprocess.DEVNULL)
    assert os.path.getmtime(os.path.join(root, "oracle", "math_tables.h")) == stamp
