"""The reference's own tests compare single generated functions ON THE DEVICE with their host implementation through tester
classes (private/test/I3CLSim*Tester + resources/kernels/*_test_kernel: resources/tests/testScalarFields.py:52-131 -- anisotropy
scaling, device vs host vs a Python port of PPC, 1e5 random directions, relative deviation <= 1e-5; testScalarFieldIceTiltZShift.py --
tilt, +-1200 m cube, <= 10 cm; testVectorTransforms.py -- matrix transforms vs numpy).  The same tests here, with
clsimhip_eval_device_function / _random as the tester: the device against the oracle's function of the same name (bit for bit, both
forms the kernels use) AND against the independent formulas the reference's tests hold (within their tolerances)."""
import math

import numpy as np
import pytest

from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setups():
    out = {}
    for name in ("mie", "lea", "photonics_mie", "c1", "flasher_led405"):
        cfg = common.config(name)
        out[name] = (cfg, common.oracle_tables(cfg), common.product_converter(cfg, 1024))
    return out


def unit_vectors(n, seed):
    rng = np.random.default_rng(seed)
    cos_t = rng.uniform(-1, 1, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    s = np.sqrt(1 - cos_t * cos_t)
    return np.stack([s * np.cos(phi), s * np.sin(phi), cos_t], axis=1).astype(np.float32)


@pytest.mark.parametrize("name", ["mie", "lea", "photonics_mie", "c1"])
def test_medium_functions_device_equals_oracle(setups, name):
    cfg, T, conv = setups[name]
    wl = np.linspace(266e-9, 674e-9, 4001).astype(np.float32)
    forms = (False, True) if name in ("mie", "lea") else (False,)
    n_layers = int(T.t.num_layers)
    for layer in sorted({0, n_layers // 3, n_layers // 2, n_layers - 1}):
        for fast in forms:
            dev = conv.EvaluateOnDevice("lengths", wl, layer=layer, fast=fast)
            assert np.array_equal(dev[:, 0].view(np.uint32), capi.eval_medium(T, 0, wl, layer).view(np.uint32))
            assert np.array_equal(dev[:, 1].view(np.uint32), capi.eval_medium(T, 1, wl, layer).view(np.uint32))
    dev = conv.EvaluateOnDevice("refraction", wl)
    assert np.array_equal(dev[:, 0].view(np.uint32), capi.eval_medium(T, 2, wl).view(np.uint32))
    assert np.array_equal(dev[:, 1].view(np.uint32), capi.eval_medium(T, 3, wl).view(np.uint32))
    wide = np.linspace(200e-9, 750e-9, 2001).astype(np.float32)          # beyond both ends of the acceptance table
    assert np.array_equal(conv.EvaluateOnDevice("wavelength_bias", wide)[:, 0].view(np.uint32), capi.eval_medium(T, 4, wide).view(np.uint32))
    if name in ("mie", "lea"):
        # physics: lengths within what the ice model can give, refractive index of ice
        a = conv.EvaluateOnDevice("lengths", wl, layer=n_layers // 2)
        assert 1.0 < a[:, 0].min() and a[:, 0].max() < 1000.0 and 0.1 < a[:, 1].min() and a[:, 1].max() < 200.0
        assert 1.30 < dev[:, 0].min() and dev[:, 0].max() < 1.40


def test_group_velocity_from_the_dispersion_device_equals_oracle():
    """no group refractive index override (CLSIMHIP_REFINDEX_DISPERSION; MediumPropertiesSource.cxx:274-300): device = oracle bit for
    bit, and it is not the override's function"""
    cfg = common.config("mie_dispersion")
    T, conv = common.oracle_tables(cfg), common.product_converter(cfg, 1024)
    wl = np.linspace(266e-9, 674e-9, 4001).astype(np.float32)
    dev = conv.EvaluateOnDevice("refraction", wl)
    assert np.array_equal(dev[:, 0].view(np.uint32), capi.eval_medium(T, 2, wl).view(np.uint32))
    assert np.array_equal(dev[:, 1].view(np.uint32), capi.eval_medium(T, 3, wl).view(np.uint32))
    override = capi.eval_medium(common.oracle_tables(common.config("mie")), 3, wl)
    assert np.all(dev[:, 1] != override) and np.abs(dev[:, 1] / override - 1.0).max() < 0.011
    assert conv.KernelForBunch(1 << 20) in ("pool", "classic") and conv.GetTable("fast_variant")[0] == 0.0


def tilt_independent(medium, pos):
    """the tilt in double precision numpy, stated from ScalarFieldIceTiltZShift.cxx:145-213: bilinear between the dust-logger
    profiles along the tilt direction and in z, the outermost cells extended linearly beyond the table"""
    tl = medium["tilt"]
    d = np.asarray(tl["distances"], dtype=np.float64)
    zc = np.asarray(tl["zcoords"], dtype=np.float64)
    corr = np.asarray(tl["zcorr"], dtype=np.float64).reshape(len(d), len(zc))
    p = pos.astype(np.float64)
    zr = (p[:, 2] - zc[0]) / ((zc[-1] - zc[0]) / (len(zc) - 1))
    k = np.clip(np.floor(zr).astype(int), 0, len(zc) - 2)
    fz = zr - k
    nr = math.cos(tl["azimuth"]) * p[:, 0] + math.sin(tl["azimuth"]) * p[:, 1]
    j = np.clip(np.searchsorted(d, nr, side="right"), 1, len(d) - 1)
    lower = corr[j - 1, k + 1] * fz + corr[j - 1, k] * (1 - fz)
    upper = corr[j, k + 1] * fz + corr[j, k] * (1 - fz)
    f_lower = (d[j] - nr) / (d[j] - d[j - 1])
    return upper * (1 - f_lower) + lower * f_lower


@pytest.mark.parametrize("name", ["mie", "lea"])
def test_tilt_device_equals_oracle_and_an_independent_interpolation(setups, name):
    """testScalarFieldIceTiltZShift.py: positions in a +-1200 m cube, device vs host, tolerance 10 cm"""
    cfg, T, conv = setups[name]
    rng = np.random.default_rng(7)
    pos = rng.uniform(-1200, 1200, (20000, 3)).astype(np.float32)
    for fast in (False, True):
        dev = conv.EvaluateOnDevice("tilt", pos, fast=fast)[:, 0]
        assert np.array_equal(dev.view(np.uint32), capi.eval_field(T, 0, pos).view(np.uint32))
    ref = tilt_independent(cfg["med_o"], pos)
    assert np.abs(dev - ref).max() < 0.10
    near = (np.abs(pos[:, 0]) < 600) & (np.abs(pos[:, 1]) < 600) & (np.abs(pos[:, 2]) < 500)
    assert np.abs(dev[near]).max() < 90.0 and np.abs(dev[near]).max() > 5.0      # tens of metres across the detector


def test_constant_tilt(setups):
    cfg, T, conv = setups["c1"]
    pos = np.random.default_rng(1).uniform(-500, 500, (100, 3)).astype(np.float32)
    assert np.all(conv.EvaluateOnDevice("tilt", pos)[:, 0] == 0.0)               # ScalarFieldConstant(0)
    assert np.all(conv.EvaluateOnDevice("abs_len_scaling", unit_vectors(100, 2))[:, 0] == 1.0)
    d = unit_vectors(100, 3)
    assert np.array_equal(conv.EvaluateOnDevice("pre_scatter_transform", d)[:, :3], d)


def ppc_anisotropy(dirs, azimuth, k1, k2):
    """the formula resources/tests/testScalarFields.py:52-91 tests the device against (its Python port of PPC), vectorised"""
    azx, azy = math.cos(azimuth), math.sin(azimuth)
    k1e, k2e = math.exp(k1), math.exp(k2)
    kz = 1.0 / (k1e * k2e)
    n = dirs.astype(np.float64)
    s1 = (azx * n[:, 0] + azy * n[:, 1]) ** 2
    s2 = (-azy * n[:, 0] + azx * n[:, 1]) ** 2
    s3 = n[:, 2] ** 2
    l1, l2, l3 = k1e * k1e, k2e * k2e, kz * kz
    B2 = 1.0 / l1 + 1.0 / l2 + 1.0 / l3
    nB = s1 / l1 + s2 / l2 + s3 / l3
    An = s1 * l1 + s2 * l2 + s3 * l3
    return 1.0 / ((B2 - nB) * An / 2.0)


def test_anisotropy_device_equals_oracle_and_the_ppc_port(setups):
    """testScalarFields.py: 1e5 random directions, relative deviation <= 1e-5"""
    cfg, T, conv = setups["lea"]
    d = unit_vectors(100000, 11)
    for fast in (False, True):
        dev = conv.EvaluateOnDevice("abs_len_scaling", d, fast=fast)[:, 0]
        assert np.array_equal(dev.view(np.uint32), capi.eval_field(T, 1, d).view(np.uint32))
    an = cfg["med_o"]["aniso"]
    ref = ppc_anisotropy(d, an["azimuth"], an["k1"], an["k2"])
    assert np.abs(dev / ref - 1.0).max() < 1e-5
    assert 0.8 < dev.min() < 1.0 < dev.max() < 1.3


def test_direction_transforms_device_equals_oracle_and_numpy(setups):
    """testVectorTransforms.py: matrix x direction, renormalised, vs numpy"""
    cfg, T, conv = setups["lea"]
    d = unit_vectors(50000, 13)
    for what, code, key in (("pre_scatter_transform", 2, "pre"), ("post_scatter_transform", 3, "post")):
        for fast in (False, True):
            dev = conv.EvaluateOnDevice(what, d, fast=fast)[:, :3]
            assert np.array_equal(dev.view(np.uint32), capi.eval_field(T, code, d).view(np.uint32))
        m = np.asarray(cfg["med_o"][key]["matrix"], dtype=np.float64).reshape(3, 3)
        ref = d.astype(np.float64) @ m.T
        if cfg["med_o"][key]["renormalize"]:
            ref /= np.linalg.norm(ref, axis=1)[:, None]
        assert np.abs(dev - ref).max() < 2e-6
    # the two transforms of SPICE-Lea undo each other (GetSpiceLeaAnisotropyTransforms.py)
    back = conv.EvaluateOnDevice("post_scatter_transform", conv.EvaluateOnDevice("pre_scatter_transform", d)[:, :3])[:, :3]
    assert np.abs(back - d).max() < 2e-6


@pytest.mark.parametrize("name", ["mie", "lea", "photonics_mie", "flasher_led405"])
def test_random_distributions_device_equals_oracle(setups, name):
    """I3CLSimRandomDistributionTester: one stream per work item, N draws each"""
    cfg, T, conv = setups[name]
    x, a = common.streams(512, seed=99)
    for what, gens in (("uniform", (0,)), ("scattering_cosine", (0,)), ("wavelength", range(int(T.t.num_gen)))):
        for g in gens:
            for fast in ((False, True) if name in ("mie", "lea") else (False,)):
                dev, x_dev = conv.SampleOnDevice(what, x, a, 200, generator=g, fast=fast)
                ref, x_ref = capi.sample(T, what, x, a, 200, generator=g)
                assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32)), (what, g, fast)
                assert np.array_equal(x_dev, x_ref)
    u, _ = conv.SampleOnDevice("uniform", x, a, 200)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01
    c, _ = conv.SampleOnDevice("scattering_cosine", x, a, 200)
    mean_cos = 0.9 if name != "photonics_mie" else float(c.mean())
    assert -1.0 <= c.min() and c.max() <= 1.0 and abs(c.mean() - mean_cos) < 0.01         # <cos theta> = 0.9 (cfg.txt), whatever the mixture
    w, _ = conv.SampleOnDevice("wavelength", x, a, 200)
    assert 2.6e-7 <= w.min() and w.max() <= 6.8e-7
    if name == "flasher_led405":
        w1, _ = conv.SampleOnDevice("wavelength", x, a, 200, generator=1)
        assert 3.5e-7 <= w1.min() and w1.max() <= 4.55e-7 and abs(float(np.median(w1)) - 4.05e-7) < 1.0e-8


def test_tester_refusals(setups):
    from clsim_amd import converter as CV
    cfg, T, conv = setups["photonics_mie"]
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="no such layer"):
        conv.EvaluateOnDevice("lengths", [4e-7], layer=10000)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="FAST"):
        conv.EvaluateOnDevice("lengths", [4e-7], fast=True)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="no such wavelength generator"):
        conv.SampleOnDevice("wavelength", [1], [4294967118], 4, generator=5)
