"""The search filter's rules against the ORACLE (which has no filter: every segment goes through the reference's whole search) on
the inputs that sit on their decision boundaries.  The filter (prop_device.hip.h: free_flight_bound, segment_misses_string,
dom_search_needed) skips a search only when it can prove that the search would find nothing; each rule has a margin against
rounding, and these bunches put photons where the margins matter:
  * inside a DOM's oversized sphere at every radius, on its surface to a millimetre, just outside (the rule "a photon that starts
    inside a DOM leaves it", sparse_collision_kernel.c.cl:150-159, and the closest-approach test);
  * at an xy distance from a string axis right at the reach of its DOMs (segment_misses_string: lines tangent to the cylinder),
    on the axes themselves, between two DOMs of a string;
  * with the DOM centre exactly behind, beside or ahead of the photon (urdot = 0 to rounding);
  * flasher steps and cascade steps in the same bunch through the flasher instantiations."""
import numpy as np
import pytest

from clsim_amd import synthetic as S
from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu


def boundary_steps(cfg, n, seed, reach, photons=24):
    g = cfg["geom"]
    rng = np.random.Generator(np.random.PCG64(seed))
    steps = S.cascade_steps(n, seed=seed, photons_per_step=photons)
    dom = rng.integers(0, len(g["x"]), size=n)
    kind = rng.integers(0, 8, size=n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    R = float(g["om_radius"])
    radius = np.select([kind == 0, kind == 1, kind == 2, kind == 3],
                       [R * rng.random(n) ** (1 / 3.0),                    # anywhere inside the sphere
                        R + rng.normal(0, 1e-3, n),                        # on the surface, +-1 mm
                        R + 0.01 + 0.08 * rng.random(n),                   # within the closest-approach margin and just beyond
                        0.0], default=-1.0)                                # the centre itself
    at_dom = radius >= 0
    x = np.where(at_dom, g["x"][dom] + u[:, 0] * radius, 0.0)
    y = np.where(at_dom, g["y"][dom] + u[:, 1] * radius, 0.0)
    z = np.where(at_dom, g["z"][dom] + u[:, 2] * radius, 0.0)
    # strings: axis = mean DOM position of the string (GeometrySource.cxx:1153-1269)
    ids = np.unique(g["string_ids"])
    sidx = np.searchsorted(ids, g["string_ids"])
    ax = np.array([g["x"][sidx == s].mean() for s in range(len(ids))])
    ay = np.array([g["y"][sidx == s].mean() for s in range(len(ids))])
    s = rng.integers(0, len(ids), size=n)
    phi = rng.uniform(0, 2 * np.pi, n)
    rho = np.select([kind == 4, kind == 5, kind == 6], [reach + rng.normal(0, 0.02, n), rng.uniform(0, 2 * reach, n), 0.0], default=rng.uniform(0, 60.0, n))
    zz = rng.uniform(-520, 520, n)
    x = np.where(at_dom, x, ax[s] + rho * np.cos(phi))
    y = np.where(at_dom, y, ay[s] + rho * np.sin(phi))
    z = np.where(at_dom, z, zz)
    steps["x"], steps["y"], steps["z"] = x.astype(np.float32), y.astype(np.float32), z.astype(np.float32)
    # directions: a third of the steps aimed exactly at / away from / across their DOM or string axis (theta, phi of the step; the
    # Cherenkov cone then spreads the photons around it)
    aim = rng.integers(0, 6, size=n)
    tx = np.where(at_dom, g["x"][dom], ax[s]) - x
    ty = np.where(at_dom, g["y"][dom], ay[s]) - y
    tz = np.where(at_dom, g["z"][dom] - z, 0.0)
    norm = np.sqrt(tx * tx + ty * ty + tz * tz)
    ok = (norm > 1e-6) & (aim < 3)
    sign = np.where(aim == 1, -1.0, 1.0)
    dx, dy, dz = sign * tx / np.maximum(norm, 1e-9), sign * ty / np.maximum(norm, 1e-9), sign * tz / np.maximum(norm, 1e-9)
    across = aim == 2                                       # perpendicular in xy: tangent lines
    dx, dy = np.where(across, -ty / np.maximum(norm, 1e-9), dx), np.where(across, tx / np.maximum(norm, 1e-9), dy)
    theta = np.arccos(np.clip(dz, -1, 1))
    ph = np.arctan2(dy, dx) % (2 * np.pi)
    steps["theta"] = np.where(ok, theta, steps["theta"]).astype(np.float32)
    steps["phi"] = np.where(ok, ph, steps["phi"]).astype(np.float32)
    return steps


@pytest.mark.parametrize("name", ["mie", "lea", "flasher", "mie_regular", "clear"])
def test_filter_rules_on_their_boundaries(name):
    cfg = common.config(name)
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.Compile()
    reach = float(conv.GetTable("STRING_PROXIMITY_GRID")[5])
    del conv
    n = 1 << 17
    steps = boundary_steps(cfg, n, seed=31, reach=reach)
    if cfg["flasher"]:
        # every other step a flasher step (generator 1, no Cherenkov cone: the photons leave along the step direction itself)
        steps["sourceType"][::2] = 1
        steps["length"][::2] = 0.0
    x, a = common.streams(n)
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=16)
    assert cnt_o > 20000                                     # a third of the photons start at a DOM
    conv = common.product_converter(cfg, n)
    conv.EnqueueSteps(steps, 3)
    ident, ph_p = conv.GetConversionResult()
    assert len(ph_p) == cnt_o
    expect = capi.replace_indices_with_ids(ph_o, T.geo)
    assert common.sort_photons(ph_p).tobytes() == common.sort_photons(expect).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)


def test_filter_rules_without_stopping_detected_photons():
    cfg = common.config("mie")
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.Compile()
    reach = float(conv.GetTable("STRING_PROXIMITY_GRID")[5])
    del conv
    n = 1 << 16
    steps = boundary_steps(cfg, n, seed=32, reach=reach)
    x, a = common.streams(n)
    T = common.oracle_tables(cfg, stop_detected=False)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=16)
    conv = common.product_converter(cfg, n, stop_detected=False)
    conv.EnqueueSteps(steps, 3)
    ident, ph_p = conv.GetConversionResult()
    assert len(ph_p) == cnt_o and cnt_o > 10000
    assert common.sort_photons(ph_p).tobytes() == common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)


@pytest.mark.parametrize("stop", [True, False])
def test_pooled_kernel_on_the_boundaries_with_the_aim_level_on_and_off(monkeypatch, stop):
    """the same boundary steps through the POOLED kernel (its filter reads k_aim / k_wait from the launch parameters) with the
    string-aimed level at its default, switched off (CLSIMHIP_K_AIM=0) and with parked lanes never waiting (CLSIMHIP_K_WAIT=0):
    every setting must give the oracle's records -- the filter is conservative, no result depends on it (a filter-off
    point in the bit-identical tests); with and without STOP_PHOTONS_ON_DETECTION"""
    cfg = common.config("mie")
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.Compile()
    reach = float(conv.GetTable("STRING_PROXIMITY_GRID")[5])
    del conv
    n = 1 << 16
    steps = boundary_steps(cfg, n, seed=33, reach=reach)
    x, a = common.streams(n)
    T = common.oracle_tables(cfg, stop_detected=stop)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=16)
    expect = common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes()
    assert cnt_o > 10000
    monkeypatch.setenv("CLSIMHIP_KERNEL", "pool")
    for env in ({}, {"CLSIMHIP_K_AIM": "0"}, {"CLSIMHIP_K_AIM": "64", "CLSIMHIP_K_WAIT": "0"}):
        for k in ("CLSIMHIP_K_AIM", "CLSIMHIP_K_WAIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        conv = common.product_converter(cfg, n, stop_detected=stop)
        assert conv.UsesPooledKernel()
        conv.EnqueueSteps(steps, 3)
        _, ph_p = conv.GetConversionResult()
        assert len(ph_p) == cnt_o, env
        assert common.sort_photons(ph_p).tobytes() == expect, env
        assert np.array_equal(conv.GetRNGState(n), x_o), env
        del conv
