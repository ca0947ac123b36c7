"""Flasher step producer: bunch plan and time delay tables of the product's C++ against the
restatement of I3CLSimLightSourceToStepConverterFlasher::MakeSteps and of I3CLSimRandomValueIceCubeFlasherTimeProfile
in oracle/builders.py (CPU); the GPU kernel against oracle/stepgen_oracle.c bit for bit, and the distributions it
produces against the closed forms of the reference's formulas (GPU)."""
import math

import numpy as np
import pytest

from clsim_amd import converter as CV
from clsim_amd.synthetic import STEP_DTYPE
from oracle import builders as B
from oracle import capi

NS = 1.0


def led_config(**kw):
    """python/GetFlasherParameterizationList.py:58-79: LED flashers."""
    return (CV.FlasherStepConverterConfig((CV.DIST_NORMAL, 0.0), (CV.DIST_NORMAL, 0.0), (CV.DIST_FLASHER_TIME_PROFILE, 0.0), False, **kw),
            capi.flasher_config(("normal", 0.0), ("normal", 0.0), ("flasher_time_profile", 0.0), False,
                                **{{"photonsPerStep": "photons_per_step", "maxBunchSize": "max_bunch_size", "bunchSizeGranularity": "granularity"}[k]: v
                                   for k, v in kw.items()}))


def candle_config(**kw):
    """:61-65, 82-91: standard candles (constant polar angle, uniform azimuth, Gaussian delay around 2 ns), polar mode."""
    return (CV.FlasherStepConverterConfig((CV.DIST_CONSTANT, 0.0), (CV.DIST_UNIFORM, 0.0), (CV.DIST_NORMAL, 2.0 * NS), True, **kw),
            capi.flasher_config(("constant", 0.0), ("uniform", 0.0), ("normal", 2.0 * NS), True,
                                **{{"photonsPerStep": "photons_per_step", "maxBunchSize": "max_bunch_size", "bunchSizeGranularity": "granularity"}[k]: v
                                   for k, v in kw.items()}))


def pulses(n, seed=1, width=35.0, photons=(1, 3000000)):
    rng = np.random.Generator(np.random.PCG64(seed))
    q = np.zeros(n, dtype=CV.FLASHER_REQUEST_DTYPE)
    q["x"], q["y"], q["z"] = rng.uniform(-500, 500, (3, n)).astype(np.float32)
    q["time"] = rng.uniform(0, 1000, n)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    q["dx"], q["dy"], q["dz"] = d.T.astype(np.float32)
    q["sigma_polar"], q["sigma_azimuthal"] = np.radians(9.7), np.radians(9.8)
    q["pulse_width"] = width
    q["identifier"] = np.arange(n) + 17
    q["source_type"] = 1 + np.arange(n) % 5
    q["num_photons_with_bias"] = rng.integers(photons[0], photons[1], n)
    return q


def test_time_profile_tables_follow_the_reference_python():
    """_the_pulse for narrow (FB_WIDTH <= 15) and wide pulses, then InterpolatedDistribution::InitTables."""
    for width in (3.5, 7.5, 10.0, 35.0, 63.5):
        y = B.flasher_time_profile(width)
        assert y.shape == (240,) and y.min() >= 0 and 0.9 < y.max() <= 1.0
        dens_o, cum_o = B.interpolated_distribution_tables(0.5, y)
        dens_p, cum_p = CV.FlasherTimeProfile(width)
        assert np.array_equal(dens_p, dens_o) and np.array_equal(cum_p, cum_o)
        assert cum_p[0] == 0.0 and cum_p[-1] == 1.0 and np.all(np.diff(cum_p) >= 0)
    # the wide pulse has its plateau: FB_WIDTH = 70 -> rising edge 12.76 ns, plateau 30.0 ns
    y = B.flasher_time_profile(35.0)
    x = np.linspace(0, 120, 240, endpoint=False)
    rising = math.log(70 - 12.0) * 1.91 + 5.0
    plateau = (70 - 15.0) * 59.5 / 109.0
    assert np.all(y[(x > rising) & (x <= rising + plateau)] == 1.0) and y[0] == 0.0 and y[-1] < 0.01
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception):
        CV.FlasherTimeProfile(0.0)


@pytest.mark.parametrize("pps,max_bunch,gran", [(400, 512000, 512), (400, 1024, 256), (7, 64, 64), (10, 20, 1)])
def test_bunch_plan_is_make_steps(pps, max_bunch, gran):
    cfg_p, cfg_o = led_config(photonsPerStep=pps, maxBunchSize=max_bunch, bunchSizeGranularity=gran)
    q = pulses(40, seed=3, photons=(1, 40 * pps * 50))
    # the cases MakeSteps distinguishes: no photons, one step, evenly divisible (loses a step), whole results only
    q["num_photons_with_bias"][:8] = [0, 1, pps, pps + 1, 2 * pps, 5 * pps, max_bunch * pps, max_bunch * pps + 3 * pps]
    total_p, real_p = CV.CountFlasherSteps(cfg_p, q)
    plan, total_o, _, counts = capi.plan_flasher_steps(cfg_o, q)
    assert total_p == total_o == sum(len(c) for c in counts) and real_p == int(plan["n_real"].sum())
    assert [sum(c) for c in counts[:8]] == [0, 1, pps, pps + 1, pps, 4 * pps, max_bunch * pps, max_bunch * pps + 2 * pps]
    assert all(len(c) % gran == 0 or len(c) >= max_bunch for c in counts)
    for bad in (dict(photonsPerStep=0), dict(maxBunchSize=100, bunchSizeGranularity=64)):
        with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception):
            CV.CountFlasherSteps(led_config(**bad)[0], q)
    q["dx"][3] = float("nan")
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception):
        CV.CountFlasherSteps(cfg_p, q)


def test_oracle_steps_have_the_reference_structure():
    cfg_p, cfg_o = led_config(photonsPerStep=400, maxBunchSize=2048, bunchSizeGranularity=64)
    q = pulses(6, seed=5, photons=(1000, 400000))
    steps = capi.generate_flasher_steps(cfg_o, q, seed=99)
    plan, total, _, counts = capi.plan_flasher_steps(cfg_o, q)
    assert len(steps) == total
    for i, c in enumerate(counts):
        s = steps[int(plan["first_out"][i]):int(plan["first_out"][i]) + len(c)]
        assert np.array_equal(s["num"], np.array(sorted(c, key=lambda v: v == 0), dtype=np.uint32))      # real steps, then dummies
        real = s[s["num"] > 0]
        assert np.all(real["x"] == q["x"][i]) and np.all(real["length"] == 0) and np.all(real["beta"] == 1) and np.all(real["weight"] == 1)
        assert np.all(real["sourceType"] == q["source_type"][i]) and np.all(s["id"] == q["identifier"][i])
        dummy = s[s["num"] == 0]
        assert np.all(dummy["weight"] == 0) and np.all(dummy["theta"] == 0) and np.all(dummy["sourceType"] == 0)
        assert np.all(real["t"] >= q["time"][i]) and np.all(real["t"] <= q["time"][i] + 120.0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["led", "led_narrow", "candle"])
def test_gpu_steps_equal_the_oracle(kind):
    if kind == "candle":
        cfg_p, cfg_o = candle_config(photonsPerStep=400, maxBunchSize=4096, bunchSizeGranularity=256)
        q = pulses(12, seed=8)
        q["sigma_polar"] = np.radians(41.13)      # the standard candle's cone
        q["sigma_azimuthal"] = 2 * np.pi
        q["pulse_width"] = 1.0                    # sigma of the delay
        q["dx"][0], q["dy"][0], q["dz"][0] = 0.0, 0.0, -1.0       # vertical: the "trivial" branch
    else:
        cfg_p, cfg_o = led_config(photonsPerStep=400, maxBunchSize=4096, bunchSizeGranularity=256)
        q = pulses(12, seed=9, width=(5.0 if kind == "led_narrow" else 35.0))
        q["pulse_width"][::3] = 63.5              # several time profiles in one call
    exp = capi.generate_flasher_steps(cfg_o, q, seed=4242)
    got = CV.GenerateFlasherSteps(cfg_p, q, seed=4242)
    assert len(got) == len(exp) > 10000 and got.tobytes() == exp.tobytes()
    assert int(got["num"].sum()) == sum(sum(c) for c in capi.plan_flasher_steps(cfg_o, q)[3])


@pytest.mark.gpu
def test_gpu_step_distributions():
    """Smearing and delays follow the reference's formulas: Gaussian polar/azimuthal offsets about the pulse direction
    (horizontal-plane interpretation), delays distributed like the LED pulse shape."""
    cfg_p, _ = led_config(photonsPerStep=10, maxBunchSize=1 << 20, bunchSizeGranularity=1)
    q = pulses(1, seed=2, width=35.0)
    th0, ph0 = 1.1, 2.3
    q["dx"], q["dy"], q["dz"] = math.sin(th0) * math.cos(ph0), math.sin(th0) * math.sin(ph0), math.cos(th0)
    q["num_photons_with_bias"] = 10 * 400000 + 5     # (an exact multiple would lose its last step, see the plan test)
    s = CV.GenerateFlasherSteps(cfg_p, q, seed=7)
    assert len(s) == 400001 and np.all(s["num"][:-1] == 10) and s["num"][-1] == 5
    s = s[:-1]
    from scipy import stats
    dpol = (s["theta"].astype(np.float64) - th0) / float(q["sigma_polar"][0])
    dazi = (s["phi"].astype(np.float64) - ph0) / float(q["sigma_azimuthal"][0])
    assert stats.kstest(dpol[::7], "norm").pvalue > 1e-3 and stats.kstest(dazi[::7], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(dpol, dazi)[0, 1]) < 0.01
    # delays: CDF against the tabulated cumulative distribution
    dens, cum = CV.FlasherTimeProfile(35.0)
    delay = s["t"].astype(np.float64) - float(q["time"][0])
    x = np.arange(240) * 0.5
    emp = np.searchsorted(np.sort(delay), x[1:], side="right") / len(delay)
    assert np.max(np.abs(emp - cum[1:])) < 5e-3
    # polar interpretation: the angle to the pulse direction is |N(0, sigma)|, the orientation around it is N(0, sigma) too
    cfg_c, _ = candle_config(photonsPerStep=10, maxBunchSize=1 << 20, bunchSizeGranularity=1)
    q["sigma_polar"], q["sigma_azimuthal"], q["pulse_width"] = 0.7, 2 * np.pi, 1.5
    s = CV.GenerateFlasherSteps(cfg_c, q, seed=8)[:-1]
    d = np.stack([np.sin(s["theta"]) * np.cos(s["phi"]), np.sin(s["theta"]) * np.sin(s["phi"]), np.cos(s["theta"])], axis=1).astype(np.float64)
    d0 = np.array([float(q["dx"][0]), float(q["dy"][0]), float(q["dz"][0])])
    assert np.allclose(np.arccos(np.clip(d @ d0, -1, 1)), 0.7, atol=2e-3)                   # constant opening angle
    assert stats.kstest(((s["t"].astype(np.float64) - float(q["time"][0])) - 2.0) / 1.5, "norm").pvalue > 1e-3
