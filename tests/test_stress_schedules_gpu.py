"""tools/stress_schedules.py inside the driver-run suite: eight seeded cases -- random bunch sizes, photon counts
per step (empty steps, single photons, steps of 3000), grids, slice counts, batching thresholds, ring sizes, both schedulings,
specialised and generic instantiations, the four media -- each compared with the SAME bunch run as whole steps on a small grid AND
with the oracle: hit multiset and final RNG states, bit for bit (propagation_kernel.c.cl:458-461, 911-912: a step's draws come
from its own stream in its own order, whatever lane runs it)."""
import numpy as np
import pytest

from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu

KNOBS = ("CLSIMHIP_GRID", "CLSIMHIP_SLICES", "CLSIMHIP_K_NEW", "CLSIMHIP_K_SEARCH", "CLSIMHIP_KERNEL", "CLSIMHIP_POOL_R", "CLSIMHIP_K_POP", "CLSIMHIP_NO_FAST")


def run(cfg, steps, env, monkeypatch, keep):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    n = len(steps)
    conv = common.product_converter(cfg, n, stop_detected=not keep)
    conv.EnqueueSteps(steps, 3)
    _, ph = conv.GetConversionResult()
    return common.sort_photons(ph).tobytes(), conv.GetRNGState(n), len(ph)


@pytest.mark.parametrize("case", range(8))
def test_schedule_does_not_change_results(case, monkeypatch):
    rng = np.random.Generator(np.random.PCG64(7000 + case))
    keep = case >= 6                                # two cases without STOP_PHOTONS_ON_DETECTION
    name = (["mie", "lea", "clear", "flasher"] if keep else ["mie", "lea", "c1", "flasher"])[case % 4]
    cfg = common.config(name)
    n = 256 * int(rng.integers(1, [40, 200, 400][case % 3]))
    steps = common.steps_for(cfg, n, seed=100 + case)
    mode = case % 5
    if mode == 0:
        steps["num"] = rng.choice([0, 1, 2, 7, 63, 64, 65, 200, 399, 1500], size=n).astype(np.uint32)
    elif mode == 1:
        steps["num"] = rng.integers(0, 60, n).astype(np.uint32)
    elif mode == 2:
        steps["num"] = 0
        steps["num"][rng.integers(0, n, 5)] = 3000
    env = dict(CLSIMHIP_GRID=int(rng.integers(1, 1793)), CLSIMHIP_SLICES=int(rng.choice([1, 2, 3, 5, 16, 33, 64])),
               CLSIMHIP_K_NEW=int(rng.choice([1, 4, 12, 40, 64])), CLSIMHIP_K_SEARCH=int(rng.choice([1, 3, 5, 20])))
    if case % 2:        # the pooled kernel: ring size, service threshold, specialised or generic instantiation
        env.update(CLSIMHIP_KERNEL="pool", CLSIMHIP_POOL_R=int(rng.choice([4, 7, 16, 34, 45])), CLSIMHIP_K_POP=int(rng.choice([1, 4, 17, 64])),
                   CLSIMHIP_NO_FAST=int(rng.integers(0, 2)), CLSIMHIP_GRID=int(rng.integers(1, 513)))
    else:
        env.update(CLSIMHIP_KERNEL="classic")
    ref = run(cfg, steps, dict(CLSIMHIP_GRID=64, CLSIMHIP_SLICES=1, CLSIMHIP_K_NEW=1, CLSIMHIP_K_SEARCH=1, CLSIMHIP_KERNEL="classic"), monkeypatch, keep)
    got = run(cfg, steps, env, monkeypatch, keep)
    assert got[2] == ref[2] and got[0] == ref[0] and np.array_equal(got[1], ref[1]), (case, name, n, env)
    x, a = common.streams(n)
    T = common.oracle_tables(cfg, stop_detected=not keep)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=16)
    assert cnt_o == got[2]
    assert common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes() == got[0]
    assert np.array_equal(got[1], x_o)
