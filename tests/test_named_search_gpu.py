"""GPU: the DOM search confined to the one DOM the proximity map names (prop_device.hip.h: find_collision_named) against the
full search (find_collision: sparse_collision_kernel.c.cl:27-587) on whole production bunches.

The third filter level knows that only one DOM is within reach of a step; the confined search evaluates, for that DOM alone,
every pruning decision the reference's search would take on its way to it (cell range of the string's subdetector, the
string's tests, the z layers that name the DOM, the sphere test).  It is correct iff it returns what the full search returns
on every input: both are run here on the same bunches -- the C5 flasher bunch (photons born at a DOM: a search on every
other trip), the C2 and C3 cascade bunches, a bunch of steps placed right at string axes and cell borders -- and compared
through the multiset of all 80-byte records, the hit count and every final RNG state.  (Both also equal the oracle on the
4096-step bunches of tests/test_parity_gpu.py, which run with the confined search on.)

CLSIMHIP_NO_NAMED_SEARCH=1 at Compile() marks every DOM "not nameable": the kernel then takes the full search everywhere."""
import os

import numpy as np
import pytest
import torch

from clsim_amd import synthetic as S
from tests import common
from tests.test_production_size_gpu import big_run, multiset_checksum

pytestmark = pytest.mark.gpu


def run_both(cfg, steps, capacity):
    n = len(steps)
    dev = torch.device("cuda", 0)
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    out = []
    for no_named in ("0", "1"):
        os.environ["CLSIMHIP_NO_NAMED_SEARCH"] = no_named
        try:
            conv = common.product_converter(cfg, n)
        finally:
            del os.environ["CLSIMHIP_NO_NAMED_SEARCH"]
        named = conv.GetTable("dom_named").astype(np.uint64).reshape(-1, 4)
        if no_named == "1":
            assert np.all(named[:, 0] == 0xffffffff)
        else:
            assert np.all(named[:, 0] != 0xffffffff), "every DOM of the synthetic detector can be named"
        rec, cnt = big_run(conv, d_steps, n, capacity)
        out.append((cnt, multiset_checksum(rec), conv.GetRNGState(n).copy()))
        del rec, conv
        torch.cuda.empty_cache()
    (cnt_a, sum_a, x_a), (cnt_b, sum_b, x_b) = out
    assert cnt_a == cnt_b, (cnt_a, cnt_b)
    assert sum_a == sum_b
    assert np.array_equal(x_a, x_b)
    return cnt_a


@pytest.mark.timeout(900)
def test_flasher_production_bunch_named_equals_full_search():
    cfg = common.config("flasher")
    g = cfg["geom"]
    k = int(np.argmin(np.abs(g["x"]) + np.abs(g["y"]) + np.abs(g["z"] + 100.0)))
    steps = S.flasher_steps(2621440, seed=1000, photons_per_step=400, position=(float(g["x"][k]), float(g["y"][k]), float(g["z"][k])))
    hits = run_both(cfg, steps, 48 << 20)
    assert hits > 10 ** 7


@pytest.mark.timeout(900)
@pytest.mark.parametrize("ice", ["mie", "flasher"])
def test_cascade_bunch_named_equals_full_search(ice):
    """cascade steps spread over the detector; "flasher" = SPICE-Lea with the second wavelength generator, whose kernel
    instantiations use the confined search ("mie": the switch must not change anything where it is not used)"""
    cfg = common.config(ice)
    steps = S.cascade_steps(1 << 20, seed=1000, photons_per_step=200)
    hits = run_both(cfg, steps, 8 << 20)
    assert hits > 10 ** 5


@pytest.mark.timeout(900)
def test_steps_at_strings_and_cell_borders_named_equals_full_search():
    """the inputs where the reference's own pruning matters: photons born on string axes, between two DOMs of a string, at
    DOM surfaces, and along the borders of the xy cell grids (a DOM sphere that reaches into the neighbouring cell is not
    found from there by the reference; the confined search must miss it as well).  Cascade steps through the flasher
    configuration's kernel: the instantiations that use the confined search"""
    cfg = common.config("flasher")
    g = cfg["geom"]
    rng = np.random.Generator(np.random.PCG64(77))
    n = 1 << 19
    steps = S.cascade_steps(n, seed=5, photons_per_step=100)
    pick = rng.integers(0, len(g["x"]), size=n)
    kind = rng.integers(0, 4, size=n)
    off = rng.normal(0.0, 1.0, size=(n, 3))
    scale = np.choose(kind, [0.05, 0.9, 3.0, 9.0])                          # inside the sphere, at its surface, near, between DOMs
    steps["x"] = (g["x"][pick] + off[:, 0] * scale).astype(np.float32)
    steps["y"] = (g["y"][pick] + off[:, 1] * scale).astype(np.float32)
    steps["z"] = (g["z"][pick] + off[:, 2] * scale + np.where(kind == 3, 8.5, 0.0)).astype(np.float32)
    # a quarter of the steps on the lines of the cell grids instead (GEO_CELL_k: nx, ny, width x, width y, start x, start y)
    conv = common.product_converter(cfg, 512)
    k = 0
    lines = []
    while True:
        try:
            nx, ny, wx, wy, sx, sy = conv.GetTable("GEO_CELL_%d" % k)
        except Exception:
            break
        lines.append((int(nx), int(ny), wx, wy, sx, sy))
        k += 1
    del conv
    assert lines
    m = n // 4
    which = rng.integers(0, len(lines), size=m)
    for i in range(m):
        nx, ny, wx, wy, sx, sy = lines[which[i]]
        if i & 1:
            steps["x"][i] = np.float32(sx + wx * rng.integers(0, nx + 1) + rng.normal(0, 0.3))
            steps["y"][i] = np.float32(sy + wy * ny * rng.random())
        else:
            steps["y"][i] = np.float32(sy + wy * rng.integers(0, ny + 1) + rng.normal(0, 0.3))
            steps["x"][i] = np.float32(sx + wx * nx * rng.random())
    hits = run_both(cfg, steps, 16 << 20)
    assert hits > 10 ** 5
