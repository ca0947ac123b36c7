"""GPU parity: the HIP propagator (through the C ABI) against the CPU oracle on
identical steps and RNG streams.  Bar: the sorted multiset of 80-byte photon
records is BIT-IDENTICAL (hit count, string/DOM IDs, scatter counts and every
float), and so are the RNG state words left behind."""
import numpy as np
import pytest

from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu


def run_both(name, n_steps, max_items=None, seed=3, threads=8):
    cfg = common.config(name)
    steps = common.steps_for(cfg, n_steps, seed=seed)
    n = len(steps)
    max_items = max_items or n
    x, a = common.streams(max_items)
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=threads)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    conv = common.product_converter(cfg, max_items)
    conv.EnqueueSteps(steps, 77)
    ident, ph_p = conv.GetConversionResult()
    assert ident == 77
    x_p = conv.GetRNGState(n)
    return steps, (ph_o, cnt_o, x_o), (ph_p, x_p), conv


@pytest.mark.parametrize("name,n_steps", [("c1", 1000), ("mie", 4096), ("lea", 4096), ("flasher", 2048)])
def test_hit_multiset_bit_exact(name, n_steps):
    steps, (ph_o, cnt_o, x_o), (ph_p, x_p), conv = run_both(name, n_steps)
    assert cnt_o > 10, "workload too small to be a test"
    assert len(ph_p) == cnt_o
    so, sp = common.sort_photons(ph_o), common.sort_photons(ph_p)
    assert np.array_equal(so["stringID"], sp["stringID"]) and np.array_equal(so["omID"], sp["omID"])
    assert so.tobytes() == sp.tobytes()
    assert np.array_equal(x_o, x_p)
    st = conv.GetStatistics()
    assert st["TotalNumPhotonsGenerated"] == float(steps["num"].sum())
    assert st["TotalNumPhotonsAtDOMs"] == float(cnt_o)
    assert st["NumKernelCalls"] == 1.0


def test_streams_persist_across_bunches():
    """RNG stream i belongs to step slot i and carries over to the next bunch
    (propagation_kernel.c.cl:458-461, 911-912)."""
    cfg = common.config("mie")
    steps = common.steps_for(cfg, 1024, seed=9)
    x, a = common.streams(1024)
    T = common.oracle_tables(cfg)
    conv = common.product_converter(cfg, 1024)
    xo = x
    for bunch in range(3):
        ph_o, cnt_o, xo, _ = capi.propagate(T, steps, xo, a, threads=8)
        conv.EnqueueSteps(steps, bunch)
        ident, ph_p = conv.GetConversionResult()
        assert ident == bunch
        ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
        assert common.sort_photons(ph_o).tobytes() == common.sort_photons(ph_p).tobytes()
    assert np.array_equal(conv.GetRNGState(1024), xo)


def test_output_overflow_truncates_like_reference():
    """The hit counter keeps counting past the buffer (propagation_kernel.c.cl:329-330);
    the host logs and truncates (OpenCL.cxx:1027-1032).  A flasher 1 m from a DOM
    overflows the reference-sized buffer of 10 x maxNumWorkitems records."""
    cfg = common.config("flasher")
    g = cfg["geom"]
    k = 30 * 60 + 29
    from clsim_amd import synthetic as S
    steps = S.flasher_steps(512, seed=3, position=(g["x"][k] + 1.0, g["y"][k], g["z"][k]), pad_to=256)
    x, a = common.streams(len(steps))
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, _, _ = capi.propagate(T, steps, x, a, threads=8)
    assert cnt_o > 10 * len(steps)
    conv = common.product_converter(cfg, len(steps))
    conv.EnqueueSteps(steps, 5)
    ident, ph_p = conv.GetConversionResult()
    assert len(ph_p) == 10 * len(steps)
    assert conv.GetStatistics()["TotalNumPhotonsAtDOMs"] == float(10 * len(steps))
    # every stored record is one of the oracle's records
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    raw_o = ph_o.tobytes()
    have = set(raw_o[i:i + 80] for i in range(0, len(raw_o), 80))
    raw = ph_p.tobytes()
    assert all(raw[i:i + 80] in have for i in range(0, len(raw), 80))
