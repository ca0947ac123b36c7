"""GPU parity of the pooled propagation kernel (clsim_amd/csrc/prop_pool_kernel.hip, CLSIMHIP_KERNEL=pool): same bar as
tests/test_parity_gpu.py -- the sorted multiset of 80-byte photon records and the RNG state words left behind are
BIT-IDENTICAL to the CPU oracle's -- for every ring size, service threshold and creation batch, and equal to the classic
kernel's output on bunches too large for the oracle."""
import numpy as np
import pytest

from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu


@pytest.fixture
def pooled(monkeypatch):
    monkeypatch.setenv("CLSIMHIP_KERNEL", "pool")
    for k in ("CLSIMHIP_POOL_R", "CLSIMHIP_K_POP", "CLSIMHIP_K_NEW", "CLSIMHIP_SLICES", "CLSIMHIP_K_SEARCH", "CLSIMHIP_GRID"):
        monkeypatch.delenv(k, raising=False)
    return monkeypatch


@pytest.mark.parametrize("name,n_steps", [("c1", 1000), ("mie", 4096), ("lea", 4096), ("flasher", 2048),
                                          ("photonics_mie", 4096), ("photonics_wham", 2048)])
def test_pooled_hit_multiset_bit_exact(pooled, name, n_steps):
    cfg = common.config(name)
    steps = common.steps_for(cfg, n_steps, seed=3)
    n = len(steps)
    x, a = common.streams(n)
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    conv = common.product_converter(cfg, n)
    assert conv.UsesPooledKernel()
    xo = x_o
    conv.EnqueueSteps(steps, 77)
    ident, ph_p = conv.GetConversionResult()
    assert ident == 77 and cnt_o > 10 and len(ph_p) == cnt_o
    assert common.sort_photons(ph_o).tobytes() == common.sort_photons(ph_p).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)
    # the streams carry over to the next bunch (propagation_kernel.c.cl:458-461, 911-912)
    ph_o2, cnt_o2, x_o2, _ = capi.propagate(T, steps, xo, a, threads=8)
    conv.EnqueueSteps(steps, 78)
    _, ph_p2 = conv.GetConversionResult()
    assert common.sort_photons(capi.replace_indices_with_ids(ph_o2, T.geo)).tobytes() == common.sort_photons(ph_p2).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o2)


@pytest.mark.parametrize("ring,k_pop,k_new,slices,k_search", [(4, 1, 1, 1, 1), (8, 64, 8, 3, 5), (34, 4, 30, 16, 5), (17, 2, 64, 7, 64),
                                                               (64, 16, 1, 64, 2), (33, 7, 12, 16, 13),
                                                               # a ring below the smallest the kernel runs with (4) is raised to it:
                                                               # round 2 failed the launch and with it the converter
                                                               (1, 4, 1, 16, 3), (3, 2, 3, 5, 2)])
def test_pooled_ragged_bunch_under_every_schedule(pooled, ring, k_pop, k_new, slices, k_search):
    """Pool sizes and thresholds must not change results: a bunch whose steps hold 0 ... 1500 photons (empty steps,
    single photons, steps much longer than a slice), two bunches in a row."""
    pooled.setenv("CLSIMHIP_POOL_R", str(ring)); pooled.setenv("CLSIMHIP_K_POP", str(k_pop)); pooled.setenv("CLSIMHIP_K_NEW", str(k_new))
    pooled.setenv("CLSIMHIP_SLICES", str(slices)); pooled.setenv("CLSIMHIP_K_SEARCH", str(k_search))
    cfg = common.config("mie")
    n = 2048
    steps = common.steps_for(cfg, n, seed=17)
    rng = np.random.default_rng(5)
    num = rng.integers(0, 400, n)
    num[rng.random(n) < 0.15] = 0
    num[rng.random(n) < 0.05] = 1
    num[rng.random(n) < 0.02] = 1500
    steps["num"] = num
    x, a = common.streams(n)
    T = common.oracle_tables(cfg)
    conv = common.product_converter(cfg, n)
    assert conv.UsesPooledKernel()
    xo = x
    for bunch in range(2):
        ph_o, cnt_o, xo, _ = capi.propagate(T, steps, xo, a, threads=8)
        conv.EnqueueSteps(steps, bunch)
        _, ph_p = conv.GetConversionResult()
        assert len(ph_p) == cnt_o and cnt_o > 50
        assert common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes() == common.sort_photons(ph_p).tobytes()
        assert np.array_equal(conv.GetRNGState(n), xo)


@pytest.mark.parametrize("name,n_steps", [("mie", 131072), ("lea", 65536), ("flasher", 32768)])
def test_pooled_equals_classic_kernel_on_large_bunches(pooled, name, n_steps):
    """Enough steps to fill every wave's pool on the whole chip (slices, hand-offs between XCDs, full rings): the
    pooled and the classic kernel produce the same multiset and the same stream states."""
    cfg = common.config(name)
    steps = common.steps_for(cfg, n_steps, seed=23)
    n = len(steps)
    results = []
    for kernel in ("classic", "pool"):
        pooled.setenv("CLSIMHIP_KERNEL", kernel)
        conv = common.product_converter(cfg, n)
        assert conv.UsesPooledKernel() == (kernel == "pool")
        conv.EnqueueSteps(steps, 5)
        _, ph = conv.GetConversionResult()
        results.append((common.sort_photons(ph).tobytes(), conv.GetRNGState(n)))
        del conv
    assert len(results[0][0]) > 80 * 50
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1])


@pytest.mark.parametrize("name,n_steps", [("mie", 65536), ("lea", 32768), ("flasher", 16384)])
def test_specialised_and_generic_instantiations_agree(pooled, name, n_steps, monkeypatch):
    """KVariant::fast: the instantiation with the wave-uniform proof tests compiled out against the generic one
    (CLSIMHIP_NO_FAST=1) on the same bunch"""
    cfg = common.config(name)
    steps = common.steps_for(cfg, n_steps, seed=31)
    out = []
    for no_fast in ("0", "1"):
        monkeypatch.setenv("CLSIMHIP_NO_FAST", no_fast)
        conv = common.product_converter(cfg, len(steps))
        assert int(conv.GetTable("fast_variant")[0]) == 1
        conv.EnqueueSteps(steps, 1)
        _, ph = conv.GetConversionResult()
        out.append((common.sort_photons(ph).tobytes(), conv.GetRNGState(len(steps)).tobytes()))
    assert out[0] == out[1]


def test_bunch_beyond_the_pending_entries_step_index_takes_the_classic_kernel(pooled, monkeypatch):
    """a pending entry of the pooled kernel keeps the step index in 23 bits; a converter may hold more streams than that
    (user-supplied multipliers).  Converter::pooled_for() sends such bunches to the classic kernel and the pooled launcher refuses
    them.  CLSIMHIP_POOL_INDEX_BITS lowers the guard (not the encoding), so the decision is testable with a small bunch: with 11
    bits a bunch of 1 792 steps is pooled, one of 2 048 is not, and both give the oracle's photons."""
    monkeypatch.setenv("CLSIMHIP_POOL_INDEX_BITS", "11")
    cfg = common.config("mie")
    T = common.oracle_tables(cfg)
    x, a = common.streams(2048)
    conv = common.product_converter(cfg, 2048)
    assert conv.KernelForBunch(2047) == "pool" and conv.KernelForBunch(2048) == "classic"
    xo = x
    for n, expect in ((1792, "pool"), (2048, "classic")):
        steps = common.steps_for(cfg, n, seed=41)
        assert len(steps) == n and conv.KernelForBunch(n) == expect
        ph_o, cnt_o, x_next, _ = capi.propagate(T, steps, xo[:n], a[:n], threads=8)
        xo = np.concatenate([x_next, xo[n:]])
        conv.EnqueueSteps(steps, 9)
        _, ph_p = conv.GetConversionResult()
        assert len(ph_p) == cnt_o and cnt_o > 10
        assert common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes() == common.sort_photons(ph_p).tobytes()
        assert np.array_equal(conv.GetRNGState(2048), xo)
    monkeypatch.delenv("CLSIMHIP_POOL_INDEX_BITS")
    conv = common.product_converter(cfg, 4096)
    assert conv.KernelForBunch(4096) == "pool" and conv.KernelForBunch((1 << 23) - 1) == "pool" and conv.KernelForBunch(1 << 23) == "classic"
