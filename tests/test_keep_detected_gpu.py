"""SetStopDetectedPhotons(false): the kernel without STOP_PHOTONS_ON_DETECTION (OpenCL.cxx:395-397;
sparse_collision_kernel.c.cl:85-104, :165-186, :245-253; propagation_kernel.c.cl:704-750).  Every DOM a segment enters
records the photon, the step is not shortened and the photon travels on to its absorption.  The HIP instantiations
(classic scheduling: prop_keep_kernel.hip; pooled scheduling, round 4: prop_pool_keep_kernel.hip; both through
find_collisions_keep) against the oracle's restatement of those branches, which is itself pinned on the
reference's kernel text (tests/test_verbatim_cl.py: the `*_keep` cases)."""
import numpy as np
import pytest

from clsim_amd import converter as CV
from clsim_amd import synthetic as S
from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu


def both(name, n_steps, seed=3):
    cfg = common.config(name)
    steps = common.steps_for(cfg, n_steps, seed=seed)
    x, a = common.streams(len(steps))
    T = common.oracle_tables(cfg, stop_detected=False)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    conv = common.product_converter(cfg, len(steps), stop_detected=False)
    return cfg, steps, T, (ph_o, cnt_o, x_o), conv


@pytest.fixture(params=["classic", "pool"])
def kernel(request, monkeypatch):
    monkeypatch.setenv("CLSIMHIP_KERNEL", request.param)
    for k in ("CLSIMHIP_POOL_R", "CLSIMHIP_K_POP", "CLSIMHIP_K_NEW", "CLSIMHIP_SLICES", "CLSIMHIP_K_SEARCH", "CLSIMHIP_GRID"):
        monkeypatch.delenv(k, raising=False)
    return request.param


@pytest.mark.parametrize("name,n_steps", [("c1", 2048), ("mie", 8192), ("lea", 8192), ("flasher", 2048), ("clear", 4096),
                                          ("photonics_mie", 4096), ("mie_regular", 4096), ("clear_60", 4096)])
def test_every_dom_on_the_way_records_the_photon(kernel, name, n_steps):
    cfg, steps, T, (ph_o, cnt_o, x_o), conv = both(name, n_steps)
    assert cnt_o > 100
    conv.EnqueueSteps(steps, 11)
    ident, ph_p = conv.GetConversionResult()
    assert ident == 11 and len(ph_p) == cnt_o
    expect = capi.replace_indices_with_ids(ph_o.copy(), T.geo)
    assert common.sort_photons(ph_p).tobytes() == common.sort_photons(expect).tobytes()
    assert np.array_equal(conv.GetRNGState(len(steps)), x_o)
    assert conv.KernelForBunch(len(steps)) == kernel


def test_photons_travel_on_after_a_detection():
    """the same bunch with and without STOP_PHOTONS_ON_DETECTION: photons with several records, and different streams afterwards
    (a detected photon goes on drawing random numbers).  NOT more records in total: in this ice a segment's cell rectangle
    holds a dozen strings, and the reference's string mask (`1 << n%64` on an int: strings n and n+32 share a bit) skips some."""
    cfg = common.config("clear")
    steps = common.steps_for(cfg, 4096, seed=5)
    stop = common.product_converter(cfg, len(steps))
    keep = common.product_converter(cfg, len(steps), stop_detected=False)
    stop.EnqueueSteps(steps, 1); keep.EnqueueSteps(steps, 1)
    _, ph_s = stop.GetConversionResult()
    _, ph_k = keep.GetConversionResult()
    assert len(ph_k) > 100 and len(ph_s) > 100
    # one photon = one creation point: start position + time + direction
    key = np.ascontiguousarray(ph_k).view(np.uint8).reshape(len(ph_k), 80)[:, 48:72]       # sx, sy, sz, st, stheta, sphi
    _, counts = np.unique(key, axis=0, return_counts=True)
    assert counts.max() >= 2 and (counts > 1).sum() >= 5
    assert not np.array_equal(stop.GetRNGState(len(steps)), keep.GetRNGState(len(steps)))


def test_the_counter_runs_past_a_full_buffer(kernel):
    """c.cl:329-334: the counter counts every hit, the first `capacity` arrivals are stored -- each one of the oracle's records"""
    cfg, steps, T, (ph_o, cnt_o, x_o), conv = both("clear", 4096)
    import torch
    dev = torch.device("cuda", 0)
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(len(steps), 48).copy()).to(dev)
    capacity = 500
    assert cnt_o > 2 * capacity
    d_out = torch.zeros((capacity, 80), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    conv.PropagateDevice(d_steps.data_ptr(), len(steps), d_out.data_ptr(), capacity, d_cnt.data_ptr())
    torch.cuda.synchronize()
    assert int(d_cnt.item()) == cnt_o
    raw_o = ph_o.tobytes()                      # the device path keeps the kernel's indices
    have = set(raw_o[i:i + 80] for i in range(0, len(raw_o), 80))
    raw = d_out.cpu().numpy().tobytes()
    stored = [raw[i:i + 80] for i in range(0, len(raw), 80)]
    assert len(set(stored)) == capacity and all(r in have for r in stored)
    assert np.array_equal(conv.GetRNGState(len(steps)), x_o)


def test_streams_persist_across_bunches_without_stop(kernel):
    cfg = common.config("mie")
    steps = common.steps_for(cfg, 1024, seed=9)
    x, a = common.streams(1024)
    T = common.oracle_tables(cfg, stop_detected=False)
    conv = common.product_converter(cfg, 1024, stop_detected=False)
    xo = x
    for bunch in range(3):
        ph_o, cnt_o, xo, _ = capi.propagate(T, steps, xo, a, threads=8)
        conv.EnqueueSteps(steps, bunch)
        ident, ph_p = conv.GetConversionResult()
        assert ident == bunch
        assert common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes() == common.sort_photons(ph_p).tobytes()
    assert np.array_equal(conv.GetRNGState(1024), xo)


@pytest.mark.parametrize("ring,k_pop,k_new,slices,k_search", [(4, 1, 1, 1, 1), (8, 64, 8, 3, 5), (30, 4, 29, 16, 5), (17, 2, 64, 7, 64), (64, 16, 1, 64, 2)])
def test_pooled_keep_ragged_bunch_under_every_schedule(monkeypatch, ring, k_pop, k_new, slices, k_search):
    """the pooled kernel without STOP_PHOTONS_ON_DETECTION: ring sizes and thresholds must not change results -- steps of 0 ... 1500
    photons in ice where a photon passes dozens of DOMs (`clear`: the string masks at work), two bunches in a row"""
    for k, v in (("CLSIMHIP_KERNEL", "pool"), ("CLSIMHIP_POOL_R", ring), ("CLSIMHIP_K_POP", k_pop), ("CLSIMHIP_K_NEW", k_new),
                 ("CLSIMHIP_SLICES", slices), ("CLSIMHIP_K_SEARCH", k_search)):
        monkeypatch.setenv(k, str(v))
    cfg = common.config("clear")
    n = 2048
    steps = common.steps_for(cfg, n, seed=17)
    rng = np.random.default_rng(5)
    num = rng.integers(0, 400, n)
    num[rng.random(n) < 0.15] = 0
    num[rng.random(n) < 0.05] = 1
    num[rng.random(n) < 0.02] = 1500
    steps["num"] = num
    x, a = common.streams(n)
    T = common.oracle_tables(cfg, stop_detected=False)
    conv = common.product_converter(cfg, n, stop_detected=False)
    assert conv.UsesPooledKernel()
    xo = x
    for bunch in range(2):
        ph_o, cnt_o, xo, _ = capi.propagate(T, steps, xo, a, threads=8)
        conv.EnqueueSteps(steps, bunch)
        _, ph_p = conv.GetConversionResult()
        assert len(ph_p) == cnt_o and cnt_o > 50
        assert common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes() == common.sort_photons(ph_p).tobytes()
        assert np.array_equal(conv.GetRNGState(n), xo)


@pytest.mark.parametrize("name,n_steps", [("mie", 131072), ("lea", 65536), ("flasher", 32768), ("clear", 32768)])
def test_pooled_keep_equals_classic_keep_on_large_bunches(monkeypatch, name, n_steps):
    """enough steps to fill every wave's pool on the whole chip: both schedulings, same multiset and stream states"""
    cfg = common.config(name)
    steps = common.steps_for(cfg, n_steps, seed=23)
    n = len(steps)
    results = []
    for kern in ("classic", "pool"):
        monkeypatch.setenv("CLSIMHIP_KERNEL", kern)
        conv = common.product_converter(cfg, n, stop_detected=False)
        assert conv.UsesPooledKernel() == (kern == "pool")
        conv.EnqueueSteps(steps, 5)
        _, ph = conv.GetConversionResult()
        results.append((common.sort_photons(ph).tobytes(), conv.GetRNGState(n)))
        del conv
    assert len(results[0][0]) > 80 * 50
    assert results[0][0] == results[1][0]
    assert np.array_equal(results[0][1], results[1][1])


def test_default_kernel_choice_without_stop_follows_the_bunch_size():
    """no CLSIMHIP_KERNEL: large bunches take the pooled kernel in this mode too (round 4), small ones the classic kernel"""
    import os
    assert "CLSIMHIP_KERNEL" not in os.environ
    cfg = common.config("mie")
    conv = common.product_converter(cfg, 1 << 20, stop_detected=False)
    assert conv.KernelForBunch(1 << 20) == "pool" and conv.KernelForBunch(4096) == "classic"
