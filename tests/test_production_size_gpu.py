"""GPU: BASELINE configs[2] (C3) and configs[4] (C5, flasher half) at their production bunch sizes -- the launch
geometry bench.py measures (10 000 000 cascade steps x 200 photons on SPICE-Lea in two bunches; 2 621 440 flasher steps x 400 photons
from a point source at a DOM, 48 Mi-record photon buffer).  Pattern of test_baseline_size_properties: the whole bunch
runs once on each of two converters (determinism: the multiset of all 80-byte records through an order-independent
64-bit checksum computed on the device, and every final stream state), and the records of the first 2048 steps are
pulled out of the big launch and compared bit for bit with the oracle run on those steps alone.

Round 5: a prefix never sees the end of the work queues, sub-queue exhaustion, unit retirement or the padded
tail.  Steps are independent units with their own streams, so the oracle run on ANY subset of a bunch's steps, each with its own
x[i], a[i], is exact: besides the prefix, 65 536 step indices drawn uniformly over the WHOLE bunch (the last 512 real steps and the
first padding step always among them) are run through the oracle and compared, bit for bit, with the records of those identifiers
inside the big launch and with those streams' final states -- both C3 bunches (the second from the first's final states at those
indices), C5, and C2 without STOP_PHOTONS_ON_DETECTION on the pooled kernel."""
import os

import numpy as np
import pytest
import torch

from clsim_amd import synthetic as S
from clsim_amd.synthetic import PHOTON_DTYPE
from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu


def big_run(conv, d_steps, n, capacity):
    dev = d_steps.device
    d_out = torch.empty((capacity, 80), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    conv.PropagateDevice(d_steps.data_ptr(), n, d_out.data_ptr(), capacity, d_cnt.data_ptr(),
                         stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    cnt = int(d_cnt.item())
    assert cnt <= capacity, "photon buffer overflow: %d > %d" % (cnt, capacity)
    return d_out[:cnt], cnt


def multiset_checksum(records):
    """Order-independent checksum of n x 80-byte records: per record a 64-bit mix of its 10 quadwords, summed mod 2^64."""
    q = records.view(torch.int64)                                    # n x 10
    mult = torch.tensor([0x9E3779B97F4A7C15 - (1 << 64), 0x3C6EF372FE94F82B, 0x5BD1E9955BD1E995, 0x2545F4914F6CDD1D,
                         0x1B873593CC9E2D51, 0x27D4EB2F165667C5, 0x165667B19E3779F9, 0x7FB5D329728EA185,
                         0x0AEF17502108EF2F, 0x62A9D9ED799705F5], dtype=torch.int64, device=records.device)
    h = (q * mult).sum(dim=1)
    h = (h ^ (h >> 29)) * 0x3C79AC492BA7B653
    h = h ^ (h >> 32)
    return int(h.sum().item()), int((h * h).sum().item())


def prefix_records(records, m):
    ids = records.view(torch.int32)[:, 10]                            # I3CLSimPhoton.identifier
    sub = records[(ids >= 0) & (ids < m)].cpu().numpy()
    return np.frombuffer(sub.tobytes(), dtype=PHOTON_DTYPE).copy()


SUBSET = 65536


def whole_bunch_subset(steps, rng, size=SUBSET):
    """sorted step indices spread over the whole bunch; always the last 512 real steps and, if the bunch is padded, the first
    padding step and the very last step"""
    n = len(steps)
    real = int((steps["num"] > 0).sum()) if (steps["num"][-1] == 0) else n
    forced = list(range(max(0, real - 512), real)) + ([real, n - 1] if real < n else [n - 1]) + [0]
    drawn = rng.choice(n, size=min(size, n), replace=False)
    return np.unique(np.concatenate([np.asarray(forced, dtype=np.int64), drawn.astype(np.int64)]))


def subset_records(records, step_ids, n):
    """records of the big launch (device tensor, n_hits x 80 bytes) whose identifier is in step_ids"""
    wanted = torch.zeros(n, dtype=torch.bool, device=records.device)
    wanted[torch.from_numpy(step_ids).to(records.device)] = True
    ids = records.view(torch.int32)[:, 10].long()
    sub = records[wanted[ids.clamp(0, n - 1)] & (ids >= 0) & (ids < n)].cpu().numpy()
    return np.frombuffer(sub.tobytes(), dtype=PHOTON_DTYPE).copy()


def check_at_size(cfg, bunches, capacity, hit_fraction_range, m=2048, stop_detected=True, expect_kernel="pool"):
    """bunches: step arrays of one size, run back to back on ONE converter (the RNG streams persist from bunch to bunch,
    propagation_kernel.c.cl:458-461, 911-912) -- and once more on a second converter."""
    if isinstance(bunches, np.ndarray):
        bunches = [bunches]
    n = len(bunches[0])
    assert all(len(b) == n for b in bunches)
    dev = torch.device("cuda", 0)
    x, a = common.streams(n)
    conv = common.product_converter(cfg, n, stop_detected=stop_detected)
    conv_b = common.product_converter(cfg, n, stop_detected=stop_detected)
    assert conv.KernelForBunch(n) == expect_kernel
    T = common.oracle_tables(cfg, stop_detected=stop_detected)
    x_before, x_oracle, total = x, x[:m], 0
    rng = np.random.default_rng(20260504)
    threads = min(os.cpu_count() or 8, 256)
    for steps in bunches:
        real = steps["num"] > 0                                          # a record's identifier IS its step's index (padding steps: no records)
        assert np.array_equal(steps["id"][real], np.arange(n, dtype=steps["id"].dtype)[real])
        d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
        rec1, cnt1 = big_run(conv, d_steps, n, capacity)
        x1 = conv.GetRNGState(n)
        sum1 = multiset_checksum(rec1)
        sub = prefix_records(rec1, m)
        pick = whole_bunch_subset(steps, rng)
        sub_pick = subset_records(rec1, pick, n)
        string_ok = bool((rec1.view(torch.int16)[:, 22] < 86).all()) and bool((rec1.view(torch.int16)[:, 22] >= 0).all())
        dom_ok = bool((rec1.view(torch.int16)[:, 23] < 60).all())
        del rec1
        torch.cuda.empty_cache()
        # determinism across converters (and across queue schedules: the order lanes take units in is not reproducible)
        rec2, cnt2 = big_run(conv_b, d_steps, n, capacity)
        assert cnt1 == cnt2
        assert multiset_checksum(rec2) == sum1
        assert np.array_equal(x1, conv_b.GetRNGState(n))
        del rec2, d_steps
        torch.cuda.empty_cache()
        live = steps["num"] > 0
        assert np.all(x1[live] != x_before[live])                    # every stream with photons advanced
        assert np.array_equal(x1[~live], x_before[~live])            # a padding step leaves its stream alone
        photons = float(steps["num"].sum())
        assert hit_fraction_range[0] < cnt1 / photons < hit_fraction_range[1], cnt1 / photons
        assert string_ok and dom_ok
        # the first m steps inside the big launch == the oracle on those steps alone, continued from the previous bunch's streams
        ph_o, cnt_o, x_oracle, _ = capi.propagate(T, steps[:m], x_oracle, a[:m], threads=16)
        assert len(sub) == cnt_o
        assert common.sort_photons(sub).tobytes() == common.sort_photons(ph_o).tobytes()
        assert np.array_equal(x1[:m], x_oracle)
        # the same over the whole bunch: the drawn steps, each with ITS stream as the previous bunch left it
        ph_s, cnt_s, x_s, _ = capi.propagate(T, steps[pick], x_before[pick], a[pick], threads=threads)
        assert len(pick) >= min(SUBSET, n) and pick[-1] == n - 1 and cnt_s > 0.5 * len(pick) / n * cnt1
        assert len(sub_pick) == cnt_s, (len(sub_pick), cnt_s)
        assert common.sort_photons(sub_pick).tobytes() == common.sort_photons(ph_s).tobytes()
        assert np.array_equal(x1[pick], x_s)
        x_before = x1
        total += cnt1
    return total


@pytest.mark.timeout(900)
def test_c3_spice_lea_production_bunches():
    """bench.py --workload c3 = BASELINE configs[2] as written: 10 000 000 cascade steps on SPICE-Lea (tilt + anisotropy + direction
    transforms) as TWO bunches of 5 000 192 (a converter holds at most 6 139 850 streams, OpenCL.cxx:250; the second bunch carries 384
    padding steps), back to back on one converter: 2.0e9 photons, about 11 steps per resident lane and bunch.  The second bunch's
    prefix is compared with the oracle continued from the first bunch's final streams."""
    cfg = common.config("lea")
    shard, n = 10000000, 5000192
    bunches = [S.cascade_steps(min(n, shard - b * n), seed=1000 + 7919 * b, photons_per_step=200, pad_to=n) for b in range(2)]
    assert int((bunches[1]["num"] == 0).sum()) == 2 * n - shard
    check_at_size(cfg, bunches, capacity=8 << 20, hit_fraction_range=(2e-4, 3e-3))


@pytest.mark.timeout(900)
def test_c5_flasher_production_bunch():
    """bench.py --workload c5: 2 621 440 flasher steps x 400 photons (1.05e9 photons), 405 nm, point source at a DOM near
    the detector centre; ~1.6 % of the photons are detected (1.3 GB of records in a 48 Mi-record buffer)."""
    cfg = common.config("flasher")
    g = cfg["geom"]
    k = int(np.argmin(np.abs(g["x"]) + np.abs(g["y"]) + np.abs(g["z"] + 100.0)))
    n = 2621440
    steps = S.flasher_steps(n, seed=1000, photons_per_step=400, position=(float(g["x"][k]), float(g["y"][k]), float(g["z"][k])))
    check_at_size(cfg, steps, capacity=48 << 20, hit_fraction_range=(5e-3, 5e-2))


@pytest.mark.timeout(900)
def test_c2_without_stop_on_detection_pooled_whole_bunch_subset():
    """BASELINE configs[1] with the reference class's default detection mode (SetStopDetectedPhotons(false), OpenCL.cxx:86) on the
    pooled kernel (prop_pool_keep_kernel.hip): 1 048 576 steps x 200 photons, a 65 536-step subset of the whole bunch and the
    prefix against the oracle's restatement of that mode."""
    cfg = common.config("mie")
    n = 1 << 20
    steps = S.cascade_steps(n, seed=1000, photons_per_step=200)
    check_at_size(cfg, steps, capacity=4 << 20, hit_fraction_range=(2e-4, 3e-3), stop_detected=False)


@pytest.mark.timeout(900)
def test_c4_one_gpu_shard_three_bunches():
    """BASELINE configs[3] = 100M steps over 8 GPUs: ONE GPU's shard as bench.py --gpus 8 runs it (rank 0) -- 12 500 000 cascade steps
    on SPICE-Mie in 3 bunches of 4 166 912 on one converter (the last one padded), streams carried from bunch to bunch.  No second GPU
    is needed to check a shard: steps are independent units and a rank's streams are its own.  Prefix and whole-bunch subset of every
    bunch against the oracle, determinism on a second converter."""
    cfg = common.config("mie")
    shard, n = 12500000, 4166912
    bunches = [S.cascade_steps(min(n, shard - b * n), seed=1000 + 7919 * b, photons_per_step=200, pad_to=n) for b in range(3)]
    assert sum(int((b["num"] > 0).sum()) for b in bunches) == shard and int((bunches[2]["num"] == 0).sum()) == 3 * n - shard
    check_at_size(cfg, bunches, capacity=4 << 20, hit_fraction_range=(2e-4, 3e-3))


@pytest.mark.timeout(600)
def test_c5_eight_gpu_split_shard_on_the_classic_kernel():
    """BASELINE configs[4] split over 8 GPUs: 10^9 photons / 8 / 400 = 312 500 flasher steps per GPU, padded to 312 832 -- fewer than the
    pooled kernel's threshold, so the classic kernel runs them (bench.py: stream_count_note).  The whole shard's subset and prefix
    against the oracle."""
    cfg = common.config("flasher")
    g = cfg["geom"]
    k = int(np.argmin(np.abs(g["x"]) + np.abs(g["y"]) + np.abs(g["z"] + 100.0)))
    steps = S.flasher_steps(312500, seed=1000, photons_per_step=400, position=(float(g["x"][k]), float(g["y"][k]), float(g["z"][k])), pad_to=512)
    assert len(steps) == 312832
    check_at_size(cfg, steps, capacity=8 << 20, hit_fraction_range=(5e-3, 5e-2), expect_kernel="classic")


TAB_SUBSET = 65536


@pytest.mark.timeout(1800)
def test_c5_table_maker_one_gpu_share_of_a_billion_photons():
    """BASELINE configs[4], table-maker half, at the size ONE of 8 GPUs sees: 10^9 / 8 = 1.25e8 photons = 625 000 cascade-like
    steps x 200 photons at the origin (padded to 625 152 with empty steps) through I3CLSimStepToTableConverterHIP, the default
    spherical axes of python/tablemaker/tabulator.py:621-641 (200 x 36 x 100 x 105 bins + under/overflow = 670 MB of binary64
    sums), SPICE-Mie, 42 absorption lengths, one sample per metre -- what bench.py --workload tab runs, 2.4 times its bunch.
    (reference: propagation_kernel.c.cl:755-785, I3CLSimStepToTableConverter.cxx:178-265.)

    (i)   two converters fill equal tables: same occupied bins, every sum equal to 1e-12 (fp64 atomics: the order differs);
    (ii)  every stream of a 2 048-step prefix AND of 65 536 step indices drawn over the WHOLE bunch (the last 512 real steps,
          the first padding step and the last step always among them) ends bit-equal to the oracle's run of those steps alone,
          each from its own (x, a) -- a stream's draws depend on its own step only, and with a fixed number of absorption lengths
          every draw of a photon lies before a branch the table's arithmetic decides (leaving the table ends the photon);
    (iii) the drawn steps' OWN table -- a third converter that is given those steps and their streams alone -- equals the
          oracle's entries of those steps added up in binary64: same occupied bins, sums to 1e-12, same sum of weights;
    (iv)  nothing is left over: the reference's kernel hands a stream back with `numPhotons` left when its entry buffer is full
          (c.cl:770-776); this table maker has no entry buffers, so every photon of the bunch must be in the statistics, every
          stream with photons must have advanced, every padding stream must be where it was -- and the oracle, run with the
          reference's resume loop, reports no unfinished call either (capi.tabulate_accumulate raises if one occurs)."""
    import math
    import time
    from clsim_amd import converter as CV
    from clsim_amd import tabulator as TB
    from oracle import builders as B
    cfg = common.config("mie")
    share = 10 ** 9 // 8 // 200
    steps = S.cascade_steps(share, seed=1000, vertex=(0.0, 0.0, 0.0), photons_per_step=200, pad_to=256)
    n = len(steps)
    assert share == 625000 and n == 625152 and int(steps["num"].sum()) == 125000000
    x, a = common.streams(n)
    ang = [0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435]
    area = math.pi * 0.16510 ** 2
    ref7 = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0)

    def axes():
        return TB.SphericalAxes([TB.PowerAxis(0, 580, 200, 2), TB.LinearAxis(0, 180, 36), TB.LinearAxis(-1, 1, 100), TB.PowerAxis(0, 7e3, 105, 2)])

    def fill(streams, bunch):
        tab = TB.I3CLSimStepToTableConverterHIP(0, axes(), False, cfg["med_p"], area, CV.GetIceCubeDOMAcceptance(),
                                                TB.I3CLSimFunctionPolynomial(ang), streams)
        tab.EnqueueSteps(bunch, ref7)
        tab.Finish()
        st = tab.GetStatistics()
        return tab.GetBinSums(), tab.GetRNGState(len(bunch)), st

    sums_a, x_a, st_a = fill((x, a), steps)
    assert st_a["NumKernelCalls"] == 1 and st_a["NumPhotons"] == 125000000.0                                        # (iv)
    assert abs(st_a["SumOfPhotonWeights"] - float((steps["num"] * steps["weight"].astype(np.float64)).sum())) < 1e-3
    live = steps["num"] > 0
    assert np.all(x_a[live] != x[live]) and np.array_equal(x_a[~live], x[~live])                                    # (iv)
    total_a, occupied_a = float(sums_a.sum()), int((sums_a > 0).sum())
    assert occupied_a > 0.6 * sums_a.size and np.all(sums_a >= 0)
    # a photon leaves about 1 660 samples behind whose weights fall with the absorption lengths travelled: about 30 per photon (bench.py --workload tab)
    assert 10.0 < total_a / 125000000.0 < 100.0
    sums_b, x_b, st_b = fill((x, a), steps)                                                                         # (i)
    assert np.array_equal(x_a, x_b) and st_b["NumPhotons"] == st_a["NumPhotons"]
    assert np.array_equal(sums_a > 0, sums_b > 0)
    assert np.allclose(sums_a, sums_b, rtol=1e-12, atol=0)
    assert abs(float(sums_b.sum()) / total_a - 1) < 1e-12
    del sums_b
    # ---- the oracle on the prefix and on the whole-bunch subset ----
    o_axes = [B.power_axis(0, 580, 200, 2), B.linear_axis(0, 180, 36), B.linear_axis(-1, 1, 100), B.power_axis(0, 7e3, 105, 2)]
    tb = B.tabulator_config("spherical", o_axes, cfg["med_o"], ang, entries_per_stream=65536)          # 16 photons of <= 2 200 samples a call
    bias_o = B.icecube_dom_acceptance()
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    T = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias_o, cfg["med_o"])], bias_o, pancake=1.0, tabulator=tb)
    ref_o = B.reference_particle(ref7[:3], ref7[3], ref7[4:])
    threads = min(os.cpu_count() or 8, 256)
    m = 2048
    t0 = time.time()
    _, cnt_p, x_p = capi.tabulate_accumulate(T, steps[:m], x[:m], a[:m], ref_o, bins=None, photons_per_call=16, threads=threads)
    per_step = (time.time() - t0) / m
    assert np.array_equal(x_a[:m], x_p)                                                                             # (ii) prefix
    assert 1000.0 < cnt_p.sum() / float(steps["num"][:m].sum()) < 2200.0
    # the subset: 65 536 steps if the host finishes them in about four minutes at the prefix's pace, never fewer than 8 192
    size = int(min(TAB_SUBSET, max(8192, 240.0 / per_step)))
    rng = np.random.default_rng(20261005)
    pick = whole_bunch_subset(steps, rng, size=size)
    # (a converter wants a multiple of 256 streams: the surplus is taken from the drawn indices in front of the forced ones)
    pick = np.concatenate([pick[:1], pick[1 + len(pick) % 256:]])
    assert len(pick) % 256 == 0 and len(pick) >= 8192 and pick[-1] == n - 1 and pick[0] == 0
    assert np.isin(np.arange(share - 512, share + 1), pick).all()
    bins_o = np.zeros(tb["n_bins"], dtype=np.float64)
    sum_s, cnt_s, x_s = capi.tabulate_accumulate(T, steps[pick], x[pick], a[pick], ref_o, bins=bins_o, photons_per_call=16, threads=threads)
    assert np.array_equal(x_a[pick], x_s)                                                                           # (ii) whole-bunch subset
    assert np.all((cnt_s > 0) == (steps["num"][pick] > 0))
    # (iii) the subset alone on the GPU, on streams of its own
    sums_s, x_g, st_s = fill((x[pick].copy(), a[pick].copy()), steps[pick])
    assert np.array_equal(x_g, x_s) and st_s["NumPhotons"] == float(steps["num"][pick].sum())
    assert np.array_equal(sums_s > 0, bins_o > 0) and int((bins_o > 0).sum()) > 1000000
    assert np.allclose(sums_s, bins_o, rtol=1e-12, atol=0)
    assert abs(float(sums_s.sum()) / float(sum_s.sum()) - 1) < 1e-12
    # and the subset is a part of the whole: bin by bin its sums do not exceed the big table's
    assert np.all(sums_s <= sums_a * (1 + 1e-12))
    print("table maker at size: %d photons, %d occupied bins of %d, sum of weights %.6e; oracle subset %d steps, %.0f samples/photon"
          % (int(st_a["NumPhotons"]), occupied_a, sums_a.size, total_a, len(pick), cnt_s.sum() / float(steps["num"][pick].sum())))
