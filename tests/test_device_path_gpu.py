"""GPU: the device-resident entry point (clsimhip_propagate_device) used by
bench.py and by multi-GPU sharding, and size-independent properties at the
BASELINE bunch size."""
import numpy as np
import pytest
import torch

from clsim_amd.distributed import shard_range
from clsim_amd.synthetic import PHOTON_DTYPE
from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu


def device_run(conv, steps, capacity, rng_offset=0):
    dev = torch.device("cuda", 0)
    n = len(steps)
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    d_out = torch.zeros((capacity, 80), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    conv.PropagateDevice(d_steps.data_ptr(), n, d_out.data_ptr(), capacity, d_cnt.data_ptr(),
                         stream=torch.cuda.current_stream().cuda_stream, rng_offset=rng_offset)
    torch.cuda.synchronize()
    cnt = int(d_cnt.item())
    raw = d_out[:min(cnt, capacity)].cpu().numpy()
    return np.frombuffer(raw.tobytes(), dtype=PHOTON_DTYPE).copy(), cnt


def test_device_path_equals_oracle_and_host_path():
    cfg = common.config("lea")
    steps = common.steps_for(cfg, 3072, seed=8)
    n = len(steps)
    x, a = common.streams(n)
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    conv = common.product_converter(cfg, n)
    ph_d, cnt_d = device_run(conv, steps, capacity=8192)
    assert cnt_d == cnt_o
    assert common.sort_photons(ph_d).tobytes() == common.sort_photons(ph_o).tobytes()     # raw records: indices
    assert np.array_equal(conv.GetRNGState(n), x_o)
    ids = conv.ReplaceIndicesWithIDs(ph_d)
    assert common.sort_photons(ids).tobytes() == common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes()
    ms, launches = conv.KernelTimeMs(reset=True)
    assert launches == 1 and ms > 0


def test_sharding_invariance():
    """Config C4 by construction: propagating a bunch as 2, 3 or 8 contiguous shards (each with
    the matching slice of the RNG streams) yields the same photon multiset as one launch --
    steps are independent units, nothing is exchanged."""
    cfg = common.config("mie")
    steps = common.steps_for(cfg, 4096, seed=12)
    n = len(steps)
    conv = common.product_converter(cfg, n)
    whole, cnt = device_run(conv, steps, capacity=16384)
    x_whole = conv.GetRNGState(n)
    for world in (2, 3, 8):
        conv2 = common.product_converter(cfg, n)
        parts = []
        for rank in range(world):
            lo, hi = shard_range(n, rank, world)
            ph, c = device_run(conv2, steps[lo:hi], capacity=16384, rng_offset=lo)
            parts.append(ph)
        allp = np.concatenate(parts)
        assert len(allp) == cnt
        assert common.sort_photons(allp).tobytes() == common.sort_photons(whole).tobytes()
        assert np.array_equal(conv2.GetRNGState(n), x_whole)


@pytest.mark.parametrize("kernel", ["auto", "pool"])
def test_bunches_side_by_side_on_several_streams(kernel, monkeypatch):
    """clsimhip_set_concurrent_device_launches: four bunches in flight on four HIP streams (disjoint RNG stream ranges, output
    buffers of their own, each launch sized for a quarter of the chip) give what one launch over all the steps gives."""
    cfg = common.config("mie")
    k, m = 4, 8192
    steps = common.steps_for(cfg, k * m, seed=21)
    n = len(steps)
    conv = common.product_converter(cfg, n)
    whole, cnt = device_run(conv, steps, capacity=1 << 16)
    x_whole = conv.GetRNGState(n)
    dev = torch.device("cuda", 0)
    if kernel == "pool":
        monkeypatch.setenv("CLSIMHIP_KERNEL", "pool")
    for _ in range(1):
        conv2 = common.product_converter(cfg, n)
        conv2.SetConcurrentDeviceLaunches(k)
        d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
        outs = [torch.zeros((1 << 15, 80), dtype=torch.uint8, device=dev) for _ in range(k)]
        cnts = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(k)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
        torch.cuda.synchronize()
        bounds = [0, 5000, 5000 + 12288, 5000 + 12288 + 256, n]             # ragged bunches
        for j in range(k):
            lo, hi = bounds[j], bounds[j + 1]
            conv2.PropagateDevice(d_steps.data_ptr() + 48 * lo, hi - lo, outs[j].data_ptr(), 1 << 15, cnts[j].data_ptr(),
                                  stream=streams[j].cuda_stream, rng_offset=lo)
        torch.cuda.synchronize()
        parts = [np.frombuffer(outs[j][:int(cnts[j].item())].cpu().numpy().tobytes(), dtype=PHOTON_DTYPE) for j in range(k)]
        allp = np.concatenate(parts)
        assert len(allp) == cnt
        assert common.sort_photons(allp).tobytes() == common.sort_photons(whole).tobytes()
        assert np.array_equal(conv2.GetRNGState(n), x_whole)


def test_baseline_size_properties():
    """BASELINE configs[1] size (1M steps x 200 photons, SPICE-Mie, 86 strings): determinism across
    converters, every stream advanced, hit fraction and record sanity; the first 2048 steps are
    checked bit-exactly against the oracle as part of the big launch (queue order must not matter)."""
    cfg = common.config("mie")
    n = 1 << 20
    steps = common.steps_for(cfg, n, seed=1000)
    x, a = common.streams(n)
    conv = common.product_converter(cfg, n)
    ph1, cnt1 = device_run(conv, steps, capacity=1 << 21)
    x1 = conv.GetRNGState(n)
    conv_b = common.product_converter(cfg, n)
    ph2, cnt2 = device_run(conv_b, steps, capacity=1 << 21)
    assert cnt1 == cnt2 and common.sort_photons(ph1).tobytes() == common.sort_photons(ph2).tobytes()
    assert np.array_equal(x1, conv_b.GetRNGState(n))
    assert np.all(x1 != x)
    photons = float(steps["num"].sum())
    assert 2e-4 < cnt1 / photons < 3e-3
    assert np.all(ph1["stringID"] < 86) and np.all(ph1["omID"] < 60) and np.all(ph1["id"] < n)
    T = common.oracle_tables(cfg)
    m = 2048
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps[:m], x, a, threads=8)
    sub = ph1[ph1["id"] < m]
    assert len(sub) == cnt_o and common.sort_photons(sub).tobytes() == common.sort_photons(ph_o).tobytes()
    assert np.array_equal(x1[:m], x_o)


def test_converter_queue_semantics():
    """EnqueueSteps / GetConversionResult: several bunches in flight, identifiers round-trip,
    statistics add up (OpenCL.cxx:1525-1640); misuse raises the reference's messages."""
    from clsim_amd import converter as CV
    cfg = common.config("c1")
    conv = common.product_converter(cfg, 1024)
    assert conv.IsInitialized() and conv.GetMaxNumWorkitems() == 1024 and conv.GetWorkgroupSize() == conv.GetMaxWorkgroupSize() == 256
    steps = common.steps_for(cfg, 1024, seed=4, pad_to=512)
    for ident in (11, 22, 33):
        conv.EnqueueSteps(steps, ident)
    got = [conv.GetConversionResult()[0] for _ in range(3)]
    assert got == [11, 22, 33]
    st = conv.GetStatistics()
    assert st["NumKernelCalls"] == 3 and st["TotalNumPhotonsGenerated"] == 3 * float(steps["num"].sum())
    assert st["TotalDeviceTime"] > 0 and 0 < st["DeviceUtilization"] <= 1.0001
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="multiple of the workgroup size"):
        conv.EnqueueSteps(steps[:100], 1)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="greater than maximum"):
        conv.EnqueueSteps(np.concatenate([steps, steps]), 1)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="empty"):
        conv.EnqueueSteps(steps[:0], 1)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="already initialized"):
        conv.SetDOMPancakeFactor(2.0)
    # all-zero bunch: no photons, streams untouched
    x_before = conv.GetRNGState(1024)
    zero = steps.copy(); zero["num"] = 0
    conv.EnqueueSteps(zero, 9)
    ident, ph = conv.GetConversionResult()
    assert ident == 9 and len(ph) == 0 and np.array_equal(conv.GetRNGState(1024), x_before)


def test_double_buffering_pipelines_bunches_with_identical_results():
    """EnableDoubleBuffering (OpenCL.cxx:232, 296-340): two buffer sets, the next bunch's kernel runs while the
    previous bunch is downloaded and converted.  Bunches still see the RNG streams in order: every result equals the
    single-buffered converter's and the oracle's, identifiers come back in order, from a producer thread."""
    import threading
    cfg = common.config("mie")
    n = 2048
    T = common.oracle_tables(cfg)
    x, a = common.streams(n)
    bunches = [common.steps_for(cfg, n, seed=40 + b) for b in range(5)]
    bunches[3]["num"][:] = 0                       # an empty bunch in the middle of the pipeline
    expected, xo = [], x
    for st in bunches:
        ph, cnt, xo, _ = capi.propagate(T, st, xo, a, threads=8)
        expected.append(common.sort_photons(capi.replace_indices_with_ids(ph, T.geo)).tobytes())
    for double in (False, True):
        conv = common.product_converter(cfg, n, double_buffering=double)
        producer = threading.Thread(target=lambda: [conv.EnqueueSteps(st, 100 + b) for b, st in enumerate(bunches)])
        producer.start()
        for b in range(len(bunches)):
            ident, ph = conv.GetConversionResult()
            assert ident == 100 + b
            assert common.sort_photons(ph).tobytes() == expected[b], (double, b)
        producer.join()
        assert np.array_equal(conv.GetRNGState(n), xo)
        st = conv.GetStatistics()
        assert st["NumKernelCalls"] == len(bunches) and not conv.MorePhotonsAvailable()


@pytest.mark.parametrize("slices,k_new,k_search", [(1, 8, 1), (3, 8, 5), (16, 8, 64), (64, 1, 2), (7, 64, 13)])
def test_ragged_bunch_under_every_schedule(slices, k_new, k_search, monkeypatch):
    """Work-unit scheduling must not change results: a bunch whose steps hold 0 ... 1500 photons (empty steps, single
    photons, steps longer than a slice, steps shorter than the slice grid) is cut into `slices` slices per step with
    photon creation deferred until `k_new` lanes wait and the DOM search until `k_search` lanes are parked, and
    compared with the oracle, which knows none of this."""
    monkeypatch.setenv("CLSIMHIP_SLICES", str(slices))
    monkeypatch.setenv("CLSIMHIP_K_NEW", str(k_new))
    monkeypatch.setenv("CLSIMHIP_K_SEARCH", str(k_search))
    cfg = common.config("mie")
    steps = common.steps_for(cfg, 2048, seed=77)
    n = len(steps)
    rng = np.random.Generator(np.random.PCG64(5))
    num = rng.choice([0, 1, 2, 7, 63, 64, 65, 200, 399, 1500], size=n, p=[.1, .1, .1, .1, .1, .1, .1, .2, .08, .02])
    steps["num"] = num.astype(np.uint32)
    x, a = common.streams(n)
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    conv = common.product_converter(cfg, n)
    ph_d, cnt_d = device_run(conv, steps, capacity=1 << 16)
    assert cnt_d == cnt_o and cnt_o > 50
    assert common.sort_photons(ph_d).tobytes() == common.sort_photons(ph_o).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)
    # streams of empty steps are untouched
    assert np.array_equal(x_o[num == 0], np.asarray(x)[:n][num == 0])


@pytest.mark.timeout(900)
def test_full_baseline_bunch_equals_oracle():
    """BASELINE configs[1] in full: all 1 048 576 steps x 200 photons through the kernel's production schedule
    (7 workgroups per CU, 16 slices, eight sub-queues, parked searches) against the oracle run on every host core -- the complete hit
    multiset and all final RNG states, bit for bit.  (About a minute of oracle time on the GPU box's 256 threads.)"""
    import os
    cfg = common.config("mie")
    n = 1 << 20
    steps = common.steps_for(cfg, n, seed=1000)
    x, a = common.streams(n)
    conv = common.product_converter(cfg, n)
    ph_d, cnt_d = device_run(conv, steps, capacity=1 << 21)
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=os.cpu_count() or 8)
    assert cnt_d == cnt_o and cnt_o > 150000
    assert common.sort_photons(ph_d).tobytes() == common.sort_photons(ph_o).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)


@pytest.mark.parametrize("kernel", ["classic", "pool"])
def test_overflow_keeps_counting(kernel, monkeypatch):
    """More hits than the output buffer holds: the counter keeps counting, only the first `capacity` arrivals are
    stored and nothing is written behind them (propagation_kernel.c.cl:329-334), with either scheduling."""
    monkeypatch.setenv("CLSIMHIP_KERNEL", kernel)
    cfg = common.config("flasher")
    n = 2048
    steps = common.steps_for(cfg, n, seed=3)
    conv = common.product_converter(cfg, n)
    assert conv.UsesPooledKernel() == (kernel == "pool")
    dev = torch.device("cuda", 0)
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    cap = 64
    out = torch.zeros((cap + 8, 80), dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    conv.PropagateDevice(d_steps.data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cnt.item()) > cap
    assert int(out[cap:].sum().item()) == 0
    assert int((out[:cap].sum(dim=1) == 0).sum().item()) == 0


def test_rccl_gather_through_the_c_abi_world_of_one():
    """clsimhip_comm_create / clsimhip_gather_hits (RCCL called from C++, no torch.distributed): a world of one rank
    exercises the whole path on a single GPU -- unique id, communicator, ncclAllGather of the counts, the root's own
    copy -- including a counter that ran past the photon buffer's capacity."""
    from clsim_amd.distributed import HitGatherer
    cfg = common.config("flasher")
    n = 2048
    steps = common.steps_for(cfg, n, seed=3)
    conv = common.product_converter(cfg, n)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    gatherer = HitGatherer(0, 0, 1, HitGatherer.unique_id())
    for cap in (1 << 16, 100):
        out = torch.zeros((cap, 80), dtype=torch.uint8, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        gathered = torch.zeros((cap, 80), dtype=torch.uint8, device=dev)
        conv2 = common.product_converter(cfg, n)
        conv2.PropagateDevice(d_steps.data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=stream)
        counts = gatherer.gather(out.data_ptr(), cnt.data_ptr(), cap, 0, gathered.data_ptr(), cap, stream)
        torch.cuda.synchronize()
        hits = int(cnt.item())
        assert counts.tolist() == [hits] and hits > 1000
        k = min(hits, cap)
        assert torch.equal(gathered[:k], out[:k])
        if k < cap:
            assert int(gathered[k:].sum().item()) == 0
    gatherer.close()
