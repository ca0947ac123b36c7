"""The feeder (clsimhip_feeder_*, csrc/feeder.cpp): the worker thread of the reference's asynchronous light-source
converter (private/clsim/I3CLSimLightSourceToStepConverterAsync.cxx :178-392, 470-600) against a plain-Python restatement of
its loop (oracle/builders.py: feeder_model), and its queue / barrier semantics.  CPU tests feed light sources that come
with their steps (the role of a propagator); the GPU test lets the PPC front end and the GPU step producer make them."""
import threading

import numpy as np
import pytest

from clsim_amd import converter as CV
from clsim_amd import step_store as SS
from clsim_amd.converter import I3CLSimStepToPhotonConverter_exception
from clsim_amd.synthetic import STEP_DTYPE
from oracle import builders as B
from tests import common


def random_steps(rng, n, identifier):
    s = np.zeros(n, dtype=STEP_DTYPE)
    s["x"] = rng.normal(size=n)
    s["num"] = rng.choice([0, 1, 2, 7, 200, 200, 200, 37], size=n)
    s["weight"] = 1.0
    s["beta"] = 1.0
    s["id"] = identifier
    return s


def same(a, b):
    return np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(np.array(b, dtype=STEP_DTYPE)).tobytes()


def drain(feeder):
    out = []
    while True:
        r = feeder.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=20000)
        assert r is not None, "feeder timed out"
        out.append(r)
        if r[2]:
            return out


@pytest.mark.parametrize("granularity,max_bunch", [(1, 64), (64, 256), (256, 256), (32, 1024)])
def test_bunches_and_markers_follow_the_reference_loop(granularity, max_bunch):
    rng = np.random.Generator(np.random.PCG64(5))
    items = [(i, random_steps(rng, int(rng.integers(0, 900)), i)) for i in (100, 7, 55, 8, 9, 3)]
    expected = B.feeder_model(items + [None], max_bunch, granularity, SS.no_op_step()[0])
    f = SS.I3CLSimLightSourceToStepConverterAsync(maxQueueItems=3)
    f.SetMaxBunchSize(max_bunch); f.SetBunchSizeGranularity(granularity); f.Initialize()
    got = []
    consumer = threading.Thread(target=lambda: got.extend(drain(f)))      # the queues are short: producer and consumer overlap
    consumer.start()
    for identifier, steps in items:
        f.EnqueueSteps(identifier, steps)
    f.EnqueueBarrier()
    consumer.join(60)
    assert not consumer.is_alive()
    assert len(got) == len(expected)
    for (gs, gf, gl), (es, ef, el) in zip(got, expected):
        assert same(gs, es) and gf == ef and gl == el
    assert all(len(s) == max_bunch for s, _, last in got if not last) and len(got[-1][0]) % granularity == 0
    assert [i for _, fin, _ in got for i in fin] == [100, 7, 55, 8, 9, 3]      # every light source finished once, in order
    assert not f.BarrierActive() and not f.MoreStepsAvailable()


def test_barrier_semantics_and_messages():
    f = SS.I3CLSimLightSourceToStepConverterAsync()
    f.SetMaxBunchSize(128); f.SetBunchSizeGranularity(64); f.Initialize()
    assert f.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=50) is None          # nothing enqueued: timeout
    rng = np.random.Generator(np.random.PCG64(1))
    f.EnqueueSteps(1, random_steps(rng, 10, 1))
    f.EnqueueBarrier()
    assert f.BarrierActive()
    with pytest.raises(I3CLSimStepToPhotonConverter_exception, match="A barrier is already enqueued!"):
        f.EnqueueBarrier()
    with pytest.raises(I3CLSimStepToPhotonConverter_exception, match="A barrier is enqueued! You must receive all steps"):
        f.EnqueueSteps(2, random_steps(rng, 3, 2))
    steps, finished, reset = f.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=20000)
    assert reset and finished == [1] and len(steps) == 64 and int((steps["num"] > 0).sum() + 0) <= 10
    assert np.all(steps["weight"][10:] == 0) and np.all(steps["beta"][10:] == 1)                   # no-op padding (:246-254)
    assert not f.BarrierActive()
    # a barrier on an empty store: one granule of no-op steps (Async.cxx:256)
    f.EnqueueBarrier()
    steps, finished, reset = f.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=20000)
    assert reset and finished == [] and len(steps) == 64 and np.all(steps["num"] == 0)
    with pytest.raises(I3CLSimStepToPhotonConverter_exception, match="without a particle parameterisation"):
        f.EnqueueLightSource(np.zeros(1, dtype=CV.PARTICLE_DTYPE))
    with pytest.raises(I3CLSimStepToPhotonConverter_exception, match="not a multiple"):
        g = SS.I3CLSimLightSourceToStepConverterAsync(); g.SetMaxBunchSize(100); g.SetBunchSizeGranularity(64); g.Initialize()


@pytest.mark.gpu
def test_particles_through_the_front_end_and_the_gpu_step_producer():
    """particles -> PPC front end -> steps born on the GPU -> step store -> bunches: photon totals and identifiers add up, and
    the bunches are what the model makes of the same steps"""
    cfg = common.config("mie")
    ppc = CV.I3CLSimLightSourceToStepConverterPPC()
    ppc.SetWlenBias(CV.GetIceCubeDOMAcceptance()); ppc.SetMediumProperties(cfg["med_p"]); ppc.SetRandomSeed(11); ppc.Initialize()
    parts = np.zeros(5, dtype=CV.PARTICLE_DTYPE)
    parts["type"] = [CV.ParticleType.EMinus, CV.ParticleType.MuMinus, CV.ParticleType.Hadrons, CV.ParticleType.EMinus, CV.ParticleType.Gamma]
    parts["energy"] = [30.0, 100.0, 50.0, 0.002, 10.0]
    parts["length"] = [np.nan, 120.0, np.nan, np.nan, np.nan]
    parts["dz"] = -1.0
    parts["identifier"] = [11, 12, 13, 14, 15]
    f = SS.I3CLSimLightSourceToStepConverterAsync()
    f.SetMaxBunchSize(2048); f.SetBunchSizeGranularity(256); f.SetLightSourceParameterization(ppc, seed=5, device=0); f.Initialize()
    got = []
    consumer = threading.Thread(target=lambda: got.extend(drain(f)))
    consumer.start()
    for p in parts:
        f.EnqueueLightSource(p)
    f.EnqueueBarrier()
    consumer.join(120)
    assert not consumer.is_alive()
    allsteps = np.concatenate([s for s, _, _ in got])
    # (a front end of its own with the same seed: on `ppc` these identifiers would now be seen for the second time and get
    # fresh fluctuations, lightsource.h: OccurrenceCounter)
    ppc2 = CV.I3CLSimLightSourceToStepConverterPPC()
    ppc2.SetWlenBias(CV.GetIceCubeDOMAcceptance()); ppc2.SetMediumProperties(cfg["med_p"]); ppc2.SetRandomSeed(11); ppc2.Initialize()
    req = ppc2.EnqueueLightSources(parts)
    expected_photons = int((req["num_steps"] * req["photons_per_step"] + req["num_photons_in_last_step"]).sum())
    assert int(allsteps["num"].sum()) == expected_photons
    for ident in (11, 12, 13, 14, 15):
        r = req[req["identifier"] == ident]
        assert int(allsteps["num"][allsteps["id"] == ident].sum()) == int((r["num_steps"] * r["photons_per_step"] + r["num_photons_in_last_step"]).sum())
    assert [i for _, fin, _ in got for i in fin] == [11, 12, 13, 14, 15]
    assert all(len(s) == 2048 for s, _, last in got if not last) and len(got[-1][0]) % 256 == 0
    # ascending photon count inside every bunch (the store's order)
    for s, _, last in got:
        real = s["num"][s["num"] > 0] if last else s["num"]
        assert np.all(np.diff(real.astype(np.int64)) >= 0)


def test_destroying_a_feeder_with_work_in_flight_does_not_hang():
    """the worker is blocked on the full output queue and light sources wait in the input queue: destruction closes both
    queues and joins the thread (no caller is inside a call at that moment, as for any object that is being destroyed)"""
    rng = np.random.Generator(np.random.PCG64(3))
    f = SS.I3CLSimLightSourceToStepConverterAsync(maxQueueItems=2)
    f.SetMaxBunchSize(64); f.SetBunchSizeGranularity(1); f.Initialize()
    for i in range(3):
        f.EnqueueSteps(i, random_steps(rng, 1000, i))           # 15 bunches each: the output queue (2 places) is full at once
    assert f.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=20000) is not None
    finished = threading.Event()

    def destroy():
        h, f._h = f._h, None
        f._lib.clsimhip_feeder_destroy(h)
        finished.set()
    t = threading.Thread(target=destroy, daemon=True)
    t.start()
    assert finished.wait(20), "clsimhip_feeder_destroy did not return"


def test_a_dead_worker_wakes_producers_and_refuses_further_input():
    """the worker thread dies of a device error (here: a device ordinal that does not exist, so the GPU step producer
    fails on its first light source): a consumer waiting for steps gets the error, a producer blocked on the full input
    queue wakes up with it, later EnqueueLightSource / EnqueueBarrier calls are refused instead of being dropped silently"""
    cfg = common.config("mie")
    ppc = CV.I3CLSimLightSourceToStepConverterPPC()
    ppc.SetWlenBias(CV.GetIceCubeDOMAcceptance()); ppc.SetMediumProperties(cfg["med_p"]); ppc.SetRandomSeed(3); ppc.Initialize()
    f = SS.I3CLSimLightSourceToStepConverterAsync(maxQueueItems=1)
    f.SetMaxBunchSize(512); f.SetBunchSizeGranularity(64); f.SetLightSourceParameterization(ppc, seed=3, device=4242); f.Initialize()
    parts = np.zeros(6, dtype=CV.PARTICLE_DTYPE)
    parts["type"], parts["energy"], parts["dz"], parts["length"] = CV.ParticleType.EMinus, 5.0, -1.0, np.nan
    parts["identifier"] = np.arange(6)
    outcome = []

    def produce():
        try:
            for p in parts:
                f.EnqueueLightSource(p)
            outcome.append("accepted everything")
        except I3CLSimStepToPhotonConverter_exception as e:
            outcome.append(str(e))
    t = threading.Thread(target=produce, daemon=True)
    t.start()
    t.join(30)
    assert not t.is_alive(), "a producer hangs on the input queue of a feeder whose worker is dead"
    assert outcome and "feeder thread" in outcome[0], outcome
    with pytest.raises(I3CLSimStepToPhotonConverter_exception, match="feeder thread"):
        f.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=5000)
    with pytest.raises(I3CLSimStepToPhotonConverter_exception, match="feeder thread"):
        f.EnqueueLightSource(parts[0])
    with pytest.raises(I3CLSimStepToPhotonConverter_exception, match="feeder thread"):
        f.EnqueueBarrier()
    assert not f.BarrierActive()
