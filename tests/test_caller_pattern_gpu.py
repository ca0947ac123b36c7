"""The reference's caller pattern on one converter (private/clsim/I3CLSimServer.cxx:126-135, 324-331): the server
starts FIVE threads per converter, each looping `EnqueueSteps(batch, id)` then `GetConversionResult()` -- "not necessarily from the
batch we just enqueued" -- and routes the result by its identifier.  OpenCL.cxx:1525-1619: a bounded input queue, a worker that
launches bunches in queue order, results handed over one at a time.

Here: five threads, 40 bunches of 4 distinct step sets, double buffering off and on.  The RNG streams chain in LAUNCH order
(propagation_kernel.c.cl:458-461, 911-912), which is the order the bunches entered the input queue: the test records that order
under a lock that covers the EnqueueSteps call only (the queue's own mutex serialises it anyway; GetConversionResult runs unlocked,
as in the server), replays it through the oracle and demands every returned identifier's photons to be the oracle's for THAT bunch
at THAT position of the chain, the final stream states to be the chain's end, `QueueSize() == 0` and `!MorePhotonsAvailable()` at the
end, and no thread still waiting after 120 s."""
import threading
import time

import numpy as np
import pytest

from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu

THREADS, BUNCHES, SETS = 5, 40, 4


@pytest.mark.timeout(600)
@pytest.mark.parametrize("double_buffering", [False, True])
def test_five_threads_enqueue_then_get_on_one_converter(double_buffering):
    cfg = common.config("mie")
    n = 2048
    T = common.oracle_tables(cfg)
    x, a = common.streams(n)
    step_sets = [common.steps_for(cfg, n, seed=300 + k) for k in range(SETS)]
    step_sets[2]["num"][::3] = 0                    # a ragged set
    step_sets[3]["num"][:] = rng_counts = np.random.default_rng(9).integers(0, 300, n).astype(np.uint32)
    assert rng_counts.sum() > 0
    conv = common.product_converter(cfg, n, double_buffering=double_buffering)
    assert conv.QueueSize() == 0 and not conv.MorePhotonsAvailable()

    order, order_lock = [], threading.Lock()        # identifiers in the order they entered the input queue
    results, results_lock = {}, threading.Lock()    # identifier -> (sorted record bytes, taken by thread)
    next_bunch = iter(range(BUNCHES))
    next_lock = threading.Lock()
    failures = []

    def caller(t):
        try:
            while True:
                with next_lock:
                    k = next(next_bunch, None)
                if k is None:
                    return
                ident = 1000 + k
                with order_lock:                    # (covers the enqueue only: the order of `order` IS the queue's order)
                    conv.EnqueueSteps(step_sets[k % SETS], ident)
                    order.append(ident)
                got, ph = conv.GetConversionResult()        # whatever bunch is finished next, not necessarily ours
                with results_lock:
                    assert got not in results, "identifier %d returned twice" % got
                    results[got] = (common.sort_photons(ph).tobytes(), t)
        except BaseException as e:                  # noqa: BLE001 -- reported by the main thread
            failures.append((t, repr(e)))

    threads = [threading.Thread(target=caller, args=(t,), daemon=True) for t in range(THREADS)]
    t0 = time.time()
    for th in threads:
        th.start()
    for th in threads:
        th.join(max(1.0, 120.0 - (time.time() - t0)))
    assert not any(th.is_alive() for th in threads), "a caller thread is still waiting after 120 s (deadlock)"
    assert not failures, failures
    assert sorted(order) == sorted(results) == [1000 + k for k in range(BUNCHES)]
    assert conv.QueueSize() == 0 and not conv.MorePhotonsAvailable()
    st = conv.GetStatistics()
    assert st["NumKernelCalls"] == BUNCHES
    # results do come back to threads that did not enqueue them (otherwise this test would not exercise the pattern);
    # with five threads and forty bunches at least one hand-over crosses threads in practice -- not asserted, it is scheduling

    # replay the chain in launch order through the oracle
    xo = x
    for position, ident in enumerate(order):
        steps = step_sets[(ident - 1000) % SETS]
        ph_o, cnt_o, xo, _ = capi.propagate(T, steps, xo, a, threads=16)
        want = common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes()
        assert results[ident][0] == want, "bunch %d (position %d in the launch order, taken by thread %d) differs from the oracle" % (
            ident, position, results[ident][1])
    assert np.array_equal(conv.GetRNGState(n), xo)
