"""Step store and bunching rule: the product's C++ (through the C ABI) against the restatement of
public/clsim/I3CLSimStepStore.h and I3CLSimLightSourceToStepConverterAsync.cxx:209-273 in oracle/builders.py."""
import numpy as np
import pytest

from clsim_amd import step_store as SS
from clsim_amd.converter import I3CLSimStepToPhotonConverter_exception
from clsim_amd.synthetic import STEP_DTYPE
from oracle import builders as B


def random_steps(rng, n, identifier, max_photons=40):
    s = np.zeros(n, dtype=STEP_DTYPE)
    s["x"], s["y"], s["z"] = rng.normal(size=(3, n)).astype(np.float32)
    s["num"] = rng.integers(0, max_photons, n)
    s["weight"] = 1.0
    s["beta"] = 1.0
    s["id"] = identifier
    return s


def same(a, b):
    a = np.array(a, dtype=STEP_DTYPE) if not isinstance(a, np.ndarray) else a
    b = np.array([np.asarray(v).reshape(()) for v in b], dtype=STEP_DTYPE) if not isinstance(b, np.ndarray) else b
    return len(a) == len(b) and a.tobytes() == b.tobytes()


def test_pop_order_counts_and_fill():
    rng = np.random.Generator(np.random.PCG64(7))
    store, model = SS.I3CLSimStepStore(8), B.StepStoreModel()
    for identifier in (3, 9, 4):
        for s in random_steps(rng, 500, identifier):
            store.insert_copy(s); model.insert_copy(s)
    assert store.size() == model.size() == 1500 and [store.count(i) for i in (3, 4, 9, 11)] == [500, 500, 500, 0]
    got, exp = store.pop_bunch_to_vector(700), model.pop_bunch_to_vector(700)
    assert same(got, exp)
    assert np.all(np.diff(got["num"].astype(np.int64)) >= 0)                       # ascending photon count
    assert [store.count(i) for i in (3, 4, 9)] == [model.count(i) for i in (3, 4, 9)] and store.size() == 800
    # more than is left: the store is emptied; with a template the rest is filled
    fill = SS.no_op_step()
    got, exp = store.pop_bunch_to_vector(1000, fill=fill), model.pop_bunch_to_vector(1000, fill=fill[0])
    assert same(got, exp) and store.empty() and np.all(got["num"][800:] == 0) and np.all(got["weight"][800:] == 0)
    assert store.pop_bunch_to_vector(5).shape == (0,) and store.count(3) == 0


def test_fifo_within_a_photon_count():
    store = SS.I3CLSimStepStore()
    s = np.zeros(6, dtype=STEP_DTYPE)
    s["num"] = [5, 2, 5, 2, 5, 0]
    s["id"] = [10, 11, 12, 13, 14, 15]
    store.insert_copy(s)
    assert list(store.pop_bunch_to_vector(6)["id"]) == [15, 11, 13, 10, 12, 14]


@pytest.mark.parametrize("granularity,max_bunch", [(1, 64), (64, 256), (256, 256)])
def test_bunching_follows_the_feeder_thread(granularity, max_bunch):
    rng = np.random.Generator(np.random.PCG64(11))
    sources = [(identifier, random_steps(rng, int(rng.integers(1, 400)), identifier)) for identifier in (100, 7, 55, 8)]
    exp = B.bunch_steps_model(sources, max_bunch, granularity, SS.no_op_step()[0])
    b = SS.StepBuncher(max_bunch, granularity)
    got = []
    for identifier, steps in sources:
        b.begin_light_source(identifier)
        got.extend(b.emit(steps))
    got.extend(b.flush())
    assert len(got) == len(exp)
    for (gs, gf, gl), (es, ef, el) in zip(got, exp):
        assert same(gs, es) and gf == ef and gl == el
    assert all(len(s) == max_bunch for s, _, last in got if not last) and len(got[-1][0]) % granularity == 0
    # every light source is reported finished exactly once, in order, and never before its last step has left
    reported = [i for _, f, _ in got for i in f]
    assert reported == [100, 7, 55, 8]
    total = sum(int((s["num"] > 0).sum() + ((s["num"] == 0) & (s["beta"] == 1) & (s["weight"] == 1)).sum()) for s, _, _ in got)
    assert total == sum(len(st) for _, st in sources)


def test_padding_quirk_of_the_reference():
    """Async.cxx:256: ((size/granularity)+1)*granularity -- a store that already holds a multiple of the granularity
    still gets one whole granule of no-op steps; an empty store yields one granule."""
    store = SS.I3CLSimStepStore()
    assert store.size_with_dummy_fill(64) == 64 and store.size_with_dummy_fill(1) == 0
    s = np.zeros(128, dtype=STEP_DTYPE); s["num"] = 3
    store.insert_copy(s)
    assert store.size_with_dummy_fill(64) == 192 and store.size_with_dummy_fill(1) == 128 and store.size_with_dummy_fill(100) == 200
    with pytest.raises(I3CLSimStepToPhotonConverter_exception):
        SS.StepBuncher(100, 64)


def test_bulk_insertion_and_run_wise_popping_equal_the_step_by_step_model():
    """round 5: the feeder inserts a light source's steps run by run (equal photon count and identifier: one range insertion, one
    count) and bunches leave a FIFO in one copy; the store must stay the reference's (I3CLSimStepStore.h:163-198, 266-296) whatever the
    run structure: long runs (a cascade: all steps but the last carry 200 photons), single steps, interleaved identifiers, partial
    pops that cut runs and identifiers in two."""
    rng = np.random.Generator(np.random.PCG64(21))
    store, model = SS.I3CLSimStepStore(4), B.StepStoreModel()
    for round_ in range(6):
        parts = []
        for identifier in (5, 6, 5, 9):
            run = random_steps(rng, int(rng.integers(1, 300)), identifier, max_photons=3)
            if rng.random() < 0.5:
                run["num"] = 200                    # one long run
            parts.append(run)
        steps = np.concatenate(parts)
        store.insert_copy(steps)                     # a whole array: one call into clsimhip_step_store_insert
        for s in steps:
            model.insert_copy(s)
        assert store.size() == model.size()
        assert [store.count(i) for i in (5, 6, 9, 1)] == [model.count(i) for i in (5, 6, 9, 1)]
        k = int(rng.integers(1, store.size()))
        assert same(store.pop_bunch_to_vector(k), model.pop_bunch_to_vector(k))
        assert [store.count(i) for i in (5, 6, 9)] == [model.count(i) for i in (5, 6, 9)]
    n = store.size()
    assert same(store.pop_bunch_to_vector(n + 7), model.pop_bunch_to_vector(n + 7)) and store.empty()
