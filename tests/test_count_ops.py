"""The counting build of the oracle (oracle/count_ops.hpp, `make -C oracle liboracle_count.so`): the restatement compiled as C++ with
float operators that count themselves.  It must BE the oracle -- same records, same streams, bit for bit -- and its counters must
satisfy the identities the reference's draw order implies.  The stored figures bench.py quotes
(profiles/r06/reference_ops.json) must come from the oracle sources as they are now."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import capi
from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,stop,births", [("mie", True, 4), ("lea", True, 4), ("flasher", True, 2), ("c1", True, 4), ("clear", False, 4), ("photonics_mie", True, 4)])
def test_counting_build_is_the_oracle_and_counts_every_draw(name, stop, births):
    cfg = common.config(name)
    steps = common.steps_for(cfg, 512, seed=7)
    x, a = common.streams(len(steps))
    T = common.oracle_tables(cfg, stop_detected=stop)
    ref = capi.propagate(T, steps, x, a, threads=4)
    got, ops, ev = capi.count_ops(T, steps, x, a, threads=4)
    assert got[1] == ref[1] and got[3] == ref[3]
    assert capi.sort_photons(got[0]).tobytes() == capi.sort_photons(ref[0]).tobytes()
    assert np.array_equal(got[2], ref[2])
    # the checker is back in place afterwards
    again = capi.propagate(T, steps, x, a, threads=4)
    assert again[1] == ref[1] and np.array_equal(again[2], ref[2])
    photons = int(steps["num"].sum())
    assert ev["photons"] == photons and ev["trips"] == ref[3] and ev["steps"] == len(steps)
    assert ev["scatters"] + photons == ev["trips"]                      # every trip ends in a scatter or in the photon's end
    draws = sum(d.get("rng_draw", 0) for d in ops.values())
    # position, wavelength (not for a delta-peak spectrum), azimuth (Cherenkov only), absorption budget; one per trip; two per scatter
    assert draws == births * photons + ev["trips"] + 2 * ev["scatters"]
    logs = sum(d.get("log", 0) for d in ops.values())
    assert logs == photons + ev["trips"]                                # -log(u) for the budget and for every scattering step
    if stop:
        assert ev["hits"] == ref[1]
    assert ev["search_calls"] == ev["trips"]                            # the reference searches on every trip (c.cl:704)
    if name in ("mie", "lea", "flasher"):
        assert ev["liu"] + ev["hg"] == ev["scatters"]
        evals = ops["layer_lengths"]
        assert evals["powr"] == 2 * ev["layer_length_evals"] and evals["exp"] == ev["layer_length_evals"]
        assert ev["layer_length_evals"] == ev["trips"] + ev["layer_crossings"]


def test_stored_reference_ops_are_current():
    path = os.path.join(ROOT, "profiles", "r06", "reference_ops.json")
    with open(path) as f:
        stored = json.load(f)
    src = b"".join(open(os.path.join(ROOT, "oracle", f), "rb").read() for f in ("clsim_oracle.c", "oracle_math.h", "count_ops.hpp", "count_ops_calls.hpp"))
    assert stored["oracle_sha16"] == hashlib.sha256(src).hexdigest()[:16], "oracle sources changed: run tools/count_reference_ops.py again"
    for w in ("c2", "c3", "c5"):
        v = stored["workloads"][w]["valu_per_photon"]
        assert v["as_written"] > v["as_written_without_search"] > v["transformed"] > 1000
