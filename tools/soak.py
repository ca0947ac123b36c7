#!/usr/bin/env python3
"""Soak test (GPU): many bunches through EnqueueSteps/GetConversionResult with double buffering and photon
histories from several producer threads; checks identifiers, determinism of a repeated bunch and host memory."""
import os, sys, threading, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clsim_amd import converter as CV, synthetic as S
from tests import common

cfg = common.config("lea")
# usage: soak.py [bunches] [pool]   -- "pool": bunches large enough for the pooled kernel, no photon histories
POOL = len(sys.argv) > 2 and sys.argv[2] == "pool"
n = 786432 if POOL else 16384
HIST = 0 if POOL else 3
bias = CV.GetIceCubeDOMAcceptance()
conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_dict(cfg["geom"]), cfg["med_p"], bias, [CV.makeCherenkovWavelengthGenerator(bias, cfg["med_p"])],
                        pancakeFactor=5.0, photonHistoryEntries=HIST, enableDoubleBuffering=True, approximateNumberOfWorkItems=n, seed=7)
assert conv.KernelForBunch(n) == ("pool" if POOL else "classic")
bunches = [S.cascade_steps(n, seed=s) for s in range(4)]
total = int(sys.argv[1]) if len(sys.argv) > 1 else 400
lock = threading.Lock(); nxt = [0]
def producer():
    while True:
        with lock:
            i = nxt[0]; nxt[0] += 1
        if i >= total: return
        conv.EnqueueSteps(bunches[i % 4], i)
threads = [threading.Thread(target=producer) for _ in range(3)]
t0 = time.time(); rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for t in threads: t.start()
seen, hits = set(), 0
for k in range(total):
    if HIST:
        ident, ph, hist = conv.GetConversionResult(with_histories=True)
        assert len(hist) == len(ph) and all(len(h) == min(int(s), 3) for h, s in zip(hist, ph["numScatters"]))
    else:
        ident, ph = conv.GetConversionResult()
    assert ident not in seen and 0 <= ident < total
    seen.add(ident); hits += len(ph)
    if k == min(50, total // 2): rss50 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss      # after the warm-up half (or 50 bunches)
for t in threads: t.join()
st = conv.GetStatistics()
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("soak ok: %d bunches, %d hits, %.1f s, kernel calls %d, utilisation %.2f, maxrss %d -> %d -> %d KB" % (total, hits, time.time() - t0, st["NumKernelCalls"], st["DeviceUtilization"], rss0, rss50, rss1))
assert st["NumKernelCalls"] == total and not conv.MorePhotonsAvailable() and conv.QueueSize() == 0
assert rss1 - rss50 < 200 * 1024, "host memory keeps growing"
