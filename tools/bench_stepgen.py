#!/usr/bin/env python3
"""Rate of the GPU step producer (steps born in HBM): tools/bench_stepgen.py [n_steps]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clsim_amd import converter as CV
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 1024 * 1024
req = np.zeros(64, dtype=CV.REQUEST_DTYPE)
rng = np.random.Generator(np.random.PCG64(1))
for i in range(64):
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    req[i] = (rng.uniform(-400, 400), rng.uniform(-400, 400), rng.uniform(-400, 400), 0.0, d[0], d[1], d[2], 0.0,
              rng.uniform(2.0, 6.0), 0.6, 0, i, 200, 0, n // 64)
dev = torch.device("cuda", 0)
buf = torch.zeros((n, 48), dtype=torch.uint8, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = CV.GenerateStepsDevice(req, 5 + rep, buf.data_ptr(), n, granularity=256)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%d cascade steps in %.2f ms: %.3g steps/s (= %.3g photons/s of work for the propagator)" % (got, dt * 1e3, got / dt, 200 * got / dt))

# flasher pulses: 64 LED pulses of 5e7 photons each (400 per step)
cfg = CV.FlasherStepConverterConfig((CV.DIST_NORMAL, 0.0), (CV.DIST_NORMAL, 0.0), (CV.DIST_FLASHER_TIME_PROFILE, 0.0), False,
                                    photonsPerStep=400, maxBunchSize=512000, bunchSizeGranularity=512)
q = np.zeros(64, dtype=CV.FLASHER_REQUEST_DTYPE)
for i in range(64):
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    q[i] = (rng.uniform(-400, 400), rng.uniform(-400, 400), rng.uniform(-400, 400), 0.0, d[0], d[1], d[2], 0.17, 0.17,
            35.0 + (i % 4), i, 1, 50_000_000 + 7 * i)
total, real = CV.CountFlasherSteps(cfg, q)
fbuf = torch.zeros((total, 48), dtype=torch.uint8, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = CV.GenerateFlasherStepsDevice(cfg, q, 11 + rep, fbuf.data_ptr(), total)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%d flasher steps (%d with photons) in %.2f ms: %.3g steps/s (= %.3g photons/s of work for the propagator)"
      % (got, real, dt * 1e3, got / dt, 400 * real / dt))
