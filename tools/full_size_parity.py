#!/usr/bin/env python3
"""One-off check on the GPU box: a PRODUCTION-size bunch of BASELINE configs[2] (c3: 5 242 880 cascade steps x 200 photons,
SPICE-Lea) or configs[4] (c5: 2 621 440 flasher steps x 400 photons at a DOM) through the kernel's production schedule
against the oracle run on all host cores for the WHOLE bunch -- every detected photon (80 bytes each, as a sorted multiset)
and every final RNG state, bit for bit.  (tests/test_production_size_gpu.py checks the first 2048 steps of such a launch
in every test run; this takes 5-7 minutes of oracle time per workload.)   usage: full_size_parity.py c2|c3|c5|c2keep|c5keep [chunks=6]
(c2keep: BASELINE configs[1], 1 048 576 cascade steps in SPICE-Mie, c5keep: the flasher bunch -- both WITHOUT
STOP_PHOTONS_ON_DETECTION, SetStopDetectedPhotons(false))"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from clsim_amd import synthetic as S
from clsim_amd.synthetic import PHOTON_DTYPE
from oracle import capi
from tests import common

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
keep = which.endswith("keep")
if which in ("c2keep", "c2"):
    cfg = common.config("mie")
    n = 1 << 20
    steps = S.cascade_steps(n, seed=1000, photons_per_step=200)
    capacity = 4 << 20
elif which == "c3":
    cfg = common.config("lea")
    n = 5 * (1 << 20)
    steps = S.cascade_steps(n, seed=1000, photons_per_step=200)
    capacity = 8 << 20
else:
    cfg = common.config("flasher")
    g = cfg["geom"]
    k = int(np.argmin(np.abs(g["x"]) + np.abs(g["y"]) + np.abs(g["z"] + 100.0)))
    n = 2621440
    steps = S.flasher_steps(n, seed=1000, photons_per_step=400, position=(float(g["x"][k]), float(g["y"][k]), float(g["z"][k])))
    capacity = 48 << 20
capi.build()
x, a = common.streams(n)
dev = torch.device("cuda", 0)
conv = common.product_converter(cfg, n, stop_detected=not keep)
d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
d_out = torch.empty((capacity, 80), dtype=torch.uint8, device=dev)
d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
t0 = time.time()
conv.PropagateDevice(d_steps.data_ptr(), n, d_out.data_ptr(), capacity, d_cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
cnt = int(d_cnt.item())
assert cnt <= capacity
print("%s: kernel %s, %d steps, %.3g photons, %d detected, %.2f s" % (which, conv.KernelForBunch(n), n, float(steps["num"].sum()), cnt, time.time() - t0), flush=True)
got = np.frombuffer(d_out[:cnt].cpu().numpy().tobytes(), dtype=PHOTON_DTYPE)
x_dev = conv.GetRNGState(n)
del d_out
T = common.oracle_tables(cfg, stop_detected=not keep)
parts, x_parts = [], []
threads = os.cpu_count() or 8
for c in range(chunks):
    lo, hi = (n * c) // chunks, (n * (c + 1)) // chunks
    t1 = time.time()
    ph, c_o, x_o, _ = capi.propagate(T, steps[lo:hi], x[lo:hi], a[lo:hi], threads=threads)
    parts.append(ph)                              # string / DOM indices, as the device path delivers them
    x_parts.append(x_o)
    print("  oracle chunk %d/%d: steps %d-%d, %d detected, %.0f s on %d threads" % (c + 1, chunks, lo, hi, c_o, time.time() - t1, threads), flush=True)
want = np.concatenate(parts)
assert len(want) == cnt, (len(want), cnt)
ok_records = common.sort_photons(want).tobytes() == common.sort_photons(got.copy()).tobytes()
ok_streams = np.array_equal(np.concatenate(x_parts), x_dev)
print("%s full size: %d detected photons %s, %d final RNG states %s" % (which, cnt, "IDENTICAL" if ok_records else "DIFFER", n, "IDENTICAL" if ok_streams else "DIFFER"), flush=True)
sys.exit(0 if (ok_records and ok_streams) else 1)
