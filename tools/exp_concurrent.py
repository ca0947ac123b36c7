#!/usr/bin/env python3
"""Small bunches side by side (GPU).  ANALYSIS TOOL.  k bunches of n steps on k HIP streams with
clsimhip_set_concurrent_device_launches(k) against the same bunches one after the other and against one bunch of k*n steps.
usage: exp_concurrent.py [n=262144] [k=4] [repeats=5]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clsim_amd import converter as CV, synthetic as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
bias = CV.GetIceCubeDOMAcceptance(); gen = CV.makeCherenkovWavelengthGenerator(bias, medium)
geom = CV.I3CLSimSimpleGeometry.from_dict(S.ic86_geometry())
dev = torch.device("cuda", 0)
conv = CV.initializeHIP(0, geom, medium, bias, [gen], pancakeFactor=5.0, approximateNumberOfWorkItems=k * n, seed=12345)
steps = S.cascade_steps(k * n, seed=1000)
d_steps = torch.from_numpy(steps.view(np.uint8).reshape(k * n, 48).copy()).to(dev)
cap = 1 << 20
outs = [torch.empty((cap, 80), dtype=torch.uint8, device=dev) for _ in range(k)]
cnts = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(k)]
streams = [torch.cuda.Stream(device=dev) for _ in range(k)]
photons = float(steps["num"].sum())


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if mode == "one":
            conv.PropagateDevice(d_steps.data_ptr(), k * n, outs[0].data_ptr(), cap, cnts[0].data_ptr(), stream=streams[0].cuda_stream)
        else:
            for j in range(k):
                s = streams[j if mode == "side by side" else 0]
                conv.PropagateDevice(d_steps.data_ptr() + 48 * j * n, n, outs[j].data_ptr(), cap, cnts[j].data_ptr(), stream=s.cuda_stream, rng_offset=j * n)
    torch.cuda.synchronize()
    return photons * reps / (time.perf_counter() - t0)


for mode, share in (("one", 1), ("in sequence", 1), ("side by side", k), ("one", 1)):
    conv.SetConcurrentDeviceLaunches(share)
    run(mode)
    label = (1, k * n, mode) if mode == "one" else (k, n, mode)
    print("%d x %d steps, %-13s: %.3e photons/s" % (label + (run(mode),)), flush=True)
