#!/usr/bin/env python3
"""Experiment: launch geometry scan (workgroups per CU x slices x k_search) per bunch size.  ANALYSIS TOOL.
usage: exp_geometry_scan.py n [n ...]"""
import devlib  # noqa: F401  (the developer build of the library: this tool steers it through the environment)
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clsim_amd import converter as CV, synthetic as S

medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
bias = CV.GetIceCubeDOMAcceptance(); gen = CV.makeCherenkovWavelengthGenerator(bias, medium)
geom = CV.I3CLSimSimpleGeometry.from_dict(S.ic86_geometry())
dev = torch.device("cuda", 0)
cap = 8 << 20
out = torch.empty((cap, 80), dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
for n in (int(v) for v in sys.argv[1:]):
    steps = S.cascade_steps(n, seed=1000)
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    rows = []
    for per_cu in (5, 6, 7):
        for sl in (8, 12, 16, 24):
            for ks in (3, 5):
                os.environ["CLSIMHIP_GRID"] = str(256 * per_cu); os.environ["CLSIMHIP_SLICES"] = str(sl); os.environ["CLSIMHIP_K_SEARCH"] = str(ks)
                conv = CV.initializeHIP(0, geom, medium, bias, [gen], pancakeFactor=5.0, approximateNumberOfWorkItems=n, seed=12345)
                best = 1e9
                for rep in range(3):
                    conv.KernelTimeMs(reset=True)
                    conv.PropagateDevice(d_steps.data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    ms, _ = conv.KernelTimeMs(reset=True)
                    if rep: best = min(best, ms)
                rows.append((best, per_cu, sl, ks))
                del conv
    rows.sort()
    print("n %d (r at 7/CU %.2f): " % (n, n / (1792 * 256.0)) + "  ".join("%d/CU S%d k%d %.1fms" % (p, s, k, t) for t, p, s, k in rows[:6])
          + "  | worst %.1f" % rows[-1][0], flush=True)
    for p in (5, 6, 7):
        b = min(r for r in rows if r[1] == p)
        print("      best at %d/CU: S%d k%d %.1f ms (%.3e ph/s)" % (p, b[2], b[3], b[0], n * 200 / b[0] * 1e3), flush=True)
