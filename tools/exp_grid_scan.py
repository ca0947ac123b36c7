#!/usr/bin/env python3
"""Experiment: kernel time per bunch size / grid / slices with the product library.  ANALYSIS TOOL.
usage: exp_grid_scan.py n:grid:slices ...   (0 = automatic)"""
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
cache = {}
for spec in sys.argv[1:]:
    n, grid, sl = (int(v) for v in spec.split(":"))
    os.environ.pop("CLSIMHIP_GRID", None); os.environ.pop("CLSIMHIP_SLICES", None)
    if grid: os.environ["CLSIMHIP_GRID"] = str(grid)
    if sl: os.environ["CLSIMHIP_SLICES"] = str(sl)
    conv = CV.initializeHIP(0, geom, medium, bias, [gen], pancakeFactor=5.0, approximateNumberOfWorkItems=n, seed=12345)
    dev = torch.device("cuda", 0)
    if n not in cache:
        steps = S.cascade_steps(n, seed=1000)
        cache[n] = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    cap = 8 << 20
    out = torch.empty((cap, 80), dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(3):
        conv.KernelTimeMs(reset=True)
        conv.PropagateDevice(cache[n].data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ms, _ = conv.KernelTimeMs(reset=True)
        if rep: best = min(best, ms)
    print("n %8d grid %5d slices %2d: %.1f ms %.3e ph/s" % (n, grid, sl, best, n * 200 / best * 1e3), flush=True)
    del conv, out
