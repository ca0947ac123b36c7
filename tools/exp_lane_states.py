#!/usr/bin/env python3
"""Experiment: lane-state census of prop_kernel per bunch size / grid / slices.  ANALYSIS TOOL: needs the instrumented
library built by the recipe in DESIGN.md 5 ("lane census"), CLSIMHIP_LIB=... ; prints fractions of lane-trips."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clsim_amd import converter as CV, synthetic as S, _lib

medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
bias = CV.GetIceCubeDOMAcceptance(); gen = CV.makeCherenkovWavelengthGenerator(bias, medium)
geom = CV.I3CLSimSimpleGeometry.from_dict(S.ic86_geometry())
for spec in sys.argv[1:]:
    n, grid, sl = (int(v) for v in spec.split(":"))
    os.environ.pop("CLSIMHIP_GRID", None); os.environ.pop("CLSIMHIP_SLICES", None)
    if grid: os.environ["CLSIMHIP_GRID"] = str(grid)
    if sl: os.environ["CLSIMHIP_SLICES"] = str(sl)
    conv = CV.initializeHIP(0, geom, medium, bias, [gen], pancakeFactor=5.0, approximateNumberOfWorkItems=n, seed=12345)
    steps = S.cascade_steps(n, seed=1000)
    dev = torch.device("cuda", 0)
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    cap = 8 << 20
    out = torch.empty((cap, 80), dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    for rep in range(2):
        conv.KernelTimeMs(reset=True)
        conv.PropagateDevice(d_steps.data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ms, _ = conv.KernelTimeMs(reset=True)
    c = (C.c_uint64 * 8)()
    lib = _lib.load(); lib.clsimhip_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    lib.clsimhip_debug_counters(conv._h, c)
    trips, run, need, wait, parked, dead, phases, created = (float(v) for v in c)
    lanes = 64.0 * trips
    print("n %8d grid %5d slices %2d: %.1f ms %.3e ph/s | wave trips %.3e | run %.1f%% need %.1f%% wait-pred %.1f%% parked %.1f%% dead %.1f%% | "
          "creation phases/trip %.3f lanes/phase %.1f" % (n, grid, sl, ms, n * 200 / ms * 1e3, trips, 100 * run / lanes, 100 * need / lanes,
                                                         100 * wait / lanes, 100 * parked / lanes, 100 * dead / lanes, phases / trips, created / max(phases, 1)))
    del conv
