#!/usr/bin/env python3
"""Experiment: work-unit slicing -- kernel time and wait counters per slice count.  ANALYSIS TOOL
(needs a library built with -DCLSIMHIP_DEBUG_COUNTERS, CLSIMHIP_LIB=...)."""
import devlib  # noqa: F401  (the developer build of the library: this tool steers it through the environment)
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clsim_amd import converter as CV, synthetic as S, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
bias = CV.GetIceCubeDOMAcceptance(); gen = CV.makeCherenkovWavelengthGenerator(bias, medium)
geom = CV.I3CLSimSimpleGeometry.from_dict(S.ic86_geometry())
x = a = None
for sl in [int(v) for v in (sys.argv[2:] or ["1", "2", "4", "8", "16", "24"])]:
    os.environ["CLSIMHIP_SLICES"] = str(sl)
    conv = CV.initializeHIP(0, geom, medium, bias, [gen], pancakeFactor=5.0, approximateNumberOfWorkItems=n, seed=12345)
    steps = S.cascade_steps(n, seed=1000)
    dev = torch.device("cuda", 0)
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    cap = 4 << 20
    out = torch.empty((cap, 80), dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    for rep in range(2):
        conv.KernelTimeMs(reset=True)
        conv.PropagateDevice(d_steps.data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ms, _ = conv.KernelTimeMs(reset=True)
    c = (C.c_uint32 * 4)()
    lib = _lib.load(); lib.clsimhip_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    lib.clsimhip_debug_counters(conv._h, c)
    print("slices %2d: %.1f ms  units handed %d  max photons %d  failed polls %d  all-waiting sleeps %d" % (sl, ms, c[0], c[1], c[2], c[3]))
    del conv
