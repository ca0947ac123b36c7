#!/usr/bin/env python3
"""Experiment: when do the waves of prop_kernel see the queue run dry / finish.  ANALYSIS TOOL (instrumented library)."""
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
    buf = np.zeros(1 << 17, dtype=np.uint64)
    lib = _lib.load(); lib.clsimhip_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    lib.clsimhip_debug_counters(conv._h, buf.ctypes.data_as(C.c_void_p))
    t0 = int(buf[8])
    nw = (grid if grid else 1280) * 4
    rec = buf[16:16 + 3 * nw].reshape(nw, 3).astype(np.int64)
    end = (rec[:, 0] - t0) / 100e3          # ms at 100 MHz
    dry = np.where(rec[:, 1] > 0, (rec[:, 1] - t0) / 100e3, np.nan)
    trips = rec[:, 2]
    q = lambda a, p: float(np.nanpercentile(a, p))
    print("run %.1f%% wait-pred %.1f%% |" % (100.0 * buf[1] / (64.0 * buf[0]), 100.0 * buf[3] / (64.0 * buf[0])), end=" ")
    print("n %d grid %d slices %d: kernel %.1f ms | first dry sub-queue seen: min %.1f p10 %.1f p50 %.1f p90 %.1f | wave end: p1 %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | trips/wave p10 %d p50 %d p90 %d"
          % (n, grid, sl, ms, np.nanmin(dry), q(dry, 10), q(dry, 50), q(dry, 90), q(end, 1), q(end, 10), q(end, 50), q(end, 90), end.max(),
             q(trips, 10), q(trips, 50), q(trips, 90)), flush=True)
    del conv
