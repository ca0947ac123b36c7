#!/usr/bin/env python3
"""Lane-state census and per-wave clocks of prop_kernel.  ANALYSIS TOOL.

Needs the analysis build of the library:
    make -C clsim_amd/csrc clean && make -C clsim_amd/csrc EXTRA=-DCLSIMHIP_CENSUS     (then rebuild without for the product)
usage: exp_census.py n:grid:slices ...      (grid in workgroups, 0 = automatic)
Prints, per configuration: kernel time; fractions of lane trips spent running / waiting for a creation batch / waiting for
a predecessor slice / parked for the DOM search / without work; photon creation batches; when the waves first found a
sub-queue used up and when they ended; trips per wave."""
import devlib  # noqa: F401  (the developer build of the library: this tool steers it through the environment)
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
    buf = np.zeros(1 << 19, dtype=np.uint64)
    lib = _lib.load(); lib.clsimhip_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    lib.clsimhip_debug_counters(conv._h, buf.ctypes.data_as(C.c_void_p))
    trips, run, need, wait, parked, dead, phases, created = (float(v) for v in buf[:8])
    lanes = 64.0 * trips
    t0 = int(buf[8])
    rec = buf[16:16 + 3 * 10917].reshape(-1, 3).astype(np.int64)      # (region counters follow from word 32768)
    rec = rec[rec[:, 0] > 0]
    end = (rec[:, 0] - t0) / 100e3                      # ms at 100 MHz
    dry = np.where(rec[:, 1] > 0, (rec[:, 1] - t0) / 100e3, np.nan)
    q = lambda a, p: float(np.nanpercentile(a, p))
    print("n %d grid %d slices %d: %.1f ms | run %.1f%% need %.1f%% wait-pred %.1f%% parked %.1f%% dead %.1f%% | %.3f creation batches per trip, "
          "%.1f lanes each | first dry sub-queue seen p10 %.1f p50 %.1f | wave end p1 %.1f p50 %.1f p90 %.1f max %.1f ms | trips/wave p10 %d p50 %d p90 %d"
          % (n, grid, sl, ms, 100 * run / lanes, 100 * need / lanes, 100 * wait / lanes, 100 * parked / lanes, 100 * dead / lanes,
             phases / trips, created / max(phases, 1), q(dry, 10), q(dry, 50), q(end, 1), q(end, 50), q(end, 90), end.max(),
             q(rec[:, 2], 10), q(rec[:, 2], 50), q(rec[:, 2], 90)), flush=True)
    del conv
