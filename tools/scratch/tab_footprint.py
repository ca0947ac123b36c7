#!/usr/bin/env python3
"""Table maker: kernel time against the table's footprint (azimuth bins 36 / 18 / 9 / 4: 670 / 335 / 168 / 75 MB of fp64 bins), same photons.
Is the memory-side atomic rate a DRAM (random read-modify-write) limit or a request limit?  ANALYSIS TOOL (GPU)."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from clsim_amd import converter as CV, synthetic as S, tabulator as TB
n = 262144
medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
ang = [0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435]
a = CV.mwc_multipliers(n)
x = CV.seed_streams(a)
steps = S.cascade_steps(n, seed=1000, vertex=(0.0, 0.0, 0.0), photons_per_step=200)
for az in (36, 18, 9, 4):
    axes = TB.SphericalAxes([TB.PowerAxis(0, 580, 200, 2), TB.LinearAxis(0, 180, az), TB.LinearAxis(-1, 1, 100), TB.PowerAxis(0, 7e3, 105, 2)])
    tab = TB.I3CLSimStepToTableConverterHIP(0, axes, False, medium, math.pi * 0.16510 ** 2, CV.GetIceCubeDOMAcceptance(),
                                            TB.I3CLSimFunctionPolynomial(ang), (x, a))
    tab.EnqueueSteps(steps, (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0)); tab.Finish()
    k0 = tab.GetStatistics()["KernelTimeMs"]
    for _ in range(2):
        tab.EnqueueSteps(steps, (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0))
    tab.Finish()
    ms = (tab.GetStatistics()["KernelTimeMs"] - k0) / 2
    print("azimuth bins %2d: %8.1f MB of bins, kernel %.1f ms per pass, %.4g photons/s" % (az, tab.n_bins * 8 / 1e6, ms, n * 200 / ms * 1e3), flush=True)
    del tab
    torch.cuda.empty_cache()
