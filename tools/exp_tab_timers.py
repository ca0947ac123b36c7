#!/usr/bin/env python3
"""Where a table-maker wave spends its time (GPU).  ANALYSIS TOOL.  Needs the analysis build of the library:
    tools/build_variant.sh tab_timers -DCLSIMHIP_TAB_TIMERS      (CLSIMHIP_LIB=build_variants/tab_timers.so selects it)
The kernel sums shader-clock cycles per phase of a loop trip per wave and adds them into the table's first words (the table's
contents are meaningless in that build); CLSIMHIP_TAB_LAYOUT=linear so that those are the first words of GetBinSums().
usage: exp_tab_timers.py [steps per launch]        (bench.py --workload tab's table and steps)"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CLSIMHIP_TAB_LAYOUT"] = "linear"
import numpy as np
import torch
from clsim_amd import converter as CV, synthetic as S, tabulator as TB

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
axes = TB.SphericalAxes([TB.PowerAxis(0, 580, 200, 2), TB.LinearAxis(0, 180, 36), TB.LinearAxis(-1, 1, 100), TB.PowerAxis(0, 7e3, 105, 2)])
ang = [0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435]
a = CV.mwc_multipliers(n)
x = CV.seed_streams(a)
tab = TB.I3CLSimStepToTableConverterHIP(0, axes, False, medium, math.pi * 0.16510 ** 2, CV.GetIceCubeDOMAcceptance(),
                                        TB.I3CLSimFunctionPolynomial(ang), (x, a))
steps = S.cascade_steps(n, seed=1000, vertex=(0.0, 0.0, 0.0), photons_per_step=200)
tab.EnqueueSteps(steps, (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0))
tab.Finish()
ms = tab.GetStatistics()["KernelTimeMs"]
t = tab.GetBinSums().ravel()[:12].astype(np.float64)
names = ["units + creation", "layer walk", "savePath: sample loop", "advance + scattering", "trips", "lanes running", "savePath: counting, prefix sum, lists"]
cyc = np.array([t[0], t[1], t[6], t[2], t[3] + t[8] + t[9] + t[10]])
print("kernel %.1f ms, %d steps x 200 photons; wave trips %.4g, lanes with a photon per trip %.1f" % (ms, n, t[4], t[5] / t[4]))
print("shader-clock cycles per wave trip (sum over the phases %.0f):" % (cyc.sum() / t[4]))
for k in (0, 1, 6, 2, 3):
    print("  %-40s %8.0f  %5.1f %%" % (names[k], t[k] / t[4], 100 * t[k] / cyc.sum()))
print("  of advance + scattering: up to the advance %.0f, position update %.0f, scattering angle %.0f, the rest (rotation, loop end) %.0f" % (t[8] / t[4], t[9] / t[4], t[10] / t[4], t[3] / t[4]))
print("  of the sample loop, around the atomic instruction (between two s_memtime): %.0f cycles per trip" % (t[7] / t[4]))
