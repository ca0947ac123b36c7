#!/usr/bin/env python3
"""LDS image size of the bench's table-maker configuration (GPU)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clsim_amd import converter as CV, tabulator as TB
medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
axes = TB.SphericalAxes([TB.PowerAxis(0, 580, 200, 2), TB.LinearAxis(0, 180, 36), TB.LinearAxis(-1, 1, 100), TB.PowerAxis(0, 7e3, 105, 2)])
ang = [0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435]
a = CV.mwc_multipliers(256); x = CV.seed_streams(a)
tab = TB.I3CLSimStepToTableConverterHIP(0, axes, False, medium, math.pi * 0.16510 ** 2, CV.GetIceCubeDOMAcceptance(), TB.I3CLSimFunctionPolynomial(ang), (x, a))
print("LDS image words:", tab.GetTable("LDS_IMAGE_WORDS"))
