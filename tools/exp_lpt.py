#!/usr/bin/env python3
"""Experiment: does handing out expensive steps first (LPT) shorten the tail?  Sorts the bench's step
array on the host by an estimated cost (photons x absorption/scattering length ratio of the step's ice
layer at 400 nm) and times the kernel.  ANALYSIS TOOL."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clsim_amd import converter as CV, synthetic as S

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
    bias = CV.GetIceCubeDOMAcceptance(); gen = CV.makeCherenkovWavelengthGenerator(bias, medium)
    geom = CV.I3CLSimSimpleGeometry.from_dict(S.ic86_geometry())
    conv = CV.initializeHIP(0, geom, medium, bias, [gen], pancakeFactor=5.0, approximateNumberOfWorkItems=n, seed=12345)
    steps = S.cascade_steps(n, seed=1000)
    d = medium.describe()
    # per-layer cost ~ iterations per photon = absorption length / geometric scattering length at 400 nm
    x = 400.0
    absl = 1.0 / ((d["D"] * d["a_dust400"] + d["E"]) * x ** (-d["kappa"]) + d["A"] * np.exp(-d["B"] / x) * (1 + 0.01 * d["delta_tau"]))
    scal = 1.0 / d["b400"]
    w = absl / scal
    layer = np.clip(((steps["z"] - d["layers_z_start"]) / d["layers_height"]).astype(int), 0, d["num_layers"] - 1)
    key = steps["num"] * w[layer]
    dev = torch.device("cuda", 0)
    cap = 4 << 20
    out = torch.empty((cap, 80), dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    for name, order in (("unsorted", np.arange(n)), ("descending cost", np.argsort(-key, kind="stable")), ("ascending cost", np.argsort(key, kind="stable")),
                        ("descending z", np.argsort(-steps["z"], kind="stable"))):
        st = steps[order]
        d_steps = torch.from_numpy(st.view(np.uint8).reshape(n, 48).copy()).to(dev)
        for rep in range(3):
            conv.KernelTimeMs(reset=True)
            conv.PropagateDevice(d_steps.data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ms, _ = conv.KernelTimeMs(reset=True)
        print("%-16s kernel %.1f ms  %.4g photons/s  hits %d" % (name, ms, steps["num"].sum() / ms * 1e3, int(cnt.item())))

main()
