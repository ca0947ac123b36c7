#!/usr/bin/env python3
"""Scan of the pooled kernel's scheduling parameters (GPU).  ANALYSIS TOOL.
usage: exp_pool_scan.py [workload] spec ...   with spec = key=value,key=value  (keys: kernel R pop new slices search grid n)
Prints kernel time and photons/s per spec."""
import devlib  # noqa: F401  (the developer build of the library: this tool steers it through the environment)
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clsim_amd import converter as CV, synthetic as S

ENV = {"kernel": "CLSIMHIP_KERNEL", "R": "CLSIMHIP_POOL_R", "pop": "CLSIMHIP_K_POP", "new": "CLSIMHIP_K_NEW", "slices": "CLSIMHIP_SLICES",
       "search": "CLSIMHIP_K_SEARCH", "grid": "CLSIMHIP_GRID"}
args = sys.argv[1:]
workload = "c2"
if args and args[0] in ("c2", "c3", "c5"):
    workload = args.pop(0)
ice = "spice_mie" if workload == "c2" else "spice_lea"
medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", ice))
bias = CV.GetIceCubeDOMAcceptance()
gens = [CV.makeCherenkovWavelengthGenerator(bias, medium)]
g86 = S.ic86_geometry()
geom = CV.I3CLSimSimpleGeometry.from_dict(g86)
if workload == "c5":
    gens.append(CV.I3CLSimRandomValueConstant(405e-9))
dev = torch.device("cuda", 0)
cache = {}
for spec in args:
    kv = dict(item.split("=") for item in spec.split(",") if item)
    for k, e in ENV.items():
        os.environ.pop(e, None)
        if k in kv:
            os.environ[e] = kv[k]
    n = int(kv.get("n", 1 << 20))
    if n not in cache:
        if workload == "c5":
            k = int(np.argmin(np.abs(g86["x"]) + np.abs(g86["y"]) + np.abs(g86["z"] + 100.0)))
            st = S.flasher_steps(n, seed=1000, photons_per_step=400, position=(float(g86["x"][k]), float(g86["y"][k]), float(g86["z"][k])))
        else:
            st = S.cascade_steps(n, seed=1000, photons_per_step=200)
        cache[n] = (st, torch.from_numpy(st.view(np.uint8).reshape(n, 48).copy()).to(dev))
    st, d_steps = cache[n]
    conv = CV.initializeHIP(0, geom, medium, bias, gens, pancakeFactor=5.0, approximateNumberOfWorkItems=n, seed=12345)
    cap = (4 if workload != "c5" else 24) << 20
    out = torch.empty((cap, 80), dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    best = 1e30
    for rep in range(int(kv.get("reps", 3))):
        conv.KernelTimeMs(reset=True)
        conv.PropagateDevice(d_steps.data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ms, _ = conv.KernelTimeMs(reset=True)
        if rep:
            best = min(best, ms)
    print("%s %-60s %8.2f ms  %.4g photons/s  hits %d" % (workload, spec, best, st["num"].sum() / best * 1e3, int(cnt.item())), flush=True)
    del conv, out
