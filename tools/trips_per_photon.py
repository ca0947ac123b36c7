#!/usr/bin/env python3
"""Loop trips per photon of the bench workloads (CPU, oracle): photons/s is not comparable between workloads whose
photons live for different numbers of scatter-loop iterations.  ANALYSIS TOOL (imports oracle/)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import builders as B, capi
from clsim_amd import synthetic as S

capi.build()
g = S.ic86_geometry()
geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
bias = B.icecube_dom_acceptance()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
a = B.mwc_multipliers(n); x = B.seed_streams(a, 12345)
for name, ice, flasher in (("c2", "spice_mie", False), ("c3", "spice_lea", False), ("c5", "spice_lea", True)):
    med = B.load_ppc_ice(os.path.join(ROOT, "clsim_amd", "data", "ice", ice))
    gens = [B.cherenkov_wlen_generator(bias, med)]
    if flasher:
        gens.append(dict(kind="const", value=405e-9))
        k = int(np.argmin(np.abs(g["x"]) + np.abs(g["y"]) + np.abs(g["z"] + 100.0)))
        steps = S.flasher_steps(n, seed=1000, photons_per_step=400, position=(float(g["x"][k]), float(g["y"][k]), float(g["z"][k])))
    else:
        steps = S.cascade_steps(n, seed=1000, photons_per_step=200)
    T = capi.make_tables(med, geo, gens, bias, pancake=5.0)
    ph, cnt, _, it = capi.propagate(T, steps, x, a, threads=os.cpu_count())
    photons = int(steps["num"].sum())
    print("%s: %d photons, %d loop trips = %.2f trips/photon, %d hits (%.3f %%), mean scatters of hits %.1f"
          % (name, photons, it, it / photons, cnt, 100.0 * cnt / photons, float(ph["numScatters"].mean()) if cnt else 0.0))
