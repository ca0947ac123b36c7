#!/usr/bin/env python3
"""How many 64-byte table sectors do k consecutive path samples of ONE photon touch (BUILD CONTAINER TOOL, imports oracle/)?
The table maker sits on the memory side's atomic request rate: one request per wave instruction and sector.  A wave instruction that holds k consecutive
samples of a photon makes (sectors per k samples) requests for them; round 5 put one segment (2.1 samples) of every photon into an instruction, round 6 two
consecutive segments.  Default table (200 x 36 x 100 x 105 bins, tiled 4 x 2 x 1 per sector), SPICE-Mie, one photon per stream, entries from the oracle."""
import numpy as np, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import builders as B, capi
from clsim_amd import synthetic as S, converter as CV
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
med = B.load_ppc_ice(os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
o_axes = [B.power_axis(0, 580, 200, 2), B.linear_axis(0, 180, 36), B.linear_axis(-1, 1, 100), B.power_axis(0, 7e3, 105, 2)]
ang = [0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435]
tb = B.tabulator_config("spherical", o_axes, med, ang, entries_per_stream=80000)
bias = B.icecube_dom_acceptance()
g = S.single_string_geometry()
geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
T = capi.make_tables(med, geo, [B.cherenkov_wlen_generator(bias, med)], bias, pancake=1.0, tabulator=tb)
m=256; nph=1      # one photon per step: a stream's entries are one photon's samples in order
steps = S.cascade_steps(m, seed=1000, vertex=(0.,0.,0.), photons_per_step=nph)
a = CV.mwc_multipliers(m); x = CV.seed_streams(a)
ref_o = B.reference_particle((0,0,0), 0.0, (0,0,1.0))
ent,num,left,xs = capi.tabulate(T, steps, x, a, ref_o, threads=8)
shape=tb["shape"]; strides=tb["strides"]
print("shape",shape)
tot=0; res={}
for chunk in (1,2,3,4,6,8,12,16,32):
    req=0; n=0
    for i in range(m):
        k=int(num[i]); idx=ent["index"][i,:k].astype(np.int64)
        b0=idx//strides[0]; r=idx%strides[0]; b1=r//strides[1]; r=r%strides[1]; b2=r//strides[2]; b3=r%strides[2]
        sec=((b0>>2)*100000+b1)*100000*1000+(b2>>1)*1000+b3
        for s in range(0,k,chunk):
            req+=len(np.unique(sec[s:s+chunk])); 
        n+=k
    res[chunk]=req/n
    print("consecutive samples of a photon per instruction: %2d -> %.3f requests per sample"%(chunk, req/n))
