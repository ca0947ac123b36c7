#!/usr/bin/env python3
"""How many 64-byte table sectors do k consecutive path samples of ONE photon touch (BUILD CONTAINER TOOL, imports oracle/)?
The table maker sits on the memory side's atomic request rate: one request per wave instruction and sector.  A wave instruction that holds k consecutive
samples of a photon makes (sectors per k samples) requests for them; round 5 put one segment (2.1 samples) of every photon into an instruction, round 6 two
consecutive segments.  Default table (200 x 36 x 100 x 105 bins, tiled 4 x 2 x 1 per sector), SPICE-Mie, one photon per stream, entries from the oracle.
   --five-axis: the table of `bench.py --workload tab5` (200 x 12 x 50 x 105 x 10 with the impact-angle axis, kept in the reference's linear order on the
   device: a sample's fifth bin is drawn at random, ten bins = 80 bytes = the unit that stays together)."""
import numpy as np, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import builders as B, capi
from clsim_amd import synthetic as S, converter as CV
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
med = B.load_ppc_ice(os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
five = "--five-axis" in sys.argv
o_axes = [B.power_axis(0, 580, 200, 2), B.linear_axis(0, 180, 36), B.linear_axis(-1, 1, 100), B.power_axis(0, 7e3, 105, 2)]
if five:
    o_axes = [B.power_axis(0, 580, 200, 2), B.linear_axis(0, 180, 12), B.linear_axis(-1, 1, 50), B.power_axis(0, 7e3, 105, 2), B.linear_axis(-1, 1, 10)]
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
if five:
    print("sectors per sample when an instruction holds k consecutive samples of a photon, linear layout (8-byte bins, 64-byte sectors)")
    secs = [ent["index"][i, :int(num[i])].astype(np.int64) * 8 // 64 for i in range(m)]
    for c in (1, 2, 3, 4, 6, 8, 16):
        req = sum(len(np.unique(s_[j:j + c])) for s_ in secs for j in range(0, len(s_), c))
        print("k=%-3d %.3f" % (c, req / sum(len(s_) for s_ in secs)))
    sys.exit(0)
import itertools
decoded = []
for i in range(m):
    k = int(num[i]); idx = ent["index"][i, :k].astype(np.int64)
    b0 = idx // strides[0]; r = idx % strides[0]; b1 = r // strides[1]; r = r % strides[1]; b2 = r // strides[2]; b3 = r % strides[2]
    decoded.append((b0, b1, b2, b3))
def requests_per_sample(bits, chunk):
    """bits: (e0, e1, e2, e3) -- a sector holds 2^e0 x 2^e1 x 2^e2 x 2^e3 bins of distance, azimuth, polar angle, time"""
    req = n = 0
    for b0, b1, b2, b3 in decoded:
        sec = (((b0 >> bits[0]) * 64 + (b1 >> bits[1])) * 128 + (b2 >> bits[2])) * 128 + (b3 >> bits[3])
        k = len(sec)
        for s_ in range(0, k, chunk):
            req += len(np.unique(sec[s_:s_ + chunk]))
        n += k
    return req / n
print("sectors per sample when an instruction holds k consecutive samples of a photon (a segment is 2.1 samples: k = 2 is round 5, k = 4 round 6)")
print("%-28s %s" % ("tile (dist x azi x polar x time)", " ".join("k=%-5d" % c for c in (1, 2, 3, 4, 6, 8, 16))))
shapes = [b for b in itertools.product(range(4), repeat=4) if sum(b) == 3]
rows = []
for bits in shapes:
    rows.append((requests_per_sample(bits, 4), bits, [requests_per_sample(bits, c) for c in (1, 2, 3, 4, 6, 8, 16)]))
for _, bits, vals in sorted(rows):
    print("%-28s %s" % (" x ".join(str(1 << e) for e in bits) + ("   <- shipped" if bits == (2, 0, 1, 0) else ""), " ".join("%.3f " % v for v in vals)))
