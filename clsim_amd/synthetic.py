"""
Synthetic workloads of BASELINE.json (definitions: SURVEY.md section 8d).

Pure numpy, no GPU and no oracle: these generators define the INPUTS
(detector geometry, step bunches) that bench.py, the parity tests and the
oracle all consume.  Steps follow what the reference's producers emit
(private/clsim/I3CLSimLightSourceToStepConverterPPC.cxx:785-819 for cascades,
...Flasher.cxx:434-545 for flashers); record layout: public/clsim/I3CLSimStep.h:141-155.
"""
import math

import numpy as np

STEP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("t", "<f4"),
                       ("theta", "<f4"), ("phi", "<f4"), ("length", "<f4"), ("beta", "<f4"),
                       ("num", "<u4"), ("weight", "<f4"), ("id", "<u4"),
                       ("sourceType", "u1"), ("dummy1", "u1"), ("dummy2", "<u2")])
PHOTON_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("t", "<f4"),
                         ("theta", "<f4"), ("phi", "<f4"), ("wavelength", "<f4"), ("cherenkovDist", "<f4"),
                         ("numScatters", "<u4"), ("weight", "<f4"), ("id", "<u4"),
                         ("stringID", "<i2"), ("omID", "<u2"),
                         ("sx", "<f4"), ("sy", "<f4"), ("sz", "<f4"), ("st", "<f4"),
                         ("stheta", "<f4"), ("sphi", "<f4"), ("groupVelocity", "<f4"), ("distInAbsLens", "<f4")])

DOM_RADIUS = 0.16510          # python/traysegments: I3CLSimMakePhotons DOMRadius default
OVERSIZE = 5.0                # cfg.txt line 1 / DOMOversizeFactor


def single_string_geometry():
    """C1: one string of 60 DOMs at x=y=20 m, z = +500 ... -503 m in 17 m steps."""
    n = 60
    return dict(string_ids=np.full(n, 1, dtype=np.int32), dom_ids=np.arange(1, n + 1, dtype=np.uint32),
                x=np.full(n, 20.0), y=np.full(n, 20.0), z=500.0 - 17.0 * np.arange(n),
                subdetectors=["IceCube"] * n, om_radius=DOM_RADIUS * OVERSIZE)


def ic86_geometry(seed=86, jitter=0.3):
    """C2-C5: synthetic 86-string detector: 78 strings on a 125 m triangular
    grid with 60 DOMs at 17 m from z=+500 m, plus 8 DeepCore-like strings on a
    72 m ring (10 DOMs at 10 m from z=+190 m, 50 DOMs at 7 m from z=-160 m) as a
    second subdetector; Gaussian x/y jitter per DOM exercises the int16
    template path of the geometry builder."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pts = []
    for i in range(-8, 9):
        for j in range(-8, 9):
            x = (i + 0.5 * j) * 125.0
            y = j * 125.0 * math.sqrt(3.0) / 2.0
            pts.append((round(math.hypot(x, y), 6), round(math.atan2(y, x), 9), x, y))
    pts.sort()
    sid, did, xs, ys, zs, sub = [], [], [], [], [], []
    for s, (_, _, x, y) in enumerate(pts[:78]):
        for d in range(60):
            sid.append(s + 1); did.append(d + 1); sub.append("IceCube")
            xs.append(x + jitter * rng.standard_normal()); ys.append(y + jitter * rng.standard_normal())
            zs.append(500.0 - 17.0 * d)
    for k in range(8):
        ang = math.radians(22.5 + 45.0 * k)
        x0, y0 = 72.0 * math.cos(ang), 72.0 * math.sin(ang)
        zz = [190.0 - 10.0 * d for d in range(10)] + [-160.0 - 7.0 * d for d in range(50)]
        for d in range(60):
            sid.append(79 + k); did.append(d + 1); sub.append("DeepCore")
            xs.append(x0 + jitter * rng.standard_normal()); ys.append(y0 + jitter * rng.standard_normal())
            zs.append(zz[d])
    return dict(string_ids=np.array(sid, dtype=np.int32), dom_ids=np.array(did, dtype=np.uint32),
                x=np.array(xs), y=np.array(ys), z=np.array(zs), subdetectors=sub,
                om_radius=DOM_RADIUS * OVERSIZE)


def large_detector_geometry(side=24, spacing=125.0, doms=60, seed=5, jitter=0.3):
    """side x side strings on a triangular grid (576 strings, 34 560 DOMs by default): a detector whose string records
    alone exceed the LDS budget of seven workgroups per CU -- the kernels then run with fewer workgroups per CU."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sid, did, xs, ys, zs, sub = [], [], [], [], [], []
    s = 0
    for i in range(side):
        for j in range(side):
            s += 1
            x0 = (i - side / 2 + 0.5 * (j % 2)) * spacing
            y0 = (j - side / 2) * spacing * math.sqrt(3.0) / 2.0
            for d in range(doms):
                sid.append(s); did.append(d + 1); sub.append("IceCube")
                xs.append(x0 + jitter * rng.standard_normal()); ys.append(y0 + jitter * rng.standard_normal())
                zs.append(500.0 - 17.0 * d)
    return dict(string_ids=np.array(sid, dtype=np.int32), dom_ids=np.array(did, dtype=np.uint32),
                x=np.array(xs), y=np.array(ys), z=np.array(zs), subdetectors=sub, om_radius=DOM_RADIUS * OVERSIZE)


def cascade_steps(n, seed=1, photons_per_step=200, radius=500.0, half_height=500.0, vertex=None, pad_to=1):
    """Cascade-like steps: 1 mm long, beta 1, weight 1, isotropic directions;
    vertices uniform in a cylinder (or a fixed vertex).  Padded with
    numPhotons=0 steps to a multiple of `pad_to` (granularity padding,
    I3CLSimLightSourceToStepConverterAsync.cxx:210-273)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    total = ((n + pad_to - 1) // pad_to) * pad_to
    st = np.zeros(total, dtype=STEP_DTYPE)
    if vertex is None:
        r = radius * np.sqrt(rng.random(n))
        ph = 2.0 * np.pi * rng.random(n)
        st["x"][:n] = r * np.cos(ph); st["y"][:n] = r * np.sin(ph)
        st["z"][:n] = half_height * (2.0 * rng.random(n) - 1.0)
    else:
        st["x"][:n], st["y"][:n], st["z"][:n] = vertex
    st["t"][:n] = 0.0
    st["theta"][:n] = np.arccos(1.0 - 2.0 * rng.random(n))
    st["phi"][:n] = 2.0 * np.pi * rng.random(n)
    st["length"][:n] = 0.001
    st["beta"][:n] = 1.0
    st["num"][:n] = photons_per_step
    st["weight"][:n] = 1.0
    st["id"][:n] = np.arange(n, dtype=np.uint32)
    st["sourceType"][:n] = 0
    # padding steps keep harmless kinematics (beta=1 avoids 1/0)
    st["beta"][n:] = 1.0
    return st


def flasher_steps(n, seed=5, photons_per_step=400, position=(0.0, 0.0, 0.0), source_type=1, pad_to=1):
    """C5: point source; sourceType>=1 picks wavelength generator `source_type`
    and keeps the step direction (propagation_kernel.c.cl:174-182)."""
    st = cascade_steps(n, seed=seed, photons_per_step=photons_per_step, vertex=position, pad_to=pad_to)
    st["length"][:n] = 0.0
    st["sourceType"][:n] = source_type
    return st
