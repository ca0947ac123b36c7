#!/usr/bin/env python3
"""Census of the pooled kernel (GPU).  ANALYSIS TOOL.  Needs the analysis build:
    make -C clsim_amd/csrc clean && make -C clsim_amd/csrc -j EXTRA=-DCLSIMHIP_CENSUS
usage: exp_pool_census.py spec ...   (spec as in exp_pool_scan.py)"""
import devlib  # noqa: F401  (the developer build of the library: this tool steers it through the environment)
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clsim_amd import converter as CV, synthetic as S, _lib

ENV = {"kernel": "CLSIMHIP_KERNEL", "R": "CLSIMHIP_POOL_R", "pop": "CLSIMHIP_K_POP", "new": "CLSIMHIP_K_NEW", "slices": "CLSIMHIP_SLICES",
       "search": "CLSIMHIP_K_SEARCH", "grid": "CLSIMHIP_GRID"}
bias = CV.GetIceCubeDOMAcceptance()
g86 = S.ic86_geometry()
geom = CV.I3CLSimSimpleGeometry.from_dict(g86)
dev = torch.device("cuda", 0)
for spec in sys.argv[1:]:
    kv = dict(item.split("=") for item in spec.split(",") if item)
    for k, e in ENV.items():
        os.environ.pop(e, None)
        if k in kv:
            os.environ[e] = kv[k]
    os.environ.setdefault("CLSIMHIP_KERNEL", "pool")
    n = int(kv.get("n", 1 << 20))
    workload = kv.get("workload", "c2")           # bench.py's workloads: c2 (SPICE-Mie), c3 (SPICE-Lea), c5 (flasher steps at a DOM); origin: a c3 bunch with every vertex at the origin (the reference's benchmark.py: a cascade on the central string)
    medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie" if workload == "c2" else "spice_lea"))
    gens = [CV.makeCherenkovWavelengthGenerator(bias, medium)]
    if workload == "c5":
        gens.append(CV.I3CLSimRandomValueConstant(405e-9))
        k = int(np.argmin(np.abs(g86["x"]) + np.abs(g86["y"]) + np.abs(g86["z"] + 100.0)))
        steps = S.flasher_steps(n, seed=1000, photons_per_step=400, position=(float(g86["x"][k]), float(g86["y"][k]), float(g86["z"][k])))
    elif workload == "origin":
        steps = S.cascade_steps(n, seed=1000, vertex=(0.0, 0.0, 0.0))
    else:
        steps = S.cascade_steps(n, seed=1000)
    conv = CV.initializeHIP(0, geom, medium, bias, gens, pancakeFactor=5.0, approximateNumberOfWorkItems=n, seed=12345)
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    cap = (48 if workload == "c5" else 8) << 20
    out = torch.empty((cap, 80), dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    for rep in range(2):
        conv.KernelTimeMs(reset=True)
        conv.PropagateDevice(d_steps.data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ms, _ = conv.KernelTimeMs(reset=True)
    buf = np.zeros(1 << 19, dtype=np.uint64)
    lib = _lib.load(); lib.clsimhip_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
    lib.clsimhip_debug_counters(conv._h, buf.ctypes.data_as(C.c_void_p))
    trips, run, services, creations, created, vacant, polls, parked = (float(v) for v in buf[:8])
    searches, chunks, empty_ring = (float(v) for v in buf[9:12])
    t0 = int(buf[8])
    rec = buf[16:16 + 3 * 8192].reshape(-1, 3).astype(np.int64)
    rec = rec[rec[:, 0] > 0]
    end = (rec[:, 0] - t0) / 100e3
    q = lambda a, p: float(np.percentile(a, p))
    lanes = 64.0 * trips
    photons = float(steps["num"].sum())
    print("%s: %.1f ms, %d waves | lane trips with a live photon per photon %.2f, wave trips per 64 photons %.2f | lanes: run %.1f%% parked %.1f%% without photon %.1f%% | per trip: services %.3f, creation stages %.4f "
          "(%.1f photons each, %.2f chunks), searches %.3f, ring empty %.1f%% of trips, polls/trip %.3f | wave end p1 %.1f p50 %.1f p99 %.1f max %.1f ms | trips/wave p10 %d p50 %d p90 %d"
          % (spec, ms, len(rec), (run + parked) / photons, trips * 64.0 / photons, 100 * run / lanes, 100 * parked / lanes, 100 * vacant / lanes, services / trips, creations / trips,
             created / max(creations, 1), chunks / max(creations, 1), searches / trips, 100 * empty_ring / trips, polls / trips,
             q(end, 1), q(end, 50), q(end, 99), end.max(), q(rec[:, 2], 10), q(rec[:, 2], 50), q(rec[:, 2], 90)), flush=True)
    t_service, t_publish, t_take, t_create = (float(v) for v in buf[12:16])
    t_total = float(buf[24600])
    if t_total > 0:
        print("   wave time (shader clock, summed over waves) inside the service block %.1f%% -- of it: publishing finished units %.1f%%, taking new units (queue atomic + "
              "first look at the work record) %.1f%%, creation chunks (polls, record reads, creation, compaction) %.1f%%, the rest (retire, hand-out) %.1f%%"
              % (100 * t_service / t_total, 100 * t_publish / t_total, 100 * t_take / t_total, 100 * t_create / t_total,
                 100 * (t_service - t_publish - t_take - t_create) / t_total), flush=True)
        t_store, n_store, t_hand, n_hand = (float(v) for v in buf[24601:24605])
        per_trip = t_total / trips
        print("   one ring hand-over, shader clock: store of a creation chunk's photons (five 16-byte words per lane) %.0f cycles x %d chunks; hand-out block "
              "(ballot, rank, five 16-byte loads, unpack, wave barrier) %.0f cycles x %d blocks; a wave trip takes %.0f cycles of wave time"
              % (t_store / max(n_store, 1), int(n_store), t_hand / max(n_hand, 1), int(n_hand), per_trip), flush=True)
    # divergent regions (prop_device.hip.h: CENSUS_REGION): visits per wave trip and active lanes per visit
    names = ["layer crossing body", "search filter levels 2-3", "Liu branch", "HG branch", "full DOM search", "named DOM search", "photon creation",
             "service (free lanes)", "scattering (all)", "layer walk (all)", "aimed at the string?"]
    reg = buf[32768:32768 + 32 * 8192].reshape(-1, 16, 2).astype(np.float64).sum(axis=0)
    print("   region                      visits/trip   lanes/visit   lane-visits/trip")
    for k, nm in enumerate(names):
        v, l = reg[k]
        print("   %-26s %12.4f %13.2f %18.3f" % (nm, v / trips, l / max(v, 1.0), l / trips), flush=True)
    del conv
