#!/usr/bin/env python3
"""Stress test (GPU): the propagation result must not depend on the schedule.  Random bunch sizes, photon counts, grids,
slice counts and batching thresholds; every case is compared with the same bunch run as whole steps on a small grid
(hit multiset and final RNG states, bit for bit).  A hang shows up as the caller's timeout.  usage: stress_schedules.py [cases] [keep|stop] [seed]
(keep: the instantiations without STOP_PHOTONS_ON_DETECTION, classic and pooled kernel; `clear` ice among the configurations)"""
import devlib  # noqa: F401  (the developer build of the library: this tool steers it through the environment)
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from clsim_amd import converter as CV, synthetic as S
from tests import common

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
KEEP = len(sys.argv) > 2 and sys.argv[2] == "keep"
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 2024
rng = np.random.Generator(np.random.PCG64(SEED))
dev = torch.device("cuda", 0)
cap = 1 << 21
out = torch.empty((cap, 80), dtype=torch.uint8, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)


def run(cfg, steps, env):
    for k in ("CLSIMHIP_GRID", "CLSIMHIP_SLICES", "CLSIMHIP_K_NEW", "CLSIMHIP_K_SEARCH", "CLSIMHIP_KERNEL", "CLSIMHIP_POOL_R", "CLSIMHIP_K_POP",
              "CLSIMHIP_NO_FAST"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    n = len(steps)
    conv = common.product_converter(cfg, n, stop_detected=not KEEP)
    d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48).copy()).to(dev)
    conv.PropagateDevice(d_steps.data_ptr(), n, out.data_ptr(), cap, cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = int(cnt.cpu().item())
    assert c <= cap
    ph = np.frombuffer(out[:c].cpu().numpy().tobytes(), dtype=S.PHOTON_DTYPE)
    return common.sort_photons(ph).tobytes(), conv.GetRNGState(n).tobytes(), c


for case in range(cases):
    # (seeds other than the first: a medium without a group index override -- generic kernels only -- among the configurations)
    name = (["mie", "lea", "clear", "flasher"] if KEEP else ["mie", "lea", "c1", "flasher"] if SEED == 2024 else ["mie", "lea_dispersion", "c1", "flasher"])[case % 4]
    cfg = common.config(name)
    n = 256 * int(rng.integers(1, [40, 400, 1200][case % 3]))
    steps = common.steps_for(cfg, n, seed=100 + case + (SEED - 2024) * 1000)
    mode = case % 5
    if mode == 0:
        steps["num"] = rng.choice([0, 1, 2, 7, 63, 64, 65, 200, 399, 1500], size=n).astype(np.uint32)
    elif mode == 1:
        steps["num"] = rng.integers(0, 60, n).astype(np.uint32)
    elif mode == 2:
        steps["num"] = 0
        steps["num"][rng.integers(0, n, 5)] = 3000
    env = dict(CLSIMHIP_GRID=int(rng.integers(1, 1793)), CLSIMHIP_SLICES=int(rng.choice([1, 2, 3, 5, 16, 33, 64])),
               CLSIMHIP_K_NEW=int(rng.choice([1, 4, 12, 40, 64])), CLSIMHIP_K_SEARCH=int(rng.choice([1, 3, 5, 20])))
    if case % 2:        # the pooled kernel: ring size, service threshold, specialised or generic instantiation
        env.update(CLSIMHIP_KERNEL="pool", CLSIMHIP_POOL_R=int(rng.choice([4, 7, 16, 34])), CLSIMHIP_K_POP=int(rng.choice([1, 4, 17, 64])),
                   CLSIMHIP_NO_FAST=int(rng.integers(0, 2)), CLSIMHIP_GRID=int(rng.integers(1, 513)))
    ref = run(cfg, steps, dict(CLSIMHIP_GRID=64, CLSIMHIP_SLICES=1, CLSIMHIP_K_NEW=1, CLSIMHIP_K_SEARCH=1))
    got = run(cfg, steps, env)
    assert got[2] == ref[2] and got[0] == ref[0] and got[1] == ref[1], (case, name, n, env)
    print("case %2d %-8s n %6d photons %9d hits %7d %s ok" % (case, name, n, int(steps["num"].sum()), got[2], env), flush=True)
print("stress ok: %d cases" % cases)
