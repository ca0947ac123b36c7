#!/usr/bin/env python3
"""Wave scheduling model of the v3 kernel (ANALYSIS TOOL, uses the oracle built
with -DORACLE_TRACE).  Replays per-step iteration traces through the kernel's
wave-level schedule (step queue, deferred creation) and reports, per code
section, the VALU cost a wave pays versus what its lanes needed."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import common  # noqa: E402

# rough VALU instruction weights per section (from the static ISA attribution)
W = dict(create=480, base=330, layer_trip=55, subdet=70, cell=14, string=75, dom=95, finish=60, scatter=230, liu=85, hg=40)


def traces(ice, n):
    so = "/tmp/liboracle_trace.so"
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-mfma", "-mavx2", "-fopenmp", "-w",
                           "-DORACLE_TRACE", "-shared", "-o", so, os.path.join(ROOT, "oracle", "clsim_oracle.c"), "-lm"])
    L = C.CDLL(so)
    L.oracle_trace_step.restype = C.c_uint64
    cfg = common.config(ice)
    T = common.oracle_tables(cfg)
    steps = common.steps_for(cfg, n, seed=21)
    x, a = common.streams(n)
    out = []
    cap = 40000
    for i in range(n):
        buf = np.zeros((cap, 8), dtype=np.uint8)
        st = steps[i:i + 1].copy()
        k = L.oracle_trace_step(C.byref(T.t), st.ctypes.data_as(C.c_void_p), C.c_uint64(int(x[i])), C.c_uint32(int(a[i])),
                                buf.ctypes.data_as(C.c_void_p), C.c_uint64(cap))
        out.append(buf[:k].copy())
    return out


def simulate(tr, k_new, steps_per_lane):
    """one wave, 64 lanes, queue of len(tr) steps"""
    nxt = 0
    cur = [None] * 64          # (trace, pos)
    need = [True] * 64
    alive = [True] * 64
    paid = dict.fromkeys(W, 0.0)
    used = dict.fromkeys(W, 0.0)
    trips = 0
    limit = 64 * steps_per_lane
    while True:
        n_need = sum(1 for l in range(64) if alive[l] and need[l])
        n_ready = sum(1 for l in range(64) if alive[l] and not need[l])
        if n_need == 0 and n_ready == 0:
            break
        trips += 1
        if n_ready == 0 or n_need >= k_new:
            created = 0
            for l in range(64):
                if alive[l] and need[l]:
                    if cur[l] is None or cur[l][1] >= len(cur[l][0]):
                        if nxt < limit and nxt < len(tr):
                            cur[l] = [tr[nxt], 0]
                            nxt += 1
                        else:
                            alive[l] = False
                            continue
                    need[l] = False
                    created += 1
            if created:
                paid["create"] += W["create"]
                used["create"] += W["create"] * created / 64.0
        run = [l for l in range(64) if alive[l] and not need[l]]
        if not run:
            continue
        recs = np.array([cur[l][0][cur[l][1]] for l in run], dtype=np.int32)
        for l in run:
            cur[l][1] += 1
        m = len(run)
        def sec(name, counts):
            paid[name] += W[name] * counts.max()
            used[name] += W[name] * counts.sum() / 64.0
        ones = np.ones(m, dtype=np.int32)
        sec("base", ones)
        sec("layer_trip", recs[:, 1])
        sec("subdet", 2 * ones)
        sec("cell", recs[:, 2])
        sec("string", recs[:, 3])
        sec("dom", recs[:, 4])
        sec("finish", ones)
        sec("scatter", recs[:, 6])
        sec("liu", recs[:, 5])
        sec("hg", recs[:, 6] - recs[:, 5])
        # lanes whose photon ended need a new one (next record has create flag or trace ended)
        for l in run:
            t, p = cur[l]
            if p >= len(t) or t[p][0]:
                need[l] = True
    return trips, paid, used


if __name__ == "__main__":
    ice = sys.argv[1] if len(sys.argv) > 1 else "mie"
    spl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tr = traces(ice, 64 * spl)
    for k in (1, 4, 6, 8, 12):
        trips, paid, used = simulate(tr, k, spl)
        tp, tu = sum(paid.values()), sum(used.values())
        print("k_new=%2d trips=%6d paid=%.3g used=%.3g util=%.1f%%" % (k, trips, tp, tu, 100 * tu / tp))
        if k == 6:
            for s in W:
                print("   %-10s paid %5.1f%%  util %5.1f%%" % (s, 100 * paid[s] / tp, 100 * used[s] / max(paid[s], 1e-9)))
