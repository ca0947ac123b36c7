#!/usr/bin/env python3
"""Lockstep (SIMT) divergence model of the propagation loop.

TEST/ANALYSIS INFRASTRUCTURE (uses the oracle built with -DORACLE_TRACE):
traces, per work item and loop iteration, which phases ran and the trip counts
of the inner loops, groups 64 consecutive steps into a wave and reports what a
wave-uniform loop pays (max / any over lanes) against what the lanes needed.
Used to decide how to restructure the HIP kernel; not part of the product.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402
from tests import common  # noqa: E402


def main(ice="mie", waves=4):
    so = "/tmp/liboracle_trace.so"
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-mfma", "-mavx2", "-fopenmp",
                           "-DORACLE_TRACE", "-shared", "-o", so, os.path.join(ROOT, "oracle", "clsim_oracle.c"), "-lm"])
    L = C.CDLL(so)
    L.oracle_trace_step.restype = C.c_uint64
    cfg = common.config(ice)
    T = common.oracle_tables(cfg)
    n = 64 * waves
    steps = common.steps_for(cfg, n, seed=21)
    x, a = common.streams(n)
    cap = 40000
    traces = []
    for i in range(n):
        buf = np.zeros((cap, 8), dtype=np.uint8)
        st = steps[i:i + 1].copy()
        k = L.oracle_trace_step(C.byref(T.t), st.ctypes.data_as(C.c_void_p), C.c_uint64(int(x[i])), C.c_uint32(int(a[i])),
                                buf.ctypes.data_as(C.c_void_p), C.c_uint64(cap))
        traces.append(buf[:k])
    names = ["create", "layer_trips", "cells", "strings", "dom_layers", "liu", "scatter", "hit"]
    tot_lane = np.zeros(8)
    tot_wave_max = np.zeros(8)
    tot_wave_any = np.zeros(8)
    lane_iters = 0
    wave_iters = 0
    active_sum = 0
    for w in range(waves):
        tr = traces[64 * w:64 * (w + 1)]
        L_ = max(len(t) for t in tr)
        M = np.zeros((64, L_, 8), dtype=np.int32)
        act = np.zeros((64, L_), dtype=bool)
        for l, t in enumerate(tr):
            M[l, :len(t)] = t
            act[l, :len(t)] = True
        lane_iters += act.sum()
        wave_iters += L_
        active_sum += act.sum()
        tot_lane += M.sum(axis=(0, 1))
        tot_wave_max += M.max(axis=0).sum(axis=0)
        tot_wave_any += (M > 0).any(axis=0).sum(axis=0)
    print("ice=%s waves=%d lane-iterations=%d wave-iterations=%d mean active lanes=%.1f" %
          (ice, waves, lane_iters, wave_iters, active_sum / wave_iters))
    print("%-12s %12s %14s %14s %10s" % ("phase", "per lane-it", "wave max/it", "wave any/it", "util"))
    for k, nm in enumerate(names):
        per_lane = tot_lane[k] / lane_iters
        wmax = tot_wave_max[k] / wave_iters
        wany = tot_wave_any[k] / wave_iters
        util = (tot_lane[k] / 64.0) / max(tot_wave_max[k], 1e-9)
        print("%-12s %12.4f %14.4f %14.4f %9.1f%%" % (nm, per_lane, wmax, wany, 100 * util))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "mie", int(sys.argv[2]) if len(sys.argv) > 2 else 4)
