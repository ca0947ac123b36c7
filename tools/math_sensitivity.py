#!/usr/bin/env python3
"""What the self-defined math library costs in physics terms (CPU; ANALYSIS TOOL, imports oracle/).

The parity bar of this repository is "bit-identical to the oracle", whose log/exp/sin/cos/powr are this repository's own
definitions (oracle/oracle_math.h).  The reference kernel runs on an OpenCL runtime's builtins (a few ulp, unpinned) with
-cl-mad-enable, which cannot be run here.  This tool measures how much a DIFFERENT conforming math library changes the
results: the same restatement built (a) with the deterministic header, (b) with glibc's logf/expf/sinf/cosf/powf/
atan2f/acosf, (c) like (b) with fused multiply-adds allowed, on the C2 miniature (SPICE-Mie, 86 strings) x many seeds.

A photon random walk is chaotic in the last bits: one photon that scatters once more consumes two more random numbers
and every later photon of the same step gets different draws.  So the per-photon agreement is small by construction;
what has to agree are the OBSERVABLES: hit counts per DOM, arrival times, scatter counts."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import builders as B, capi
from clsim_amd import synthetic as S


def ks_2samp(a, b):
    a, b = np.sort(a), np.sort(b)
    allv = np.concatenate([a, b])
    d = np.max(np.abs(np.searchsorted(a, allv, side="right") / len(a) - np.searchsorted(b, allv, side="right") / len(b)))
    en = np.sqrt(len(a) * len(b) / (len(a) + len(b)))
    lam = (en + 0.12 + 0.11 / en) * d
    j = np.arange(1, 101)
    p = float(np.clip(2 * np.sum((-1) ** (j - 1) * np.exp(-2 * (lam * j) ** 2)), 0, 1))
    return float(d), p


def header_sha16():
    """sha256 (first 16 hex digits) of the math definition both sides share: oracle_math.h + math_tables.h.  The stored table
    (profiles/r06/math_sensitivity.json) carries it; tests/test_math_sensitivity.py fails when the header has moved on."""
    import hashlib
    h = hashlib.sha256()
    for name in ("oracle_math.h", "math_tables.h"):
        with open(os.path.join(ROOT, "oracle", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def run(seeds=16, n=4096, quiet=False):
    g = S.ic86_geometry()
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    med = B.load_ppc_ice(os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
    bias = B.icecube_dom_acceptance()
    gens = [B.cherenkov_wlen_generator(bias, med)]
    a = B.mwc_multipliers(n)
    runs = {}
    try:                                # (whatever happens, the checker's own library is the one left loaded)
        for variant in (None, "liboracle_libm.so", "liboracle_libm_mad.so"):
            capi.use_variant(variant)
            T = capi.make_tables(med, geo, gens, bias, pancake=5.0)
            hits, states, trips = [], [], 0
            for s in range(seeds):
                steps = S.cascade_steps(n, seed=5000 + s, photons_per_step=200)
                x = B.seed_streams(a, 777 + s)
                ph, cnt, x_after, it = capi.propagate(T, steps, x, a, threads=os.cpu_count())
                ph = ph.copy(); ph["id"] += np.uint32(s * n)          # step identity across seeds
                hits.append(ph); states.append(x_after); trips += it
            runs[variant or "deterministic"] = (np.concatenate(hits), np.concatenate(states), trips)
            if not quiet:
                print("%s: %d hits, %d loop trips" % (variant or "deterministic", len(runs[variant or "deterministic"][0]), trips), file=sys.stderr, flush=True)
    finally:
        capi.use_variant(None)
    photons = seeds * n * 200
    base_h, base_x, base_t = runs["deterministic"]
    out = {"oracle_math_sha16": header_sha16(),
           "what": "the deterministic math library (oracle/oracle_math.h == clsim_amd/csrc/detmath.hip.h) against glibc's libm, without and "
                   "with fused multiply-adds (the reference builds its kernel with -cl-mad-enable, OpenCL.cxx:628): the bridge between "
                   "'bit-exact to our oracle' and north_star's 'within 1e-5 relative of the reference OpenCL kernel'",
           "photons": photons, "steps": seeds * n, "workload": "C2 miniature: %d seeds x %d steps x 200 photons, SPICE-Mie, 86 strings" % (seeds, n),
           "deterministic": {"hits": int(len(base_h)), "loop_trips": int(base_t)}, "variants": {}}
    n_doms = 86 * 60

    def dom_index(h):
        return (h["stringID"].astype(np.int64)) * 64 + h["omID"].astype(np.int64)

    for name in ("liboracle_libm.so", "liboracle_libm_mad.so"):
        h, x, t = runs[name]
        same_stream = float(np.mean(x == base_x))
        # identical hit records (all 80 bytes), as multisets
        ua = np.unique(capi.sort_photons(base_h).view(np.dtype((np.void, 80))), return_counts=True)
        ub = np.unique(capi.sort_photons(h).view(np.dtype((np.void, 80))), return_counts=True)
        common = np.intersect1d(ua[0], ub[0], assume_unique=True)
        # per-DOM hit counts: chi^2 of two Poisson samples, sum (a-b)^2/(a+b) over DOMs with a+b>0
        ca = np.bincount(dom_index(base_h), minlength=128 * 64); cb = np.bincount(dom_index(h), minlength=128 * 64)
        m = (ca + cb) > 0
        chi2 = float(np.sum((ca[m] - cb[m]) ** 2 / (ca[m] + cb[m])))
        ndf = int(m.sum())
        ks_t = ks_2samp(base_h["t"] - base_h["st"], h["t"] - h["st"])
        ks_s = ks_2samp(base_h["numScatters"].astype(float), h["numScatters"].astype(float))
        ks_w = ks_2samp(base_h["wavelength"], h["wavelength"])
        # the north star's bar: the same photons detected by the same DOMs, floats within 1e-5 relative.  A hit is identified by
        # (step, DOM, number of scatters) and, among the few hits that share those, by its rank in wavelength; a pair counts as
        # "found again" when its wavelengths agree to 1e-5 relative.  (Until round 5 the key held the wavelength's BITS: the
        # wavelength is drawn before any transcendental of the walk and is the same bits under another libm -- but not under
        # contraction, which rounds the spectrum's interpolation differently in the last place and made 15 % of the hits look lost.)
        def keyed(hh):
            k = np.zeros(len(hh), dtype=[("id", "u4"), ("dom", "i8"), ("ns", "u4"), ("rank", "u4")])
            k["id"], k["dom"], k["ns"] = hh["id"], dom_index(hh), hh["numScatters"]
            o = np.lexsort((hh["wavelength"], k["ns"], k["dom"], k["id"]))
            k, hh = k[o], hh[o]
            same = np.zeros(len(k), dtype=bool)
            same[1:] = (k["id"][1:] == k["id"][:-1]) & (k["dom"][1:] == k["dom"][:-1]) & (k["ns"][1:] == k["ns"][:-1])
            start = np.maximum.accumulate(np.where(~same, np.arange(len(k)), 0))
            k["rank"] = np.arange(len(k)) - start
            return k, hh
        ka, ha = keyed(base_h); kb, hb = keyed(h)
        _, ia, ib = np.intersect1d(ka.view(np.dtype((np.void, ka.dtype.itemsize))), kb.view(np.dtype((np.void, kb.dtype.itemsize))), return_indices=True)
        close = np.abs(ha[ia]["wavelength"].astype(np.float64) - hb[ib]["wavelength"].astype(np.float64)) <= 1e-5 * ha[ia]["wavelength"].astype(np.float64)
        ia, ib = ia[close], ib[close]
        ma, mb = ha[ia], hb[ib]
        def rel(f, scale=None):
            a_, b_ = ma[f].astype(np.float64), mb[f].astype(np.float64)
            den = np.maximum(np.abs(a_), 1e-30) if scale is None else scale
            return float(np.max(np.abs(a_ - b_) / den)) if len(a_) else 0.0
        def quant(f):
            a_, b_ = ma[f].astype(np.float64), mb[f].astype(np.float64)
            r = np.abs(a_ - b_) / np.maximum(np.abs(a_), 1e-30)
            return {"identical": float(np.mean(r == 0)), "median": float(np.median(r)), "p99": float(np.quantile(r, 0.99)), "within_1e-5": float(np.mean(r <= 1e-5))}
        matched = {"same_step_dom_scatters_wavelength": float(len(ia)) / len(base_h),
                   "wavelength_bits_identical": float(np.mean(ma["wavelength"].view(np.uint32) == mb["wavelength"].view(np.uint32))) if len(ia) else 1.0,
                   "rel_diff_time": quant("t"), "rel_diff_cherenkov_dist": quant("cherenkovDist"),
                   "max_rel_diff_time": rel("t"), "max_rel_diff_wavelength": rel("wavelength"), "max_rel_diff_weight": rel("weight"),
                   "max_rel_diff_dist_in_abs_lens": rel("distInAbsLens"), "max_rel_diff_cherenkov_dist": rel("cherenkovDist"),
                   # hit position is relative to the DOM centre (|r| = 0.16510 m * oversize): differences relative to that radius
                   "max_diff_position_over_radius": max(rel("x", 0.8255), rel("y", 0.8255), rel("z", 0.8255)),
                   "max_abs_diff_theta_phi_rad": max(rel("theta", 1.0), rel("phi", 1.0))}
        out["variants"][name] = {
            "matched_hits": matched,
            "hits": int(len(h)), "hits_rel_diff": (len(h) - len(base_h)) / len(base_h), "hits_poisson_sigma": (len(h) - len(base_h)) / np.sqrt(len(h) + len(base_h)),
            "loop_trips_rel_diff": (t - base_t) / base_t,
            "steps_with_identical_final_stream_state": same_stream,
            "bit_identical_hit_records_fraction": float(len(common)) / len(base_h),
            "per_dom_chi2": chi2, "per_dom_ndf": ndf, "chi2_per_ndf": chi2 / ndf,
            "ks_delay_time": {"D": ks_t[0], "p": ks_t[1]}, "ks_num_scatters": {"D": ks_s[0], "p": ks_s[1]}, "ks_wavelength": {"D": ks_w[0], "p": ks_w[1]},
            "mean_delay_ns": [float(np.mean(base_h["t"] - base_h["st"])), float(np.mean(h["t"] - h["st"]))],
            "mean_num_scatters": [float(base_h["numScatters"].mean()), float(h["numScatters"].mean())]}
    return out


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    print(json.dumps(run(seeds, n), indent=1))


if __name__ == "__main__":
    main()
