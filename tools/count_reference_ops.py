#!/usr/bin/env python3
"""The useful-work side of the roofline ("a counted flop/transcendental figure per photon from the
instrumented CPU restatement"; the metric: resources/scripts/benchmark.py:326-340).

Runs a sample of a bench.py workload (c2 | c3 | c5) through the COUNTING build of the oracle (oracle/count_ops.hpp: the
restatement of the reference kernel with float operators that count themselves, results bit-identical to the checker's) and
prices what was counted in vector instructions of gfx950, two ways:

  as_written   every operation the reference's expressions ask for, one by one, at the cost of the device's GENERIC sequence for
               it (IEEE divide 11, square root 17, ..., the math library by name; profiles/r05/math_unit_costs.json) -- including the
               DOM search's cell arithmetic, which the reference runs on every loop trip;
  transformed  the same photon histories with the bit-preserving transformations of DESIGN.md section 2 applied and every
               division / root at the cheapest form PROVEN exact for its site (price_transformed below): wavelength-only medium
               factors once per photon instead of once per layer visit, host-folded layer constants, x*1 and x/1 dropped without
               anisotropy, and NO search arithmetic at all (the filter's job is to prove it away; what the filter and the
               remaining searches cost is the kernel's overhead, not the reference's arithmetic).  This is the floor the
               kernel's issued lane operations are compared with (roofline.valu.overhead_ratio).

usage: tools/count_reference_ops.py [c2 c3 c5 ...] [--steps N] [--out profiles/r06/reference_ops.json]"""
import argparse, hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from clsim_amd import synthetic as S
from oracle import builders as B
from oracle import capi

GENERIC = {"add": 1, "mul": 1, "cmp": 1, "cvt": 1, "neg": 0, "fabs": 0, "floor_trunc": 1}      # (negation and |x| are source modifiers)
NAMED = {"div": "div", "sqrt": "sqrt", "rsqrt": "rsqrt", "log": "log", "exp": "exp", "powr": "powr", "powr_unit": "powr_unit", "sincos": "sincos",
         "sin": "sin", "cos": "sin", "acos": "acos", "atan2": "atan2", "rng_draw": "rng_draw"}


def unit_costs(cycles=False):
    """vector instructions per unit; cycles=True: full-rate issue slots (a quarter-rate instruction -- v_rcp_f32, v_sqrt_f32, the 32-bit integer
    multiplies -- holds the SIMD four times as long as an fma)"""
    with open(os.path.join(ROOT, "profiles", "r05", "math_unit_costs.json")) as f:
        u = json.load(f)["units"]
    if cycles:
        u = {k: dict(v, valu=v["valu"] + 3 * v["of_them_quarter_rate"]) for k, v in u.items()}
    c = dict(GENERIC)
    for op, name in NAMED.items():
        c[op] = u[name]["valu"]
    c["rng_draw"] -= 2                       # (the measuring kernel folds the stream state into its output: one convert, one add)
    c["other_math"] = u["powr"]["valu"]
    proven = {k: u[k + "(range-restricted)"]["valu"] for k in ("rcp", "div_near", "sqrt_near", "rsqrt_near")}
    proven["div_inv"] = u["div_by_invariant(proven)"]["valu"]
    proven["rcp_of_rcp"] = u["rcp_of_rcp(seeded)"]["valu"]
    proven["div_with"] = u["div_near_with(reciprocal at hand)"]["valu"]
    proven["rsqrt_unit"] = u["rsqrt_unit(next to one)"]["valu"]
    return c, proven


def price_as_written(ops, cost):
    by_region = {}
    for region, d in ops.items():
        if region == "rng_internal":         # the conversion and scaling inside a draw: part of the unit `rng_draw`
            continue
        by_region[region] = sum(n * cost[o] for o, n in d.items())
    return by_region


def price_transformed(ops, ev, cost, pv, aniso, tilt):
    """Per region: the simple operations as counted, the divisions / roots / named functions re-priced per SITE.  Every rule names the
    kernel code that holds the proof (clsim_amd/csrc/prop_device.hip.h)."""
    simple = lambda d, skip=(): sum(n * cost[o] for o, n in d.items() if o in GENERIC and o not in skip)
    named = lambda d, names: sum(d.get(o, 0) * cost[o] for o in names)
    g = lambda r: ops.get(r, {})
    P, T, SC, CR, CT, EV = ev["photons"], ev["trips"], ev["scatters"], ev["layer_crossings"], ev["crossing_trips"], ev["layer_length_evals"]
    out = {}
    # creation: generic sequences throughout (photon_birth, generate_wavelength, group_velocity use the IEEE divide); the rotation onto
    # the Cherenkov cone is scatter_direction: two quotients over one sine (div_near 8 + 5 with the shared reciprocal), sqrt_near x2, rsqrt_near
    c = g("create")
    cone = c.get("sincos", 0)                  # photons that are rotated onto a Cherenkov cone (flasher photons are not)
    out["create"] = simple(c) + named(c, ("log", "rng_draw", "sincos")) + (c.get("div", 0) - 2 * cone) * cost["div"] + cone * (pv["div_near"] + 5) \
        + c.get("sqrt", 0) * pv["sqrt_near"] + c.get("rsqrt", 0) * pv["rsqrt_unit"]
    w = g("wavelength")
    out["wavelength"] = simple(w) + named(w, ("div", "sqrt", "rng_draw"))
    m = g("medium_per_photon")
    out["medium_per_photon"] = simple(m) + named(m, ("div",))
    # layer lengths (ice_factors + layer_lengths): per photon x = wlen/nm, -B/x (IEEE), two powr, one exp and four multiplies; per layer
    # visit (D a + E and 1 + 0.01 dTau folded on the host) 3 multiplies, 1 add and two exact reciprocals (rcp_: lengths bounded at Compile())
    # ... and RN(1 / length) from the length's own argument beside each of them (rcp_of_rcp_, round 4)
    out["layer_lengths"] = P * (2 * cost["div"] + 2 * cost["powr"] + cost["exp"] + 4) + EV * (4 + 2 * pv["rcp"] + 2 * pv["rcp_of_rcp"]) if g("layer_lengths").get("powr") else simple(g("layer_lengths"))
    # tilt: both divisors are invariants with a proof (div_by: 3 each)
    t = g("tilt")
    out["tilt"] = simple(t) + t.get("div", 0) * pv["div_inv"]
    # layer walk per trip: layer index / thickness (invariant, 3; with tilt every trip, else once per photon and counted under `other`),
    # (boundary - z) / scattering and absorption length (div_near, 8 each), 1 / length per crossing (rcp 3 each), 1 / dz on a trip that
    # crossed (rcp 3), the budget's division when the photon scatters (div_near 8); without anisotropy `budget *= 1; budget /= 1` vanish
    k = g("walk")
    # (round 4: the three divisions by a length take the reciprocal that came with the length, 5 each; the crossing updates need no reciprocal of their own)
    icecube = bool(g("layer_lengths").get("powr"))
    by_length = pv["div_with"] if icecube else pv["div_near"]
    walk_div = (T * pv["div_inv"] if tilt else 0) + T * 2 * by_length + (0 if icecube else CR * 2 * pv["rcp"]) + CT * pv["rcp"] + SC * by_length + (T * cost["div"] if aniso else 0)
    out["walk"] = simple(k) - (0 if aniso else T * cost["mul"]) + named(k, ("log", "rng_draw")) + walk_div
    a = g("aniso")
    out["aniso"] = (simple(a) + a.get("div", 0) * (pv["rcp"] + 1)) if a else 0        # 2/x = 2 RN(1/x)
    # scattering angle: the selector's division by f or 1-f (invariant, 3); Liu: powr_unit; Henyey-Greenstein: div_near + invariant
    s = g("scatter_angle")
    out["scatter_angle"] = simple(s) + named(s, ("powr_unit", "powr", "rng_draw")) + (ev["liu"] + ev["hg"]) * pv["div_inv"] + ev["hg"] * (pv["div_near"] + pv["div_inv"])
    r = g("rotate")
    out["rotate"] = simple(r) + named(r, ("sincos",)) + SC * (pv["div_near"] + 5) + r.get("sqrt", 0) * pv["sqrt_near"] + r.get("rsqrt", 0) * pv["rsqrt_unit"]
    x = g("transform")
    out["transform"] = (simple(x) + x.get("rsqrt", 0) * pv["rsqrt_near"]) if x else 0
    d = g("advance")
    out["advance"] = simple(d) + d.get("sqrt", 0) * pv["sqrt_near"] + named(d, ("rng_draw",))
    o = g("other")
    out["other"] = simple(o) + o.get("div", 0) * pv["div_inv"]                      # loop tests; the layer index of a photon born without tilt
    h = g("hit_record")
    out["hit_record"] = price_as_written({"h": h}, cost)["h"] if h else 0
    out["per_step"] = price_as_written({"s": g("per_step")}, cost)["s"] if g("per_step") else 0
    return out


def workload(name, n):
    ice = {"c2": "spice_mie", "c3": "spice_lea", "c5": "spice_lea"}[name]
    g = S.ic86_geometry()
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    med = B.load_ppc_ice(os.path.join(ROOT, "clsim_amd", "data", "ice", ice))
    bias = B.icecube_dom_acceptance()
    gens = [B.cherenkov_wlen_generator(bias, med)]
    if name == "c5":
        gens.append(dict(kind="const", value=405e-9))
        k = int(np.argmin(np.abs(g["x"]) + np.abs(g["y"]) + np.abs(g["z"] + 100.0)))
        steps = S.flasher_steps(n, seed=1000, photons_per_step=400, position=(float(g["x"][k]), float(g["y"][k]), float(g["z"][k])))
    else:
        steps = S.cascade_steps(n, seed=1000, photons_per_step=200)          # bench.py: make_bunch(0) of rank 0
    T = capi.make_tables(med, geo, gens, bias, pancake=5.0)
    return T, steps, ice


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["c2", "c3", "c5"])
    ap.add_argument("--steps", type=int, default=16384)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06", "reference_ops.json"))
    args = ap.parse_args()
    cost, proven = unit_costs()
    cost_c, proven_c = unit_costs(cycles=True)
    src = b"".join(open(os.path.join(ROOT, "oracle", f), "rb").read() for f in ("clsim_oracle.c", "oracle_math.h", "count_ops.hpp", "count_ops_calls.hpp"))
    res = {"what": __doc__.split("\n\n")[0], "unit_costs_generic": cost, "unit_costs_proven": proven, "unit_issue_slots_generic": cost_c, "unit_issue_slots_proven": proven_c, "oracle_sha16": hashlib.sha256(src).hexdigest()[:16],
           "git_revision": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(), "workloads": {}}
    from clsim_amd import converter as CV
    for name in args.workloads:
        T, steps, ice = workload(name, args.steps)
        a = CV.mwc_multipliers(len(steps)); x = CV.seed_streams(a)
        t0 = time.time()
        (ph, cnt, _, iters), ops, ev = capi.count_ops(T, steps, x, a, threads=os.cpu_count() or 8)
        P = ev["photons"]
        assert P == int(steps["num"].sum()) and ev["trips"] == iters
        draws = sum(d.get("rng_draw", 0) for d in ops.values())
        birth = 4 if name != "c5" else 2
        assert draws == birth * P + ev["trips"] + 2 * ev["scatters"], "SURVEY 9.1: draws per photon, trip and scatter"
        aw = price_as_written(ops, cost)
        tr = price_transformed(ops, ev, cost, proven, aniso=(ice == "spice_lea"), tilt=True)
        aw_c = price_as_written(ops, cost_c)
        tr_c = price_transformed(ops, ev, cost_c, proven_c, aniso=(ice == "spice_lea"), tilt=True)
        search = sum(v for k, v in aw.items() if k.startswith("search"))
        totals = {}
        for d in ops.values():
            for o, v in d.items():
                totals[o] = totals.get(o, 0) + v
        res["workloads"][name] = {
            "sample": "%d steps of bench.py's bunch (%s), %d photons, %d hits, %.1f s" % (len(steps), ice, P, cnt, time.time() - t0),
            "events_per_photon": {k: v / P for k, v in ev.items()},
            "operations_per_photon_as_written": {k: v / P for k, v in sorted(totals.items())},
            "operations_per_photon_by_region": {r: {o: v / P for o, v in d.items()} for r, d in ops.items()},
            "valu_per_photon_as_written_by_region": {k: v / P for k, v in aw.items()},
            "valu_per_photon_transformed_by_region": {k: v / P for k, v in tr.items()},
            "valu_per_photon": {"as_written": sum(aw.values()) / P, "as_written_without_search": (sum(aw.values()) - search) / P,
                                "transformed": sum(tr.values()) / P},
            "issue_slots_per_photon": {"as_written": sum(aw_c.values()) / P, "transformed": sum(tr_c.values()) / P,
                                       "note": "the same sums with every quarter-rate instruction counted as four full-rate issue slots"},
            "valu_per_trip": {"as_written": sum(aw.values()) / ev["trips"], "transformed": sum(tr.values()) / ev["trips"]}}
        w = res["workloads"][name]
        print("%s: %s" % (name, w["sample"]))
        print("   trips/photon %.2f, crossings/trip %.3f, searched strings/trip %.3f, DOM tests/trip %.4f" %
              (ev["trips"] / P, ev["layer_crossings"] / ev["trips"], ev["strings"] / ev["trips"], ev["dom_tests"] / ev["trips"]))
        print("   VALU-equivalents per photon: as written %.0f (without the search %.0f), transformed %.0f  [per trip %.0f / %.0f]" %
              (w["valu_per_photon"]["as_written"], w["valu_per_photon"]["as_written_without_search"], w["valu_per_photon"]["transformed"],
               w["valu_per_trip"]["as_written"], w["valu_per_trip"]["transformed"]))
        print("   full-rate issue slots per photon (quarter-rate instructions x 4): as written %.0f, transformed %.0f" % (sum(aw_c.values()) / P, sum(tr_c.values()) / P))
        for k in tr:
            print("     %-20s as written %8.1f   transformed %8.1f" % (k, aw.get(k, 0) / P, tr[k] / P))
        for k in aw:
            if k.startswith("search"):
                print("     %-20s as written %8.1f   transformed        -" % (k, aw[k] / P))
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
