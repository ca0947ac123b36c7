"""Outputs of the reference's OpenCL kernel text itself.

tests/golden/verbatim_cl_<case>.npz were written by tools/verbatim_cl_check.py in the build container: the files
resources/kernels/{mwcrng_kernel, propagation_kernel.h, sparse_collision_kernel.h, sparse_collision_kernel.c,
propagation_kernel.c, spherical_coordinates.c}.cl of the reference compiled VERBATIM for x86-64 (ROCm clang, OpenCL C 1.2)
behind a generated section emitted from oracle/builders.py and OpenCL builtins on oracle/oracle_math.h, run work item by work
item over seeded step bunches.  They hold outputs only (sorted 80-byte hit records, photon histories, final RNG states; for
the table maker: entry counts, the SHA-256 of the entry stream and the table it adds up to); the inputs are regenerated here
from the same seeds.  Cases = the #ifdef branches of the static kernel files in use: plain (C1, SPICE-Mie, SPICE-Lea, flasher),
per-layer tables in 16 bits with tabulated refractive indices, SAVE_PHOTON_HISTORY, a fixed absorption budget, no pancake
factor, a flasher LED's measured spectrum (InterpolatedDistribution with its own x values), -DTABULATE with 4 axes / full azimuth / the impact-angle axis / too little entry space, and the search without
STOP_PHOTONS_ON_DETECTION (`*_keep`: SetStopDetectedPhotons(false) -- every DOM on a segment's way is saved and the photon
travels on; `clear*` = ice with a 600 m scattering length, where the quirks of that branch's bit masks decide 5 % of the hits).

What they pin: the transcription of the static kernel files by oracle/clsim_oracle.c (CPU tests below) and by the HIP kernels
(GPU tests) -- a guard against one author misreading 1 500 lines of OpenCL C twice in the same way.  They do not pin the
generated section nor the math library, which are this repository's on every side."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from clsim_amd import synthetic as S
from clsim_amd.synthetic import PHOTON_DTYPE
from oracle import builders as B
from oracle import capi
from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# case -> (configuration, converter options)   [tools/verbatim_cl_check.py: CASES]
CASES = {"c1": ("c1", {}), "mie": ("mie", {}), "lea": ("lea", {}), "flasher": ("flasher", {}), "photonics_mie": ("photonics_mie", {}),
         "mie_history": ("mie", dict(history=4)), "mie_fixed_abs": ("mie", dict(fixed_abs=1.5)), "lea_no_pancake": ("lea", dict(pancake=1.0)),
         "flasher_led405": ("flasher_led405", {}), "lea_dispersion": ("lea_dispersion", {}),
         "c1_keep": ("c1", dict(stop_detected=False)), "mie_60_keep": ("mie_60", dict(stop_detected=False)),
         "flasher_60_keep": ("flasher_60", dict(stop_detected=False)), "clear_60_keep": ("clear_60", dict(stop_detected=False)),
         "clear_keep": ("clear", dict(stop_detected=False)), "lea_60_keep_history": ("lea_60", dict(stop_detected=False, history=4))}
TAB_CASES = ["tabulate", "tabulate360", "tabulate5", "tabulate_overflow"]
ANGULAR = [0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435]


def load(case):
    return np.load(os.path.join(ROOT, "tests", "golden", "verbatim_cl_%s.npz" % case))


def inputs(case):
    name, opt = CASES[case]
    f = load(case)
    cfg = common.config(name)
    steps = common.steps_for(cfg, int(f["n_steps"]), seed=int(f["seed"]))
    x, a = common.streams(len(steps))
    return cfg, opt, steps, x, a, f


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_equals_the_verbatim_kernel(case):
    cfg, opt, steps, x, a, f = inputs(case)
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    bias = B.icecube_dom_acceptance()
    gens = common.oracle_generators(cfg, bias)
    T = capi.make_tables(cfg["med_o"], geo, gens, bias, pancake=opt.get("pancake", 5.0), stop_detected=opt.get("stop_detected", True),
                         fixed_abs_lengths=opt.get("fixed_abs"), history_entries=opt.get("history", 0))
    hits = np.frombuffer(f["hits"].tobytes(), dtype=PHOTON_DTYPE)
    if opt.get("history"):
        ph_o, cnt_o, x_o, _, hist_o = capi.propagate(T, steps, x, a, history=True)
        # serial on both sides: same order; the ring entries ConvertPhotonHistories reads (OpenCL.cxx:940-989)
        assert ph_o.tobytes() == f["unsorted_hits"].tobytes()
        n = hist_o.shape[1]
        for i, ns in enumerate(ph_o["numScatters"]):
            cur = 0 if ns <= n else int(ns) % n
            for j in range(min(int(ns), n)):
                assert hist_o[i, cur].tobytes() == f["histories"][i, j].tobytes()
                cur = (cur + 1) % n
    else:
        ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    assert cnt_o == len(hits) and cnt_o > 100
    assert common.sort_photons(ph_o).tobytes() == common.sort_photons(hits).tobytes()
    assert np.array_equal(x_o, f["rng_x"])


def tab_setup(case):
    cfg = common.config("mie")
    if case == "tabulate5":
        axes = [B.power_axis(0, 580, 40, 2), B.linear_axis(0, 180, 8), B.linear_axis(-1, 1, 20), B.power_axis(0, 7e3, 21, 2), B.linear_axis(-1, 1, 10)]
    elif case == "tabulate360":
        axes = [B.power_axis(0, 300, 30, 2), B.linear_axis(0, 360, 24), B.linear_axis(-1, 1, 20), B.power_axis(0, 3e3, 30, 2)]
    else:
        axes = [B.power_axis(0, 580, 40, 2), B.linear_axis(0, 180, 8), B.linear_axis(-1, 1, 20), B.power_axis(0, 7e3, 21, 2)]
    f = load(case)
    tb = B.tabulator_config("spherical", axes, cfg["med_o"], ANGULAR, entries_per_stream=int(f["entries_per_stream"]))
    steps = S.cascade_steps(64, seed=5, vertex=(3.0, -2.0, 10.0), photons_per_step=12, pad_to=64)
    x, a = common.streams(64)
    ref = B.reference_particle((1.0, 0.5, -2.0), 3.0, (0.3, -0.2, 0.9327379053))
    return cfg, axes, tb, steps, x, a, ref, f


@pytest.mark.parametrize("case", TAB_CASES)
def test_oracle_table_maker_equals_the_verbatim_kernel(case):
    """-DTABULATE: every table entry (bin index, weight) of every stream in order, entry counts, photons left, RNG states"""
    cfg, axes, tb, steps, x, a, ref, f = tab_setup(case)
    bias = B.icecube_dom_acceptance()
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    T = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias, cfg["med_o"])], bias, pancake=1.0, tabulator=tb)
    ent, num, left, x_o = capi.tabulate(T, steps, x, a, ref, threads=8)
    assert np.array_equal(num, f["num"]) and np.array_equal(left, f["left"]) and np.array_equal(x_o, f["rng_x"])
    flat = np.concatenate([ent[i, :num[i]] for i in range(len(num))])
    assert hashlib.sha256(flat.tobytes()).digest() == f["entries_sha256"].tobytes()
    if case == "tabulate_overflow":
        assert left.sum() > 0              # streams ran out of entry space and came back with photons left (c.cl:770-776)


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_hip_path_equals_the_verbatim_kernel(case):
    cfg, opt, steps, x, a, f = inputs(case)
    hits = np.frombuffer(f["hits"].tobytes(), dtype=PHOTON_DTYPE)
    conv = common.product_converter(cfg, len(steps), pancake=opt.get("pancake", 5.0), initialize=False, stop_detected=opt.get("stop_detected", True))
    if opt.get("fixed_abs"):
        conv.SetFixedNumberOfAbsorptionLengths(opt["fixed_abs"])
    if opt.get("history"):
        conv.SetPhotonHistoryEntries(opt["history"])
    conv.SetMaxNumWorkitems(len(steps))
    conv.Compile()
    conv.InitializeWithStreams(x, a)
    conv.EnqueueSteps(steps, 7)
    if opt.get("history"):
        ident, ph_p, hist_p = conv.GetConversionResult(with_histories=True)
    else:
        ident, ph_p = conv.GetConversionResult()
    assert ident == 7 and len(ph_p) == len(hits)
    # the converter hands out string / DOM IDs (OpenCL.cxx:1565-1619): the same translation applied to the kernel's indices
    T = common.oracle_tables(cfg)
    expect = capi.replace_indices_with_ids(hits.copy(), T.geo)
    assert common.sort_photons(ph_p).tobytes() == common.sort_photons(expect).tobytes()
    assert np.array_equal(conv.GetRNGState(len(steps)), f["rng_x"])
    if opt.get("history"):
        # a photon's history travels with it: match on the whole record
        by_record = {expect_i.tobytes(): f["histories"][i] for i, expect_i in enumerate(capi.replace_indices_with_ids(
            np.frombuffer(f["unsorted_hits"].tobytes(), dtype=PHOTON_DTYPE).copy(), T.geo))}
        for i in range(len(ph_p)):
            ref_hist = by_record[ph_p[i].tobytes()]
            k = len(hist_p[i])
            assert np.asarray(hist_p[i], dtype=np.float32).tobytes() == ref_hist[:k].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tabulate", "tabulate360"])
def test_hip_table_maker_equals_the_verbatim_kernel(case):
    """the product adds every path sample into its bin in binary64 (hardware atomics): its table equals the verbatim
    kernel's entries summed in binary64, to the rounding of a different summation order"""
    from clsim_amd import converter as CV, tabulator as TB
    cfg, axes, tb, steps, x, a, ref, f = tab_setup(case)
    p_axes = TB.SphericalAxes([(TB.PowerAxis if ax["kind"] == "power" else TB.LinearAxis)(*((ax["min"], ax["max"], ax["n_bins"]) +
                              ((ax["power"],) if ax["kind"] == "power" else ()))) for ax in axes])
    # the table maker takes bunches and stream sets in multiples of 256: the 64 steps of the fixture, then steps without
    # photons (they draw nothing); a stream set's first 64 streams are the set of 64
    x256, a256 = common.streams(256)
    assert np.array_equal(x256[:64], x) and np.array_equal(a256[:64], a)
    padded = S.cascade_steps(64, seed=5, vertex=(3.0, -2.0, 10.0), photons_per_step=12, pad_to=256)
    assert np.array_equal(padded[:64], steps) and not padded["num"][64:].any()
    tab = TB.I3CLSimStepToTableConverterHIP(0, p_axes, False, cfg["med_p"], np.pi * 0.16510 ** 2, CV.GetIceCubeDOMAcceptance(),
                                            TB.I3CLSimFunctionPolynomial(ANGULAR), (x256, a256))
    tab.EnqueueSteps(padded, tuple(float(v) for v in (ref[0], ref[1], ref[2], ref[3], ref[4], ref[5], ref[6])))
    tab.Finish()
    sums = tab.GetBinSums().ravel()
    expect = np.zeros(int(f["n_bins"]), dtype=np.float64)
    expect[f["bins_nonzero"]] = f["bins_sum"]
    assert len(sums) == len(expect)
    assert np.array_equal(sums > 0, expect > 0)
    nz = expect > 0
    assert np.max(np.abs(sums[nz] - expect[nz]) / expect[nz]) < 1e-12


@pytest.mark.skipif(not os.path.isdir("/root/reference/resources/kernels"), reason="the reference tree is not on this machine")
def test_fixture_is_what_the_reference_kernel_text_yields_today():
    """build container only: recompile the reference's .cl files and run the smallest configuration again (the tool exits
    non-zero unless the verbatim kernel and the oracle agree bit for bit); `tools/verbatim_cl_check.py` runs all twenty"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verbatim_cl_check.py"), "--configs", "c1"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "hit records IDENTICAL | final RNG states IDENTICAL" in p.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/resources/kernels"), reason="the reference tree is not on this machine")
def test_the_two_refused_modes_do_not_compile_in_the_reference_either():
    """build container only.  Compile() refuses DoublePrecision and SaveAllPhotons because the reference's own
    program cannot be built in them at this revision: with DOUBLE_PRECISION propagation_kernel.c.cl:872 hands the double4 direction
    to a function taking float4* -- for every medium, the call is unconditional --, and with SAVE_ALL_PHOTONS the geometry source is
    left out (OpenCL.cxx:459-470) while saveHit still calls geometryGetDomPosition (c.cl:339)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verbatim_cl_check.py"), "--refused-modes"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = {l.split()[0]: l for l in p.stdout.splitlines() if l.startswith(("DOUBLE_PRECISION", "SAVE_ALL_PHOTONS"))}
    assert "DOES NOT COMPILE" in lines["DOUBLE_PRECISION"] and "double4" in lines["DOUBLE_PRECISION"] and "float4" in lines["DOUBLE_PRECISION"]
    assert "DOES NOT COMPILE" in lines["SAVE_ALL_PHOTONS"] and "geometryGetDomPosition" in lines["SAVE_ALL_PHOTONS"]
