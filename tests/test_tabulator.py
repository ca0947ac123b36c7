"""Photon table maker: host logic of clsim::tabulator restated in oracle/builders.py against the
product's C++ (CPU part), and the TABULATE kernel against the oracle (GPU part)."""
import math
import os

import numpy as np
import pytest

from clsim_amd import converter as CV
from clsim_amd import synthetic as S
from clsim_amd import tabulator as TB
from oracle import builders as B
from oracle import capi
from tests import common

# resources/scripts/compareToPPCredux/test_ice_models/lea/as.dat, rows 1.. (python/GetIceCubeDOMAngularSensitivity.py:45-47)
ANGULAR = [0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435]
DOM_AREA = math.pi * 0.16510 ** 2


def axes_pair(kind, small=True):
    """python/tablemaker/tabulator.py:621-641 (defaults), optionally with fewer bins."""
    f = 5 if small else 1
    if kind == "spherical":
        o = [B.power_axis(0, 580, 200 // f, 2), B.linear_axis(0, 180, (36 if f == 1 else 8)), B.linear_axis(-1, 1, 100 // f), B.power_axis(0, 7e3, 105 // f, 2)]
        p = TB.SphericalAxes([TB.PowerAxis(0, 580, 200 // f, 2), TB.LinearAxis(0, 180, (36 if f == 1 else 8)), TB.LinearAxis(-1, 1, 100 // f), TB.PowerAxis(0, 7e3, 105 // f, 2)])
    elif kind == "spherical360":
        o = [B.power_axis(0, 300, 30, 2), B.linear_axis(0, 360, 24), B.linear_axis(-1, 1, 20), B.power_axis(0, 3e3, 30, 2)]
        p = TB.SphericalAxes([TB.PowerAxis(0, 300, 30, 2), TB.LinearAxis(0, 360, 24), TB.LinearAxis(-1, 1, 20), TB.PowerAxis(0, 3e3, 30, 2)])
    elif kind == "spherical_small":
        # a table most photons leave within a few scattering lengths: the trips that may not carry samples (save_path_wave_carry's
        # inside-the-table test fails next to the distance and time axes' ends) and the out-of-bounds bookkeeping, all the time
        o = [B.power_axis(0, 60, 20, 2), B.linear_axis(0, 180, 8), B.linear_axis(-1, 1, 10), B.power_axis(0, 400, 20, 2)]
        p = TB.SphericalAxes([TB.PowerAxis(0, 60, 20, 2), TB.LinearAxis(0, 180, 8), TB.LinearAxis(-1, 1, 10), TB.PowerAxis(0, 400, 20, 2)])
    elif kind == "spherical5":
        # a fifth axis (cosine of the impact angle) switches TABULATE_IMPACT_ANGLE on (StepToTableConverter.cxx:187-188)
        o = [B.power_axis(0, 580, 40, 2), B.linear_axis(0, 180, 8), B.linear_axis(-1, 1, 20), B.power_axis(0, 7e3, 21, 2), B.linear_axis(-1, 1, 10)]
        p = TB.SphericalAxes([TB.PowerAxis(0, 580, 40, 2), TB.LinearAxis(0, 180, 8), TB.LinearAxis(-1, 1, 20), TB.PowerAxis(0, 7e3, 21, 2), TB.LinearAxis(-1, 1, 10)])
    elif kind == "spherical_cuberoot":
        # power 3 (inverse transform cbrt) and power 4 (pow(x, 0.25)): tabulator/Axis.cxx:150-171
        o = [B.power_axis(0, 580, 40, 3), B.linear_axis(0, 180, 8), B.linear_axis(-1, 1, 20), B.power_axis(0, 7e3, 30, 4)]
        p = TB.SphericalAxes([TB.PowerAxis(0, 580, 40, 3), TB.LinearAxis(0, 180, 8), TB.LinearAxis(-1, 1, 20), TB.PowerAxis(0, 7e3, 30, 4)])
    elif kind == "cylindrical5":
        o = [B.power_axis(0, 580, 20, 2), B.linear_axis(0, math.pi, 8), B.linear_axis(-8e2, 8e2, 16), B.power_axis(0, 7e3, 21, 2), B.linear_axis(-1, 1, 10)]
        p = TB.CylindricalAxes([TB.PowerAxis(0, 580, 20, 2), TB.LinearAxis(0, math.pi, 8), TB.LinearAxis(-8e2, 8e2, 16), TB.PowerAxis(0, 7e3, 21, 2), TB.LinearAxis(-1, 1, 10)])
    else:
        o = [B.power_axis(0, 580, 100 // f, 2), B.linear_axis(0, math.pi, (36 if f == 1 else 8)), B.linear_axis(-8e2, 8e2, 80 // f), B.power_axis(0, 7e3, 105 // f, 2)]
        p = TB.CylindricalAxes([TB.PowerAxis(0, 580, 100 // f, 2), TB.LinearAxis(0, math.pi, (36 if f == 1 else 8)), TB.LinearAxis(-8e2, 8e2, 80 // f), TB.PowerAxis(0, 7e3, 105 // f, 2)])
    return o, p


def test_axes_host_logic_follows_the_reference():
    """Axis.cxx / Axes.cxx: index literals, layout, bin edges and volumes (oracle restatement; the product's C++ is
    compared with it on the GPU, where the tabulator object can be constructed)."""
    o, _ = axes_pair("spherical", small=False)
    shape, strides, n = B.axes_layout(o)
    assert shape == [202, 38, 102, 107] and strides == [38 * 102 * 107, 102 * 107, 107, 1] and n == 202 * 38 * 102 * 107
    sc, of = B.axis_index_literals(o[0])                  # PowerAxis(0, 580, 200, 2): scale = 200/sqrt(580)
    assert sc == np.float32(200.0 / math.sqrt(580.0)) and of == np.float32(0.0)
    sc, of = B.axis_index_literals(o[2])                  # LinearAxis(-1, 1, 100): 50*x - (-50)
    assert sc == np.float32(50.0) and of == np.float32(-50.0)
    e = B.axis_bin_edges(o[0])
    assert len(e) == 201 and e[0] == 0.0 and abs(e[-1] - 580.0) < 1e-9 and np.allclose(np.sqrt(e), np.linspace(0, math.sqrt(580.0), 201))
    # the spherical bin volumes add up to half a sphere of r = 580 m in (r, azimuth [deg], cos) up to the factor 2 for
    # the folded azimuth: (r^3/3) * 2*pi * 2
    total = sum(B.bin_volume("spherical", o, [i, j, k]) for i in (0, 57, 199) for j in (0, 35) for k in (0, 99))
    assert total > 0
    full = (580.0 ** 3 / 3.0) * 2 * (math.pi / 180.0) * 180.0 * 2.0
    vol = sum(B.bin_volume("spherical", o, [i, 0, 0]) for i in range(200)) * 36 * 100
    assert abs(vol / full - 1) < 1e-9
    n_group, n_phase = B.minimum_refractive_index(common.config("mie")["med_o"])
    assert 1.3 < n_phase < 1.36 and 1.3 < n_group < 1.4
    ref = B.reference_particle((1., 2., 3.), 4., (0.0, 0.6, 0.8))
    assert np.allclose(ref[8:11], (0.0, -0.8, 0.6)) and abs(np.dot(ref[4:7], ref[8:11])) < 1e-7


def test_oracle_tabulator_entries_and_misses():
    """The restated TABULATE kernel: entries stay inside the table, weights fall with depth, a stream that runs out of
    entry space comes back with its photons left and its RNG rewound (propagation_kernel.c.cl:770-776)."""
    cfg = common.config("mie")
    o, _ = axes_pair("spherical")
    tb = B.tabulator_config("spherical", o, cfg["med_o"], ANGULAR, entries_per_stream=40000)
    bias = B.icecube_dom_acceptance()
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    T = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias, cfg["med_o"])], bias, pancake=1.0, tabulator=tb)
    steps = S.cascade_steps(64, seed=5, vertex=(0.0, 0.0, 0.0), photons_per_step=12, pad_to=64)
    x, a = common.streams(64)
    ref = B.reference_particle((0., 0., 0.), 0.0, (0.0, 0.0, 1.0))
    ent, num, left, xo = capi.tabulate(T, steps, x, a, ref)
    assert left.sum() == 0 and num.min() > 1000 and num.max() < 40000
    k = int(num[0])
    assert ent["index"][0, :k].max() < tb["n_bins"] and np.all(ent["weight"][0, :k] > 0) and np.all(ent["weight"][0, :k] <= steps["weight"][0] * 1.2)
    # too little space: the same launch stops early, keeps whole photons only and rewinds the stream
    tb2 = dict(tb, entries_per_stream=1500)
    T2 = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias, cfg["med_o"])], bias, pancake=1.0, tabulator=tb2)
    ent2, num2, left2, x2 = capi.tabulate(T2, steps, x, a, ref)
    assert left2.max() > 0 and np.all(num2 <= 1500)
    i = int(np.argmax(left2 > 0))
    assert np.array_equal(ent2[i, :num2[i]], ent[i, :num2[i]])        # what was recorded is a prefix of the full record
    # re-running the returned steps from the rewound streams reproduces the missing photons: totals agree
    again = steps.copy(); again["num"] = left2
    T3 = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias, cfg["med_o"])], bias, pancake=1.0, tabulator=tb)
    ent3, num3, left3, x3 = capi.tabulate(T3, again, x2, a, ref)
    assert left3.sum() == 0 and np.array_equal(x3, xo)


def test_oracle_accumulate_mode_equals_the_entry_buffers():
    """oracle_tabulate_accumulate (the table maker's host loop around the kernel: every step in calls of a few photons, entries added up and
    dropped -- what the at-size GPU test checks against) gives the entry-buffer form's bins, counts and final streams; a call that runs out of
    entry space is an error, not a silent double count (the reference counts an interrupted photon's first segments twice, c.cl:770-776)."""
    cfg = common.config("mie")
    o, _ = axes_pair("spherical")
    bias = B.icecube_dom_acceptance()
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    tb = B.tabulator_config("spherical", o, cfg["med_o"], ANGULAR, entries_per_stream=40000)
    T = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias, cfg["med_o"])], bias, pancake=1.0, tabulator=tb)
    steps = S.cascade_steps(48, seed=5, vertex=(0.0, 0.0, 0.0), photons_per_step=12, pad_to=48)
    steps["num"][7] = 0
    x, a = common.streams(48)
    ref = B.reference_particle((0., 0., 0.), 0.0, (0.0, 0.0, 1.0))
    ent, num, left, xo = capi.tabulate(T, steps, x, a, ref)
    assert left.sum() == 0
    bins = np.zeros(tb["n_bins"], dtype=np.float64)
    sums, counts, xa = capi.tabulate_accumulate(T, steps, x, a, ref, bins=bins, photons_per_call=5, threads=4)
    assert np.array_equal(xa, xo) and np.array_equal(counts, num.astype(np.uint64)) and counts[7] == 0
    expect = capi.accumulate_entries(ent, num, tb["n_bins"], dtype=np.float64)
    assert np.array_equal(expect > 0, bins > 0) and np.allclose(expect, bins, rtol=1e-12, atol=0)
    assert np.allclose(sums, [ent["weight"][i, :int(num[i])].astype(np.float64).sum() for i in range(48)], rtol=1e-12)
    tiny = dict(tb, entries_per_stream=1500)
    T2 = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias, cfg["med_o"])], bias, pancake=1.0, tabulator=tiny)
    with pytest.raises(RuntimeError):
        capi.tabulate_accumulate(T2, steps, x, a, ref, photons_per_call=5, threads=4)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,ice,step_length", [("spherical", "mie", 1.0), ("cylindrical", "lea", 1.0), ("spherical360", "photonics_mie", 1.0),
                                                  ("spherical", "lea", 0.2), ("spherical5", "mie", 1.0), ("cylindrical5", "lea", 1.0),
                                                  ("spherical5", "lea", 0.2), ("spherical_cuberoot", "mie", 1.0),
                                                  # round 5 (samples carried across loop trips): a pool that overflows in some trips and not in
                                                  # others, and a table whose ends are never far
                                                  ("spherical", "mie", 0.3), ("spherical_small", "mie", 1.0), ("spherical_small", "lea", 0.5)])
def test_table_matches_the_oracle(kind, ice, step_length):
    check_table_against_the_oracle(kind, ice, step_length)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,ice", [("spherical", "mie"), ("cylindrical5", "lea")])
def test_fast_table_instantiations_match_the_oracle_too(kind, ice):
    """round 4: the TABULATE kernels have FAST instantiations (the medium's proofs compiled in, as in the propagation kernels);
    measured slower than the generic ones, so they run only after clsimhip_tabulator_set_tuning("fast_kernels", 1) -- under the same check"""
    check_table_against_the_oracle(kind, ice, 1.0, fast_kernels=True)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,ice,step_length,standard", [("spherical", "mie", 1.0, True), ("spherical", "mie", 1.0, False), ("spherical", "lea", 0.2, True),
                                                           ("spherical", "mie", 0.3, True), ("spherical_small", "mie", 1.0, True), ("spherical_small", "lea", 0.5, True),
                                                           ("spherical_small", "lea", 0.5, False)])
def test_standard_table_sampler_matches_the_oracle(kind, ice, step_length, standard):
    """round 6: a table of the reference's default shape (spherical, folded azimuth, square-root distance and time axes, no squared weights:
    python/tablemaker/tabulator.py:621-641) runs the sampler specialised for it (prop_kernel.hip: sample_bin<..., STD>); with
    clsimhip_tabulator_set_tuning("standard_sampler", 0) the generic one.  Both against the oracle, same bar as every other table."""
    check_table_against_the_oracle(kind, ice, step_length, squared=False, standard_sampler=standard)


def check_table_against_the_oracle(kind, ice, step_length, expect_fast="by medium", fast_kernels=False, squared=True, standard_sampler=None):
    """prop_kernel<TAB> adds every path sample to its bin with an fp64 atomic; the oracle writes the reference's
    (bin, weight) entries.  Same samples <=> the double precision sums agree to rounding; the float image agrees with
    the reference's in-order float accumulation to float accuracy.  step_length 0.2 m makes most waves exceed the
    sample pool, i.e. exercises the per-lane walk next to the pooled one.  The "5" kinds have the impact-angle axis:
    two random numbers per sample from the photon's own stream, so the final stream states check the draw count."""
    cfg = common.config(ice)
    o, p = axes_pair(kind)
    okind = "cylindrical" if kind.startswith("cylindrical") else "spherical"
    fine = step_length != 1.0
    tb = B.tabulator_config(okind, o, cfg["med_o"], ANGULAR, step_length=step_length, entries_per_stream=60000)
    bias_o = B.icecube_dom_acceptance()
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    T = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias_o, cfg["med_o"])], bias_o, pancake=1.0, tabulator=tb)
    n = 256 if fine else 512
    steps = S.cascade_steps(n, seed=15, vertex=(10.0, -20.0, 30.0), photons_per_step=(3 if fine else 14), pad_to=256)
    steps["weight"] = np.random.Generator(np.random.PCG64(3)).uniform(0.5, 1.5, n).astype(np.float32)
    steps["num"][5] = 0
    x, a = common.streams(n)
    refdir = (math.sin(0.7) * math.cos(0.3), math.sin(0.7) * math.sin(0.3), math.cos(0.7))
    ref7 = (10.0, -20.0, 30.0, 2.0) + refdir
    ref_o = B.reference_particle(ref7[:3], ref7[3], refdir)
    bins64 = np.zeros(tb["n_bins"], dtype=np.float64)
    bins32 = np.zeros(tb["n_bins"], dtype=np.float32)
    sq64 = np.zeros(tb["n_bins"], dtype=np.float64)
    xo = x
    for bunch in range(2):                       # streams carry over to the second bunch
        ent, num, left, xo = capi.tabulate(T, steps, xo, a, ref_o)
        assert left.sum() == 0
        for i in range(n):
            k = int(num[i])
            np.add.at(bins64, ent["index"][i, :k], ent["weight"][i, :k].astype(np.float64))
            np.add.at(sq64, ent["index"][i, :k], ent["weight"][i, :k].astype(np.float64) ** 2)
            np.add.at(bins32, ent["index"][i, :k], ent["weight"][i, :k])
    tab = TB.I3CLSimStepToTableConverterHIP(0, p, squared, cfg["med_p"], DOM_AREA, CV.GetIceCubeDOMAcceptance(),
                                            TB.I3CLSimFunctionPolynomial(ANGULAR), (x, a), stepLength=step_length)
    assert tab.n_bins == tb["n_bins"] and list(tab.shape) == tb["shape"]
    # the specialised sampler is chosen for exactly one shape of table (kparams.h: tab_std)
    is_standard = kind in ("spherical", "spherical_small") and not squared
    assert int(tab.GetTable("TABULATOR_STANDARD_SAMPLER")[0]) == (1 if is_standard else 0)
    if standard_sampler is not None:
        tab.SetTuning("standard_sampler", 1 if standard_sampler else 0)
    if fast_kernels:
        tab.SetTuning("fast_kernels", 1)
        with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception):
            tab.SetTuning("no_such_key", 1)
    if expect_fast == "by medium":
        assert int(tab.GetTable("fast_variant")[0]) == (0 if ice.startswith("photonics") else 1)
    for k in range(len(o)):
        assert np.array_equal(tab.GetBinEdges(k), B.axis_bin_edges(o[k]))
    assert np.array_equal(tab.GetTable("TABULATOR_SCALE"), np.array(tb["scale"], dtype=np.float64))
    assert np.array_equal(tab.GetTable("TABULATOR_OFFSET"), np.array(tb["offset"], dtype=np.float64))
    t = tab.GetTable("TABULATOR")
    assert t[0] == tb["n_group"] and t[1] == tb["n_phase"] and t[2] == tb["min_inv_groupvel"] and t[3] == tb["tan_thetac"]
    for bunch in range(2):
        tab.EnqueueSteps(steps, ref7)
    tab.Finish()
    got = tab.GetBinSums()
    assert bins64.sum() > 100 and (bins64 > 0).sum() > 1000
    assert (got > 0).sum() == (bins64 > 0).sum() and np.array_equal(got > 0, bins64 > 0)
    assert np.allclose(got, bins64, rtol=1e-12, atol=0)
    if squared:
        assert np.allclose(tab.GetBinSums(squared=True), sq64, rtol=1e-12, atol=0)
    assert np.array_equal(tab.GetRNGState(n), xo)
    raw = tab.GetBinContent().ravel()
    assert np.allclose(raw, bins32, rtol=2e-5, atol=1e-9)            # the reference's float accumulation, order dependent
    norm_o = B.normalize_table(got.astype(np.float32), okind, o, step_length, DOM_AREA)
    assert np.array_equal(tab.GetBinContent(normalized=True).ravel(), norm_o)
    st = tab.GetStatistics()
    assert st["NumPhotons"] == 2 * float(steps["num"].sum()) and st["NumKernelCalls"] == 2
    assert abs(st["SumOfPhotonWeights"] - 2 * float((steps["num"] * steps["weight"].astype(np.float64)).sum())) < 1e-6


def read_fits(path):
    """Minimal FITS reader: list of (header dict, numpy array) per HDU."""
    raw = open(path, "rb").read()
    assert len(raw) % 2880 == 0
    pos, hdus = 0, []
    while pos < len(raw):
        header, order = {}, []
        while True:
            block = raw[pos:pos + 2880]; pos += 2880
            done = False
            for k in range(36):
                card = block[80 * k:80 * k + 80].decode("ascii")
                if card.startswith("END"):
                    done = True
                    break
                if card.startswith("HIERARCH"):
                    key, _, val = card[9:].partition("=")
                elif card[8:10] == "= ":
                    key, val = card[:8], card[10:]
                else:
                    continue
                key, val = key.strip(), val.split("/")[0].strip()
                if val.startswith("'"):
                    header[key] = val.strip("'").strip()
                elif val in ("T", "F"):
                    header[key] = (val == "T")
                else:
                    header[key] = float(val) if any(c in val for c in ".E") else int(val)
                order.append(key)
            if done:
                break
        shape = [header["NAXIS%d" % (i + 1)] for i in range(header["NAXIS"])]
        count = int(np.prod(shape)) if shape else 0
        size = abs(header["BITPIX"]) // 8 * count
        dtype = {-32: ">f4", -64: ">f8"}[header["BITPIX"]]
        data = np.frombuffer(raw[pos:pos + size], dtype=dtype).reshape(shape[::-1]) if count else np.zeros(0)
        pos += (size + 2879) // 2880 * 2880
        header["_order"] = order
        hdus.append((header, data))
    return hdus


@pytest.mark.gpu
def test_fits_file_has_the_structure_the_reference_writes(tmp_path):
    """WriteFITSFile (StepToTableConverter.cxx:595-686): primary image = normalised bin content with reversed axis counts,
    HIERARCH _i3_ keywords, ERRORS and EDGESi extensions -- read back with a FITS parser written from the standard."""
    cfg = common.config("mie")
    o, p = axes_pair("spherical")
    x, a = common.streams(256)
    tab = TB.I3CLSimStepToTableConverterHIP(0, p, True, cfg["med_p"], DOM_AREA, CV.GetIceCubeDOMAcceptance(),
                                            TB.I3CLSimFunctionPolynomial(ANGULAR), (x, a), stepLength=1.0)
    steps = S.cascade_steps(256, seed=4, vertex=(0.0, 0.0, 0.0), photons_per_step=20, pad_to=256)
    tab.EnqueueSteps(steps, (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0))
    tab.Finish()
    path = tmp_path / "table.fits"
    tab.WriteFITSFile(path, {"zenith": 0.0, "energy": 1.0, "type": 11, "level": 1, "comment": "skipped: not a number", "n_photons": -1.0})
    hdus = read_fits(path)
    names = [h.get("EXTNAME", "PRIMARY") for h, _ in hdus]
    assert names == ["PRIMARY", "ERRORS", "EDGES0", "EDGES1", "EDGES2", "EDGES3"]
    h0, img = hdus[0]
    assert h0["SIMPLE"] is True and h0["BITPIX"] == -32 and h0["NAXIS"] == 4 and h0["EXTEND"] is True
    assert [h0["NAXIS%d" % (i + 1)] for i in range(4)] == list(tab.shape)[::-1]
    assert np.array_equal(img.astype(np.float32), tab.GetBinContent(normalized=True))
    st = tab.GetStatistics()
    assert h0["_i3_n_group"] == pytest.approx(st["n_group"]) and h0["_i3_n_phase"] == pytest.approx(st["n_phase"])
    assert h0["_i3_type"] == 11 and h0["_i3_level"] == 1 and h0["_i3_energy"] == 1.0 and "_i3_comment" not in h0
    # n_photons is the converter's: spectral bias factor (bare 300-600 nm Cherenkov photons per acceptance-weighted photon,
    # ~ 1/0.1 for the IceCube DOM) x sum of photon weights
    factor = h0["_i3_n_photons"] / st["SumOfPhotonWeights"]
    assert 5.0 < factor < 30.0
    h1, err = hdus[1]
    assert h1["XTENSION"] == "IMAGE" and np.array_equal(err.astype(np.float32), tab.GetBinContent(squared=True, normalized=True))
    for i in range(4):
        he, edges = hdus[2 + i]
        assert he["BITPIX"] == -64 and np.array_equal(edges.astype(np.float64), tab.GetBinEdges(i))
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="Could not create"):
        tab.WriteFITSFile(path)                       # fits_create_diskfile does not overwrite


def test_table_maker_refuses_an_acceptance_without_a_device_form():
    """FromTable(wlens, values) and a delta peak are host-only spectra (FromTable.cxx:169-170 throws when OpenCL code is asked of a
    table without equal spacing): as the table maker's wavelength acceptance they must be refused, not turned into a garbage
    device table.  Host-only configuration: no GPU needed."""
    medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(common.ICE, "spice_mie"))
    axes = TB.SphericalAxes([TB.PowerAxis(0, 580, 20, 2), TB.LinearAxis(0, 180, 6), TB.LinearAxis(-1, 1, 10), TB.PowerAxis(0, 7e3, 15, 2)])
    ang = TB.I3CLSimFunctionPolynomial([0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435])
    a = CV.mwc_multipliers(256)
    x = CV.seed_streams(a)
    wl, v = np.array([300e-9, 350e-9, 420e-9, 600e-9]), np.array([0.1, 0.3, 0.2, 0.05])
    for acceptance in (CV.I3CLSimFunctionFromTable(wl, v), CV.I3CLSimFunctionDeltaPeak(405e-9)):
        with pytest.raises(Exception) as err:
            TB.I3CLSimStepToTableConverterHIP(0, axes, False, medium, math.pi * 0.16510 ** 2, acceptance, ang, (x, a))
        assert "bias" in str(err.value) or "acceptance" in str(err.value)


def test_table_maker_refuses_a_medium_without_a_group_index_override():
    """GetMinimumRefractiveIndex (I3CLSimStepToTableConverter.cxx:103-104) log_fatal()s when a layer has no group refractive index
    override; the propagator takes such a medium (group velocity from the dispersion), the table maker does not.  Host-only."""
    medium = common.config("mie_dispersion")["med_p"]
    axes = TB.SphericalAxes([TB.PowerAxis(0, 580, 20, 2), TB.LinearAxis(0, 180, 6), TB.LinearAxis(-1, 1, 10), TB.PowerAxis(0, 7e3, 15, 2)])
    ang = TB.I3CLSimFunctionPolynomial([0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435])
    a = CV.mwc_multipliers(256)
    x = CV.seed_streams(a)
    with pytest.raises(Exception, match="group refractive indices"):
        TB.I3CLSimStepToTableConverterHIP(0, axes, False, medium, math.pi * 0.16510 ** 2, CV.GetIceCubeDOMAcceptance(), ang, (x, a))
