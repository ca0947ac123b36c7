"""Host logic, no GPU: the product's Compile() (C++: medium/spectrum/geometry ->
kernel constants) against the oracle's independent numpy restatement of the
reference's code generators, table by table and bit by bit."""
import numpy as np
import pytest

from clsim_amd import converter as CV
from clsim_amd import synthetic as S
from tests import common

PAIRS = {
    "geoStringPosX": "str_x", "geoStringPosY": "str_y", "geoStringMinZ": "str_minz", "geoStringMaxZ": "str_maxz",
    "geoStringInStringSet": "str_set", "geoLayerNum": "set_nlayers", "geoLayerStartZ": "set_startz",
    "geoLayerHeight": "set_height", "geoLayerToOMNumIndexPerStringSet": "layer_to_om",
    "geoDomPosTemplatePositionsX_flat": "dom_tx", "geoDomPosTemplatePositionsY_flat": "dom_ty",
    "geoDomPosTemplatePositionsZ_flat": "dom_tz", "geoDomPosStringStartIndexInTemplateDomList": "dom_start",
    "geoDomPosStringMeanPosX": "dom_meanx", "geoDomPosStringMeanPosY": "dom_meany",
    "_generateWavelength_0distYValues": "gen0_yv", "_generateWavelength_0distYCumulativeValues": "gen0_ycum",
    "getWavelengthBias_data": "bias_data",
}
SCALARS = {
    "MEDIUM_LAYER_BOTTOM_POS": "layer_bottom", "MEDIUM_LAYER_THICKNESS": "layer_thickness", "OM_RADIUS": "om_radius",
    "GEO_STRING_MAX_RADIUS": "string_max_radius", "GEO_DOM_POS_MAX_ABS_X_MULTIPLIER_IN_TEMPLATE": "dom_mul_x",
    "GEO_DOM_POS_MAX_ABS_Y_MULTIPLIER_IN_TEMPLATE": "dom_mul_y", "liu_beta": "liu_beta", "hg_g": "hg_g", "hg_g2": "hg_g2",
    "mix_frac": "mix_frac", "mix_frac_rest": "mix_frac_rest", "PANCAKE_FACTOR": "pancake",
    "GEO_LAYER_STRINGSET_NUM": "num_sets", "GEO_LAYER_STRINGSET_MAX_NUM_LAYERS": "max_layers", "NUM_STRINGS": "num_strings",
}


@pytest.mark.parametrize("name", ["c1", "mie", "lea", "flasher"])
def test_compiled_tables_equal_oracle(name):
    cfg = common.config(name)
    T = common.oracle_tables(cfg)
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.Compile()
    pairs = dict(PAIRS)
    if cfg["med_o"]["len_mode"] == "icecube":
        pairs.update(aDust400="aDust400", deltaTau="deltaTau", b400="b400")
    else:
        pairs.update(absorptionLength="abs_const", scatteringLength="sca_const")
    if "tilt" in cfg["med_o"]:
        pairs.update(getTiltZShift_data_distancesFromOriginAlongTilt="tilt_dist", getTiltZShift_data_zCorrections="tilt_zcorr")
    for pn, on in pairs.items():
        p, o = conv.GetTable(pn), T.arrays[on].astype(np.float64)
        assert p.shape == o.shape and np.array_equal(p, o), pn
    sc = T.scalars()
    for pn, on in SCALARS.items():
        assert np.float32(conv.GetTable(pn)[0]) == np.float32(sc[on]), pn
    for k, c in enumerate(T.geo["cells"]):
        assert np.array_equal(conv.GetTable("geoCellIndex_%d" % k), c["index"].astype(np.float64))
        exp = np.array([c["nx"], c["ny"], c["width_x"], c["width_y"], c["start_x"], c["start_y"]], dtype=np.float64)
        assert np.array_equal(conv.GetTable("GEO_CELL_%d" % k), exp)
    if "tilt" in cfg["med_o"]:
        for pn, on in (("getTiltZShift_data_firstZCoord", "tilt_first_z"), ("getTiltZShift_data_zCoordSpacing", "tilt_dz"),
                       ("getTiltZShift_lnx", "tilt_lnx"), ("getTiltZShift_lny", "tilt_lny")):
            assert np.float32(conv.GetTable(pn)[0]) == np.float32(sc[on]), pn
    if "aniso" in cfg["med_o"]:
        o = list(sc["an_l"]) + list(sc["an_rl"]) + [sc["an_azx"], sc["an_azy"], sc["an_mazy"], sc["an_B2"]]
        assert np.array_equal(conv.GetTable("anisotropy").astype(np.float32), np.array(o, dtype=np.float32))
        for nm, key in (("transformDirectionPreScatter", "pre"), ("transformDirectionPostScatter", "post")):
            assert np.array_equal(conv.GetTable(nm).astype(np.float32), np.array(sc[key], dtype=np.float32))
    assert np.array_equal(conv.GetTable("stringIndexToStringID"), T.geo["string_index_to_id"].astype(np.float64))


def test_ic86_geometry_structure():
    """Facts about the synthetic IC86 geometry the rest of the suite relies on:
    two subdetectors sorted by name, 5160 DOM template entries (jitter => no shared
    templates), cell grids with at most one string per cell."""
    cfg = common.config("mie")
    T = common.oracle_tables(cfg)
    assert T.geo["subdetectors"] == ["DeepCore", "IceCube"]
    assert T.geo["num_strings"] == 86 and len(T.geo["dom_tx"]) == 5160
    for c in T.geo["cells"]:
        idx = c["index"][c["index"] != 0xFFFF]
        assert len(set(idx.tolist())) == len(idx)
    assert sorted(set(T.geo["string_index_to_id"].tolist())) == list(range(1, 87))


def test_invariant_divisors_are_proven():
    """The kernel replaces a/b by fma(fma(-b, a*r, a), r, a*r) only for divisors Compile() proved
    exact over all significands; for the benchmark configurations every divisor qualifies."""
    for name in ("c1", "mie", "lea"):
        cfg = common.config(name)
        conv = common.product_converter(cfg, 512, initialize=False)
        conv.Compile()
        ok = int(conv.GetTable("div_ok")[0])
        assert ok & 0b11110 == 0b11110
        assert ok & 32          # every length the generators' wavelengths can meet is bounded: dm::rcp_ in the layer walk
        assert ok & 64          # direction transforms (if any) are well conditioned: dm::rsqrt_near_ in the renormalisation
        if name == "lea":
            assert ok & 128     # anisotropy divisor bounded
        # mixed scattering with every proof in hand: the pooled kernel's instantiation without the run-time tests of those facts
        assert int(conv.GetTable("fast_variant")[0]) == 1
        if name != "c1":
            assert ok & 1
        assert all(int(v) == 3 for v in conv.GetTable("div_ok_cells"))


def test_division_proof_is_sound_on_a_sample():
    """Spot check of the proof itself in numpy float32 (fma emulated in float64, exact for these)."""
    rng = np.random.Generator(np.random.PCG64(5))
    for b in (np.float32(34.14329147), np.float32(10.0), np.float32(7.00153), np.float32(0.45)):
        r = np.float32(1.0) / b
        a = (rng.random(200000) * 4000 - 2000).astype(np.float32)
        q = a * r
        rem = (a.astype(np.float64) - b.astype(np.float64) * q.astype(np.float64)).astype(np.float32)
        q1 = (q.astype(np.float64) + rem.astype(np.float64) * np.float64(r)).astype(np.float32)
        assert np.array_equal(q1, a / b)


def test_text_file_geometry(tmp_path):
    """N1: I3CLSimSimpleGeometryTextFile ingestion ("string dom x y z" records, ID filters, one subdetector)."""
    from oracle import builders as B
    g = S.ic86_geometry()
    path = tmp_path / "geometry.txt"
    with open(path, "w") as f:
        f.write("0 1 0.0 0.0 0.0\n")                       # string 0: below the default filter
        for i in range(len(g["x"])):
            f.write("%d %d %.17g %.17g %.17g\n" % (g["string_ids"][i], g["dom_ids"][i], g["x"][i], g["y"][i], g["z"][i]))
        f.write("12 61 1.0 2.0 3.0\n")                     # DOM 61: above the default filter
    go = B.geometry_from_text_file(str(path), g["om_radius"])
    assert len(go["x"]) == 5160 and np.array_equal(go["x"], g["x"]) and np.array_equal(go["z"], g["z"])
    geo = B.build_geometry(go["string_ids"], go["dom_ids"], go["x"], go["y"], go["z"], go["subdetectors"], go["om_radius"])
    assert geo["subdetectors"] == ["default"] and len(geo["cells"]) == 1
    cfg = common.config("mie")
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.SetGeometry(CV.I3CLSimSimpleGeometry.from_text_file(g["om_radius"], str(path)))
    conv.Compile()
    assert np.array_equal(conv.GetTable("geoCellIndex_0"), geo["cells"][0]["index"].astype(np.float64))
    c = geo["cells"][0]
    assert np.array_equal(conv.GetTable("GEO_CELL_0"), np.array([c["nx"], c["ny"], c["width_x"], c["width_y"], c["start_x"], c["start_y"]], dtype=np.float64))
    for pn, on in (("geoStringPosX", "str_x"), ("geoLayerToOMNumIndexPerStringSet", "layer_to_om"), ("geoDomPosTemplatePositionsX_flat", "dom_tx"),
                   ("geoStringInStringSet", "str_set")):
        assert np.array_equal(conv.GetTable(pn), geo[on].astype(np.float64)), pn
    bad = CV.I3CLSimStepToPhotonConverterHIP(0)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="Could not open input file"):
        bad.SetGeometry(CV.I3CLSimSimpleGeometry.from_text_file(0.8, str(tmp_path / "missing.txt")))


def test_tabulated_medium_tables(tmp_path):
    """N1: per-layer FromTable lengths (16-bit storage) and tabulated refractive indices: the product's folded
    (layer, bin) values against the oracle evaluating the generated function's expression at the table points."""
    from oracle import builders as B
    from oracle import capi
    cfg = common.config("photonics_mie")
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.Compile()
    T = common.oracle_tables(cfg)
    m = cfg["med_o"]
    tb = m["table"]
    q = [B.quantize_table(r) for r in tb["abs"]]
    assert np.array_equal(conv.GetTable("getAbsorptionLength_data16").reshape(171, 30), np.array([x[2] for x in q]))
    lo_hi = conv.GetTable("getAbsorptionLength_smallest_largest").reshape(171, 2)
    assert np.array_equal(lo_hi[:, 0], np.array([x[0] for x in q], dtype=np.float64))
    assert np.array_equal(lo_hi[:, 1], np.array([x[1] for x in q], dtype=np.float64))
    assert conv.GetTable("getAbsorptionLength_data16").max() == 65535 and conv.GetTable("getScatteringLength_data16").min() == 0
    # at a table point the fraction is 0 or 1, so the generated function returns the de-quantised entry itself
    wl = np.array([B.float_literal(tb["start"]) + np.float32(k) * B.float_literal(tb["step"]) for k in range(30)], dtype=np.float32)
    wl_mid = (wl[:-1] + np.float32(0.25) * (wl[1:] - wl[:-1])).astype(np.float32)
    va = conv.GetTable("getAbsorptionLength_values").reshape(171, 30).astype(np.float32)
    vs = conv.GetTable("getScatteringLength_values").reshape(171, 30).astype(np.float32)
    for layer in (0, 17, 85, 170):
        oa, os_ = capi.eval_medium(T, 0, wl_mid, layer), capi.eval_medium(T, 1, wl_mid, layer)
        q_ = (wl_mid - B.float_literal(tb["start"])) / B.float_literal(tb["step"])
        frac = (q_ - np.trunc(q_)).astype(np.float32)
        k = np.trunc(q_).astype(int)
        assert np.array_equal(oa, va[layer, k] + (va[layer, k + 1] - va[layer, k]) * frac)
        assert np.array_equal(os_, vs[layer, k] + (vs[layer, k + 1] - vs[layer, k]) * frac)
        # quantisation error bound: (largest - smallest) / 65535 per entry
        assert np.all(np.abs(va[layer] - tb["abs"][layer]) <= (tb["abs"][layer].max() - tb["abs"][layer].min()) / 65535.0 * 1.01 + 1e-6)
    assert np.array_equal(conv.GetTable("getPhaseRefIndex_func0_data"), B.float_literals(m["phase_table"]["values"]).astype(np.float64))
    assert np.array_equal(conv.GetTable("getGroupRefIndex_func0_data"), B.float_literals(m["group_table"]["values"]).astype(np.float64))
    assert conv.GetTable("MEDIUM_MIN_WLEN")[0] == B.float_literal(305e-9) and conv.GetTable("MEDIUM_MAX_WLEN")[0] == B.float_literal(m["max_wlen"])
    # the wavelength generator is built from the tabulated phase index on both sides
    bias = B.icecube_dom_acceptance()
    yv, ycum = B.interp_dist_tables(B.cherenkov_wlen_generator(bias, m))
    assert np.array_equal(conv.GetTable("_generateWavelength_0distYValues"), yv.astype(np.float64))
    assert np.array_equal(conv.GetTable("_generateWavelength_0distYCumulativeValues"), ycum.astype(np.float64))
    # error paths of the table file reader
    bad = tmp_path / "bad.txt"
    bad.write_text("NLAYER 1\nLAYER 0 10\nABS 1 1\n")
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="NWVL"):
        CV.MakeIceCubeMediumPropertiesPhotonics(str(bad))
    bad.write_text("NLAYER 2\nNWVL 2 300 10\nLAYER 0 10\nABS 1 1\nSCAT 1 1\nCOS .9 .9\nN_GROUP 1.3 1.3\nN_PHASE 1.3 1.3\n")
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="Expected 2\\*6"):
        CV.MakeIceCubeMediumPropertiesPhotonics(str(bad))


def test_string_proximity_map_is_a_lower_bound():
    """The DOM search is skipped for steps shorter than the map's entry (prop_device.hip.h: free_flight_bound): bits 0-7 of a
    word must never exceed the true xy distance from any point of its cell to the nearest DOM sphere.  Bits 16-31 name a string,
    bits 8-15 bound the distance to every OTHER string's DOMs (segment_misses_string relies on it: a step shorter than that can
    only touch the named string, and only within `reach` of its axis)."""
    for name in ("mie", "c1", "mie_regular"):
        cfg = common.config(name)
        conv = common.product_converter(cfg, 512, initialize=False)
        conv.Compile()
        n, x0, y0, inv_cell, reach, reach_f = conv.GetTable("STRING_PROXIMITY_GRID")
        n = int(n)
        words = conv.GetTable("string_proximity_map").astype(np.uint64).astype(np.uint32).reshape(n, n)
        g = cfg["geom"]
        rng = np.random.Generator(np.random.PCG64(7))
        lo = min(g["x"].min(), g["y"].min()) - 200.0
        hi = max(g["x"].max(), g["y"].max()) + 200.0
        pts = rng.uniform(lo, hi, size=(20000, 2))
        near = rng.integers(0, len(g["x"]), size=10000)                 # and points close to DOMs
        pts = np.concatenate([pts, np.stack([g["x"][near], g["y"][near]], axis=1) + rng.normal(0, 6.0, size=(10000, 2))])
        xf, yf = pts[:, 0].astype(np.float32), pts[:, 1].astype(np.float32)
        ix = np.clip(((xf - np.float32(x0)) * np.float32(inv_cell)).astype(np.int32), 0, n - 1)   # the kernel's arithmetic
        iy = np.clip(((yf - np.float32(y0)) * np.float32(inv_cell)).astype(np.int32), 0, n - 1)
        w = words[iy, ix]
        bound, second, named = (w & 0xff) * 0.25, ((w >> 8) & 0xff) * 0.25, (w >> 16).astype(np.int64)
        # the strings as the converter numbers them (sorted string IDs) and every DOM's distance in xy
        ids = np.unique(g["string_ids"])
        string_of_dom = np.searchsorted(ids, g["string_ids"])
        true, other = np.empty(len(pts)), np.empty(len(pts))
        for k in range(0, len(pts), 2000):
            sl = slice(k, k + 2000)
            d = np.hypot(xf[sl, None].astype(np.float64) - g["x"][None, :], yf[sl, None].astype(np.float64) - g["y"][None, :]) - g["om_radius"]
            true[sl] = d.min(axis=1)
            other[sl] = np.where(string_of_dom[None, :] == named[sl, None], np.inf, d).min(axis=1)
        assert np.all(bound <= np.maximum(true, 0.0)), name
        assert (bound > 0).mean() > 0.5          # and it is useful: most of the volume is free flight
        assert np.all(named < len(ids))          # (every cell of these detectors names a string)
        # DOMs of strings other than the named one: never closer than the second bound
        if len(ids) > 1:
            assert np.all(second <= np.maximum(other, 0.0)), name
            assert np.all(second >= bound) and (second > 20.0).mean() > 0.5       # useful: the next string is 125 m away
        # every DOM of the named string lies within `reach` - radius of the axis the kernel reads (string mean position)
        axis = np.array([[g["x"][string_of_dom == s].mean(), g["y"][string_of_dom == s].mean()] for s in range(len(ids))])
        off = np.hypot(g["x"] - axis[string_of_dom, 0], g["y"] - axis[string_of_dom, 1])
        assert off.max() + g["om_radius"] < reach_f and reach_f >= reach


def test_dom_proximity_map_is_a_lower_bound():
    """DOM proximity map (second / third level of the search filter, prop_device.hip.h: dom_search_needed): a cell names its
    nearest DOM and stores a bound for all others.  The kernel skips a search only when the step is shorter than the stored
    bound AND the segment stays farther from the named DOM than its radius; both conditions together imply that the step is
    shorter than min(distance to the named DOM's sphere, stored bound).  That minimum must therefore never exceed the true 3D
    distance from the point to the surface of the nearest DOM sphere -- and should be close to it.  ("mie_regular": the
    strings share DOM position templates; round 2 numbered the maps' DOMs by template entry and lost all but 120 of them.)"""
    for name in ("mie", "c1", "mie_regular"):
        cfg = common.config(name)
        conv = common.product_converter(cfg, 512, initialize=False)
        conv.Compile()
        nx, ny, nz, x0, y0, z0, inv_cell, radius = conv.GetTable("DOM_PROXIMITY_GRID")
        nx, ny, nz = int(nx), int(ny), int(nz)
        words = conv.GetTable("dom_proximity_map").astype(np.uint32).reshape(nx, ny, nz)        # z runs fastest
        centres = conv.GetTable("dom_centres").reshape(-1, 4)[:, :3]
        g = cfg["geom"]
        doms = np.stack([g["x"], g["y"], g["z"]], axis=1)
        assert len(centres) == len(doms)
        # the kernel's DOM positions are the geometry's, up to the int16 template quantisation
        assert np.abs(np.sort(centres, axis=0) - np.sort(doms, axis=0)).max() < 0.02
        rng = np.random.Generator(np.random.PCG64(11))
        lo, hi = doms.min(axis=0) - 150.0, doms.max(axis=0) + 150.0
        pts = rng.uniform(lo, hi, size=(30000, 3))
        near = rng.integers(0, len(doms), size=30000)                   # and points close to DOMs, at every scale
        pts = np.concatenate([pts, doms[near] + rng.normal(0, 1.0, size=(30000, 3)) * rng.choice([0.5, 3.0, 12.0], size=(30000, 1))])
        pf = pts.astype(np.float32)
        idx = []
        for k, (o, n) in enumerate(((x0, nx), (y0, ny), (z0, nz))):
            idx.append(np.clip(((pf[:, k] - np.float32(o)) * np.float32(inv_cell)).astype(np.int32), 0, n - 1))   # the kernel's arithmetic
        w = words[idx[0], idx[1], idx[2]]
        ident = (w & 0xffff).astype(np.int64)
        bound = ((w >> 16) & 0xff) * 0.25
        named = ident != 0xffff
        assert ident[named].max() < len(doms)
        exact = np.sqrt(((pf[named].astype(np.float64) - centres[ident[named]]) ** 2).sum(axis=1)) * 0.99999 - radius
        bound[named] = np.minimum(bound[named], exact)
        true = np.full(len(pts), np.inf)
        for k in range(0, len(doms), 128):
            d = np.sqrt(((pf[:, None, :].astype(np.float64) - centres[None, k:k + 128, :]) ** 2).sum(axis=2))
            true = np.minimum(true, d.min(axis=1))
        true -= g["om_radius"]
        assert np.all(bound <= np.maximum(true, 0.0) + 1e-9), name
        if name == "mie":
            close = true < 30.0
            assert np.median((true - bound)[close]) < 0.1         # near a DOM the bound is the exact distance (minus the safety)
            inside = np.all((pts[:30000] > doms.min(axis=0)) & (pts[:30000] < doms.max(axis=0)), axis=1)
            assert (bound[:30000][inside] > 3.0).mean() > 0.9     # and it is useful: a few metres of free flight nearly everywhere


def test_a_detector_beyond_the_lds_budget_of_seven_workgroups_compiles():
    """576 strings: the table image (string records 18 KB + ice + cells) no longer fits seven workgroups per CU; Compile()
    accepts it (the launchers run fewer workgroups per CU) and the geometry tables still equal the independent builder's."""
    from clsim_amd import synthetic as S
    cfg = common.config("mie")
    cfg["geom"] = S.large_detector_geometry()
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.Compile()
    assert conv.GetTable("lds_bytes_per_workgroup")[0] > 160 * 1024 / 7
    T = common.oracle_tables(cfg)
    assert int(conv.GetTable("NUM_STRINGS")[0]) == 576
    assert np.array_equal(conv.GetTable("geoStringPosX").astype(np.float32), np.asarray(T.geo["str_x"], dtype=np.float32))


def _check_named_records(cfg):
    """dom_named (kparams.h) against an independent reading of the oracle's geometry tables: for every DOM the string and
    DOM number, the rectangle of cells of the string's subdetector that name the string, the z layers that name the DOM"""
    T = common.oracle_tables(cfg)
    geo = T.geo
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.Compile()
    named = conv.GetTable("dom_named").astype(np.uint64).reshape(-1, 4)
    centres = conv.GetTable("dom_centres").reshape(-1, 4)
    assert len(named) == len(centres)
    ns = geo["num_strings"]
    # DOM records follow the strings in index order, dom_start.. (one record per DOM of every string)
    counts = [len(ids) for ids in geo["dom_index_to_id"]]
    starts = np.concatenate([[0], np.cumsum(counts)])
    assert starts[-1] == len(named)
    nameable = 0
    straddling = 0
    for s in range(ns):
        cells_of = []
        for k, c in enumerate(geo["cells"]):
            idx = np.asarray(c["index"]).reshape(c["ny"], c["nx"])
            ys, xs = np.nonzero(idx == s)
            cells_of += [(k, int(x), int(y)) for x, y in zip(xs, ys)]
        sds = {k for k, _, _ in cells_of}
        xs = [x for _, x, _ in cells_of]; ys = [y for _, _, y in cells_of]
        whole = len(sds) == 1 and len(cells_of) == (max(xs) - min(xs) + 1) * (max(ys) - min(ys) + 1)
        straddling += len(cells_of) > 1
        st = int(geo["str_set"][s])
        table = np.asarray(geo["layer_to_om"])[st * geo["max_layers"]: st * geo["max_layers"] + int(geo["set_nlayers"][st])]
        for d in range(counts[s]):
            rec = named[starts[s] + d]
            layers = np.nonzero(table == d)[0]
            ok = whole and len(layers) > 0 and layers[-1] - layers[0] + 1 == len(layers)
            if not ok:
                assert rec[0] == 0xffffffff
                continue
            nameable += 1
            assert rec[0] == (s | (d << 16))
            assert rec[1] == (min(xs) | (min(ys) << 12) | (next(iter(sds)) << 24)) and rec[3] == (max(xs) | (max(ys) << 12))
            assert rec[2] == (int(layers[0]) | (int(layers[-1]) << 16))
            # and the centre is the position the search reconstructs for (string, DOM)
            i = int(geo["dom_start"][s]) + d
            x = np.float32(np.float32(geo["dom_tx"][i]) * np.float32(geo["dom_mul_x"]) + np.float32(geo["dom_meanx"][s]))
            assert np.float32(centres[starts[s] + d][0]) == x and np.float32(centres[starts[s] + d][2]) == np.float32(geo["dom_tz"][i])
    return nameable, len(named), straddling


def test_named_dom_records():
    nameable, total, _ = _check_named_records(common.config("mie"))
    assert nameable == total == 5160


def test_named_dom_records_when_strings_straddle_cell_borders():
    """strings whose bounding square overlaps several cells of the grid (a string belongs to every cell it touches,
    GeometrySource.cxx:135-271): the record holds the rectangle"""
    rng = np.random.Generator(np.random.PCG64(5))
    g = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in common.config("mie")["geom"].items()}
    # lean the strings: DOM positions drift by up to +-12 m in x and y along a string, so the bounding squares grow
    for sid in np.unique(g["string_ids"]):
        m = g["string_ids"] == sid
        z = g["z"][m]
        t = (z - z.min()) / max(z.max() - z.min(), 1.0) - 0.5
        g["x"][m] += 24.0 * rng.uniform(-1, 1) * t
        g["y"][m] += 24.0 * rng.uniform(-1, 1) * t
    cfg = dict(common.config("mie"), geom=g)
    nameable, total, straddling = _check_named_records(cfg)
    assert straddling > 0, "the test geometry should have strings in several cells"
    assert nameable > 0.9 * total
