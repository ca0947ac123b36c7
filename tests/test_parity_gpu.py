"""GPU parity: the HIP propagator (through the C ABI) against the CPU oracle on
identical steps and RNG streams.  Bar: the sorted multiset of 80-byte photon
records is BIT-IDENTICAL (hit count, string/DOM IDs, scatter counts and every
float), and so are the RNG state words left behind."""
import numpy as np
import pytest

from clsim_amd import synthetic as S
from oracle import capi
from tests import common

pytestmark = pytest.mark.gpu


def run_both(name, n_steps, max_items=None, seed=3, threads=8):
    cfg = common.config(name)
    steps = common.steps_for(cfg, n_steps, seed=seed)
    n = len(steps)
    max_items = max_items or n
    x, a = common.streams(max_items)
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=threads)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    conv = common.product_converter(cfg, max_items)
    conv.EnqueueSteps(steps, 77)
    ident, ph_p = conv.GetConversionResult()
    assert ident == 77
    x_p = conv.GetRNGState(n)
    return steps, (ph_o, cnt_o, x_o), (ph_p, x_p), conv


@pytest.mark.parametrize("name,n_steps", [("c1", 1000), ("mie", 4096), ("lea", 4096), ("flasher", 2048),
                                          ("photonics_mie", 4096), ("photonics_wham", 2048),
                                          # DOMs exactly on the string axes: 86 strings share two position templates
                                          ("mie_regular", 4096), ("flasher_regular", 2048),
                                          # second generator = the 405 nm LED's measured spectrum, a table with its own wavelengths
                                          ("flasher_led405", 2048)])
def test_hit_multiset_bit_exact(name, n_steps):
    steps, (ph_o, cnt_o, x_o), (ph_p, x_p), conv = run_both(name, n_steps)
    assert cnt_o > 10, "workload too small to be a test"
    assert len(ph_p) == cnt_o
    so, sp = common.sort_photons(ph_o), common.sort_photons(ph_p)
    assert np.array_equal(so["stringID"], sp["stringID"]) and np.array_equal(so["omID"], sp["omID"])
    assert so.tobytes() == sp.tobytes()
    assert np.array_equal(x_o, x_p)
    st = conv.GetStatistics()
    assert st["TotalNumPhotonsGenerated"] == float(steps["num"].sum())
    assert st["TotalNumPhotonsAtDOMs"] == float(cnt_o)
    assert st["NumKernelCalls"] == 1.0


def test_streams_persist_across_bunches():
    """RNG stream i belongs to step slot i and carries over to the next bunch
    (propagation_kernel.c.cl:458-461, 911-912)."""
    cfg = common.config("mie")
    steps = common.steps_for(cfg, 1024, seed=9)
    x, a = common.streams(1024)
    T = common.oracle_tables(cfg)
    conv = common.product_converter(cfg, 1024)
    xo = x
    for bunch in range(3):
        ph_o, cnt_o, xo, _ = capi.propagate(T, steps, xo, a, threads=8)
        conv.EnqueueSteps(steps, bunch)
        ident, ph_p = conv.GetConversionResult()
        assert ident == bunch
        ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
        assert common.sort_photons(ph_o).tobytes() == common.sort_photons(ph_p).tobytes()
    assert np.array_equal(conv.GetRNGState(1024), xo)


def test_output_overflow_truncates_like_reference():
    """The hit counter keeps counting past the buffer (propagation_kernel.c.cl:329-330);
    the host logs and truncates (OpenCL.cxx:1027-1032).  A flasher 1 m from a DOM
    overflows the reference-sized buffer of 10 x maxNumWorkitems records."""
    cfg = common.config("flasher")
    g = cfg["geom"]
    k = 30 * 60 + 29
    from clsim_amd import synthetic as S
    steps = S.flasher_steps(512, seed=3, position=(g["x"][k] + 1.0, g["y"][k], g["z"][k]), pad_to=256)
    x, a = common.streams(len(steps))
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, _, _ = capi.propagate(T, steps, x, a, threads=8)
    assert cnt_o > 10 * len(steps)
    conv = common.product_converter(cfg, len(steps))
    conv.EnqueueSteps(steps, 5)
    ident, ph_p = conv.GetConversionResult()
    assert len(ph_p) == 10 * len(steps)
    assert conv.GetStatistics()["TotalNumPhotonsAtDOMs"] == float(10 * len(steps))
    # every stored record is one of the oracle's records
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    raw_o = ph_o.tobytes()
    have = set(raw_o[i:i + 80] for i in range(0, len(raw_o), 80))
    raw = ph_p.tobytes()
    assert all(raw[i:i + 80] in have for i in range(0, len(raw), 80))


def _run_custom(med_o, med_p, geom, gens_o, gens_p, steps, pancake=5.0, max_items=None):
    """Oracle vs product for an ad-hoc configuration (exercises the other kernel variants)."""
    from clsim_amd import converter as CV
    from oracle import builders as B
    n = len(steps)
    x, a = common.streams(max_items or n)
    geo = B.build_geometry(geom["string_ids"], geom["dom_ids"], geom["x"], geom["y"], geom["z"], geom["subdetectors"], geom["om_radius"])
    bias_o = B.icecube_dom_acceptance()
    T = capi.make_tables(med_o, geo, gens_o(bias_o), bias_o, pancake=pancake)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    bias_p = CV.GetIceCubeDOMAcceptance()
    conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_dict(geom), med_p, bias_p, gens_p(bias_p), pancakeFactor=pancake,
                            approximateNumberOfWorkItems=n, streams=(x, a))
    conv.EnqueueSteps(steps, 3)
    ident, ph_p = conv.GetConversionResult()
    assert cnt_o > 5 and len(ph_p) == cnt_o
    assert common.sort_photons(ph_o).tobytes() == common.sort_photons(ph_p).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)


def test_range_restricted_divides_fall_back_to_the_ieee_sequence():
    """The exact divide of the layer walk and of the rotation (detmath.hip.h: div_near_) serves numerators of magnitude
    >= 2^-40 only; a wave that holds anything else takes the IEEE branch.  Two kinds of steps force those branches:
    photons born EXACTLY on a layer boundary and heading down (height above the boundary = +0), and steps of a particle
    below the Cherenkov threshold (cos = min(1, 1/(beta n)) = 1, so the cone's sine and both numerators of the rotation
    are zero).  Output and RNG states must equal the oracle's, which divides the IEEE way throughout."""
    import os
    from clsim_amd import converter as CV
    from oracle import builders as B
    ice = os.path.join(common.ICE, "spice_mie")
    med_o = B.load_ppc_ice(ice, use_tilt_if_available=False)
    med_p = CV.MakeIceCubeMediumProperties(iceDataDirectory=ice, useTiltIfAvailable=False)
    geom = S.ic86_geometry()
    steps = common.steps_for(common.config("mie"), 3072, seed=77)
    bottom, height = np.float32(med_o["layers_z_start"]), np.float32(med_o["layers_height"])
    k = np.arange(1024) % 120 + 25
    on_boundary = (k.astype(np.float32) * height) + bottom                      # mediumLayerBoundary, c.cl:78-81, in float
    steps["z"][:1024] = on_boundary
    steps["length"][:1024] = 0.0                                                # photons are born at the step's start
    steps["theta"][:1024] = np.float32(np.pi)                                   # heading down
    # the layer the kernel finds for such a point must be the one whose lower boundary it sits on, at least for many of them
    layer = ((on_boundary - bottom) / height).astype(np.int32)
    assert np.count_nonzero((layer.astype(np.float32) * height) + bottom == on_boundary) > 500
    steps["beta"][1024:2048] = 0.5
    x, a = common.streams(len(steps))
    geo = B.build_geometry(geom["string_ids"], geom["dom_ids"], geom["x"], geom["y"], geom["z"], geom["subdetectors"], geom["om_radius"])
    bias_o = B.icecube_dom_acceptance()
    T = capi.make_tables(med_o, geo, [B.cherenkov_wlen_generator(bias_o, med_o)], bias_o, pancake=5.0)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    bias_p = CV.GetIceCubeDOMAcceptance()
    conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_dict(geom), med_p, bias_p, [CV.makeCherenkovWavelengthGenerator(bias_p, med_p)],
                            pancakeFactor=5.0, approximateNumberOfWorkItems=len(steps), streams=(x, a))
    conv.EnqueueSteps(steps, 5)
    _, ph_p = conv.GetConversionResult()
    assert cnt_o > 100 and len(ph_p) == cnt_o
    assert common.sort_photons(ph_o).tobytes() == common.sort_photons(ph_p).tobytes()
    assert np.array_equal(conv.GetRNGState(len(steps)), x_o)
    for part in (slice(0, 1024), slice(1024, 2048)):                            # both kinds of steps did produce detected photons
        ids = set(steps["id"][part].tolist())
        assert any(int(i) in ids for i in ph_p["id"])


@pytest.mark.parametrize("kind", ["no_pancake", "no_tilt", "single_icecube_layer", "hg_only", "liu_only", "lea_no_tilt", "flasher_c1",
                                  "table_float", "table_with_tilt_and_aniso", "no_dispersion_no_bias", "group_velocity_from_dispersion"])
def test_other_kernel_variants(kind):
    """Variants of the generated program: PANCAKE_FACTOR undefined (pancake = 1), getTiltZShift_IS_CONSTANT with
    layered ice (carried layer index), a one-layer IceCube medium (un-optimised per-function form, SURVEY 9.7 ii),
    pure HG / pure Liu scattering, anisotropy without tilt, flasher generator with constant-length medium."""
    import ctypes as C
    import os
    from clsim_amd import _lib
    from clsim_amd import converter as CV
    from clsim_amd import synthetic as S
    from oracle import builders as B
    cher_o = lambda med: (lambda bias: [B.cherenkov_wlen_generator(bias, med)])
    ice_dir = lambda m: os.path.join(common.ICE, m)

    def product_medium(desc_edit=None, directory=None, tilt=True):
        med = CV.MakeIceCubeMediumProperties(iceDataDirectory=directory, useTiltIfAvailable=tilt)
        if desc_edit is None:
            return med
        d = _lib.MediumDesc()
        assert _lib.load().clsimhip_medium_describe(med._h, C.byref(d)) == 0
        keep = desc_edit(d)
        h = C.c_void_p()
        assert _lib.load().clsimhip_medium_create(C.byref(d), C.byref(h)) == 0
        return CV.I3CLSimMediumProperties(h, keep=(med, keep))

    geom = S.ic86_geometry()
    steps = common.steps_for(common.config("mie"), 2048, seed=31)
    if kind == "no_pancake":
        med_o = B.load_ppc_ice(ice_dir("spice_mie"))
        med_p = product_medium(directory=ice_dir("spice_mie"))
        _run_custom(med_o, med_p, geom, cher_o(med_o), lambda b: [CV.makeCherenkovWavelengthGenerator(b, med_p)], steps, pancake=1.0)
    elif kind == "no_tilt":
        med_o = B.load_ppc_ice(ice_dir("spice_mie"), use_tilt_if_available=False)
        med_p = product_medium(directory=ice_dir("spice_mie"), tilt=False)
        _run_custom(med_o, med_p, geom, cher_o(med_o), lambda b: [CV.makeCherenkovWavelengthGenerator(b, med_p)], steps)
    elif kind == "lea_no_tilt":
        med_o = B.load_ppc_ice(ice_dir("spice_lea"), use_tilt_if_available=False)
        med_p = product_medium(directory=ice_dir("spice_lea"), tilt=False)
        _run_custom(med_o, med_p, geom, cher_o(med_o), lambda b: [CV.makeCherenkovWavelengthGenerator(b, med_p)], steps)
    elif kind == "group_velocity_from_dispersion":
        # no group refractive index override: getGroupVelocity from the phase index and its derivative
        # (I3CLSimHelperGenerateMediumPropertiesSource.cxx:274-300); every arrival time moves by up to 1 %
        cfg = common.config("lea_dispersion")
        med_o, med_p = cfg["med_o"], cfg["med_p"]
        _run_custom(med_o, med_p, geom, cher_o(med_o), lambda b: [CV.makeCherenkovWavelengthGenerator(b, med_p)], steps)
        x, a = common.streams(len(steps))
        with_override, _, _, _ = capi.propagate(common.oracle_tables(common.config("lea")), steps, x, a, threads=8)
        from_dispersion, _, _, _ = capi.propagate(common.oracle_tables(cfg), steps, x, a, threads=8)
        assert len(with_override) == len(from_dispersion) > 5                    # the same photons at the same DOMs ...
        order = lambda ph: ph[np.lexsort((ph["wavelength"], ph["numScatters"], ph["omID"], ph["stringID"], ph["id"]))]
        with_override, from_dispersion = order(with_override), order(from_dispersion)    # (the oracle's threads store in any order)
        for field in ("id", "stringID", "omID", "numScatters", "wavelength", "weight"):
            assert np.array_equal(with_override[field], from_dispersion[field]), field
        rel = from_dispersion["groupVelocity"] / with_override["groupVelocity"] - 1.0
        assert np.all(rel != 0.0) and np.abs(rel).max() < 0.011                  # ... at other times
        assert np.any(with_override["t"] != from_dispersion["t"])
    elif kind == "no_dispersion_no_bias":
        # generateCherenkovPhotonsWithoutDispersion with a constant bias of 1 (ModuleHelper.cxx:265-275):
        # I3CLSimRandomValueWlenCherenkovNoDispersion over the medium's wavelength range, constant wavelength bias
        med_o = B.load_ppc_ice(ice_dir("spice_mie"))
        med_p = product_medium(directory=ice_dir("spice_mie"))
        n = len(steps)
        x, a = common.streams(n)
        geo = B.build_geometry(geom["string_ids"], geom["dom_ids"], geom["x"], geom["y"], geom["z"], geom["subdetectors"], geom["om_radius"])
        T = capi.make_tables(med_o, geo, [dict(kind="nodispersion", **{"from": med_o["min_wlen"], "to": med_o["max_wlen"]})],
                             dict(kind="const", value=1.0), pancake=5.0)
        ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
        ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
        conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_dict(geom), med_p, CV.I3CLSimFunctionConstant(1.0),
                                [CV.I3CLSimRandomValueWlenCherenkovNoDispersion(med_o["min_wlen"], med_o["max_wlen"])], pancakeFactor=5.0,
                                approximateNumberOfWorkItems=n, streams=(x, a))
        conv.EnqueueSteps(steps, 3)
        _, ph_p = conv.GetConversionResult()
        assert cnt_o > 5 and len(ph_p) == cnt_o
        assert common.sort_photons(ph_o).tobytes() == common.sort_photons(ph_p).tobytes()
        assert np.array_equal(conv.GetRNGState(n), x_o)
        assert float(ph_p["wavelength"].min()) >= 264e-9 and float(ph_p["wavelength"].max()) <= 676e-9 and np.all(ph_p["weight"] == 1.0)
    elif kind in ("table_float", "table_with_tilt_and_aniso"):
        # per-layer FromTable lengths stored as floats (storeDataAsHalfPrecision=False); and tabulated lengths under
        # the tilt / anisotropy / Mixed scattering objects of SPICE-Lea (no reference loader builds this mix, the classes allow it)
        path = common.PHOTONICS["photonics_mie"]
        med_o = B.load_photonics_ice(path)
        lea_o = B.load_ppc_ice(ice_dir("spice_lea"))
        lea_p = CV.MakeIceCubeMediumProperties(iceDataDirectory=ice_dir("spice_lea"))
        lea_d = _lib.MediumDesc()
        assert _lib.load().clsimhip_medium_describe(lea_p._h, C.byref(lea_d)) == 0
        if kind == "table_float":
            med_o["table"]["store16"] = False
        else:
            for key in ("aniso", "pre", "post", "tilt"):
                med_o[key] = lea_o[key]
            med_o["scat"] = lea_o["scat"]
        tab = CV.MakeIceCubeMediumPropertiesPhotonics(path)
        d = _lib.MediumDesc()
        assert _lib.load().clsimhip_medium_describe(tab._h, C.byref(d)) == 0
        if kind == "table_float":
            d.table_store_as_16bit = 0
        else:
            for f in ("scatter_kind", "liu_fraction", "mean_cosine", "has_anisotropy", "aniso_azimuth", "aniso_k1", "aniso_k2",
                      "has_pre_transform", "pre_renormalize", "pre_matrix", "has_post_transform", "post_renormalize", "post_matrix",
                      "has_tilt", "tilt_num_distances", "tilt_num_z", "tilt_distances", "tilt_z_coordinates", "tilt_z_corrections",
                      "tilt_azimuth"):
                setattr(d, f, getattr(lea_d, f))
        h = C.c_void_p()
        assert _lib.load().clsimhip_medium_create(C.byref(d), C.byref(h)) == 0
        med_p = CV.I3CLSimMediumProperties(h, keep=(tab, lea_p))
        _run_custom(med_o, med_p, geom, cher_o(med_o), lambda b: [CV.makeCherenkovWavelengthGenerator(b, med_p)], steps)
    elif kind in ("hg_only", "liu_only"):
        med_o = B.load_ppc_ice(ice_dir("spice_mie"))
        med_o["scat"]["kind"] = "hg" if kind == "hg_only" else "liu"

        def edit(d):
            d.scatter_kind = 0 if kind == "hg_only" else 1
        med_p = product_medium(edit, ice_dir("spice_mie"))
        _run_custom(med_o, med_p, geom, cher_o(med_o), lambda b: [CV.makeCherenkovWavelengthGenerator(b, med_p)], steps)
    elif kind == "single_icecube_layer":
        full = B.load_ppc_ice(ice_dir("spice_mie"), use_tilt_if_available=False)
        med_o = dict(full, num_layers=1, layers_z_start=-1000.0, layers_height=2000.0,
                     aDust400=full["aDust400"][85:86], deltaTau=full["deltaTau"][85:86], b400=full["b400"][85:86])
        arrays = [np.ascontiguousarray(med_o[k], dtype=np.float64) for k in ("aDust400", "deltaTau", "b400")]

        def edit(d):
            d.num_layers = 1; d.layers_z_start = -1000.0; d.layers_height = 2000.0
            d.a_dust400, d.delta_tau, d.b400 = (a.ctypes.data_as(_lib.DP) for a in arrays)
            return arrays
        med_p = product_medium(edit, ice_dir("spice_mie"), tilt=False)
        _run_custom(med_o, med_p, geom, cher_o(med_o), lambda b: [CV.makeCherenkovWavelengthGenerator(b, med_p)], steps)
    else:   # flasher_c1: constant lengths + second (constant) generator, single string
        g1 = S.single_string_geometry()
        med_o = B.homogeneous_medium()
        med_p = CV.MakeHomogeneousMediumProperties()
        st = S.flasher_steps(1024, seed=4, position=(30.0, 20.0, 100.0), pad_to=256)
        st["sourceType"][:100] = 0          # a few Cherenkov steps in the same bunch
        st["length"][:100] = 0.001
        st["num"][100:120] = 0              # and a few empty ones
        _run_custom(med_o, med_p, g1, lambda bias: [B.cherenkov_wlen_generator(bias, med_o), dict(kind="const", value=405e-9)],
                    lambda b: [CV.makeCherenkovWavelengthGenerator(b, med_p), CV.I3CLSimRandomValueConstant(405e-9)], st)


@pytest.mark.gpu
def test_text_file_geometry_single_subdetector(tmp_path):
    """N1: geometry read from an I3CLSimSimpleGeometryTextFile-format file: all 86 strings in the one
    subdetector "default", i.e. a single fine cell grid instead of the IceCube / DeepCore pair."""
    from clsim_amd import converter as CV
    from clsim_amd import synthetic as S
    from oracle import builders as B
    g = S.ic86_geometry()
    path = tmp_path / "geo.txt"
    with open(path, "w") as f:
        for i in range(len(g["x"])):
            f.write("%d %d %.17g %.17g %.17g\n" % (g["string_ids"][i], g["dom_ids"][i], g["x"][i], g["y"][i], g["z"][i]))
    cfg = common.config("mie")
    steps = common.steps_for(cfg, 2048, seed=8)
    n = len(steps)
    x, a = common.streams(n)
    go = B.geometry_from_text_file(str(path), g["om_radius"])
    geo = B.build_geometry(go["string_ids"], go["dom_ids"], go["x"], go["y"], go["z"], go["subdetectors"], go["om_radius"])
    bias_o = B.icecube_dom_acceptance()
    T = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias_o, cfg["med_o"])], bias_o, pancake=5.0)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    bias_p = CV.GetIceCubeDOMAcceptance()
    conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_text_file(g["om_radius"], str(path)), cfg["med_p"], bias_p,
                            [CV.makeCherenkovWavelengthGenerator(bias_p, cfg["med_p"])], pancakeFactor=5.0,
                            approximateNumberOfWorkItems=n, streams=(x, a))
    conv.EnqueueSteps(steps, 1)
    _, ph_p = conv.GetConversionResult()
    assert cnt_o > 50 and len(ph_p) == cnt_o
    assert common.sort_photons(ph_o).tobytes() == common.sort_photons(ph_p).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)


def test_fixed_number_of_absorption_lengths():
    """PROPAGATE_FOR_FIXED_NUMBER_OF_ABSORPTION_LENGTHS (propagation_kernel.c.cl:582-588, OpenCL.cxx:425-431): every
    photon gets the same absorption budget and its creation draws one random number less."""
    from clsim_amd import converter as CV
    from oracle import builders as B
    cfg = common.config("mie")
    steps = common.steps_for(cfg, 2048, seed=17)
    n = len(steps)
    x, a = common.streams(n)
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    bias_o = B.icecube_dom_acceptance()
    T = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias_o, cfg["med_o"])], bias_o, pancake=5.0, fixed_abs_lengths=1.75)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    bias_p = CV.GetIceCubeDOMAcceptance()
    conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_dict(g), cfg["med_p"], bias_p, [CV.makeCherenkovWavelengthGenerator(bias_p, cfg["med_p"])],
                            pancakeFactor=5.0, fixedNumberOfAbsorptionLengths=1.75, approximateNumberOfWorkItems=n, streams=(x, a))
    conv.EnqueueSteps(steps, 5)
    _, ph_p = conv.GetConversionResult()
    assert cnt_o > 100 and len(ph_p) == cnt_o
    assert common.sort_photons(ph_o).tobytes() == common.sort_photons(ph_p).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)
    assert float(ph_p["distInAbsLens"].max()) <= 1.75


@pytest.mark.parametrize("entries,name", [(4, "mie"), (1, "mie"), (16, "lea")])
def test_photon_history(entries, name):
    """SAVE_PHOTON_HISTORY (propagation_kernel.c.cl:452-455, 833-837, 387-392) + ConvertPhotonHistories
    (OpenCL.cxx:940-989): the last `entries` scatter points of every detected photon, oldest first."""
    from clsim_amd import converter as CV
    from oracle import builders as B
    cfg = common.config(name)
    steps = common.steps_for(cfg, 2048, seed=23)
    n = len(steps)
    x, a = common.streams(n)
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    bias_o = B.icecube_dom_acceptance()
    T = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias_o, cfg["med_o"])], bias_o, pancake=5.0, history_entries=entries)
    ph_o, cnt_o, x_o, _, raw = capi.propagate(T, steps, x, a, history=True)
    hist_o = capi.convert_photon_histories(raw, ph_o, entries)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    bias_p = CV.GetIceCubeDOMAcceptance()
    conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_dict(g), cfg["med_p"], bias_p, [CV.makeCherenkovWavelengthGenerator(bias_p, cfg["med_p"])],
                            pancakeFactor=5.0, photonHistoryEntries=entries, enableDoubleBuffering=True, approximateNumberOfWorkItems=n, streams=(x, a))
    for bunch in range(2):                        # the second bunch continues the streams: compare the first only
        conv.EnqueueSteps(steps, bunch)
    ident, ph_p, hist_p = conv.GetConversionResult(with_histories=True)
    conv.GetConversionResult()
    assert ident == 0 and cnt_o > 100 and len(ph_p) == cnt_o and len(hist_p) == cnt_o
    # photons are unordered: key each history by its photon record
    by_record = {ph_o[i].tobytes(): hist_o[i] for i in range(cnt_o)}
    assert len(by_record) == cnt_o
    longest = 0
    for i in range(cnt_o):
        want = by_record[ph_p[i].tobytes()]
        assert len(hist_p[i]) == min(int(ph_p["numScatters"][i]), entries) == len(want)
        assert hist_p[i].tobytes() == want.tobytes()
        longest = max(longest, len(want))
    assert longest == entries
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="photon histories"):
        conv.PropagateDevice(1, 256, 1, 1, 1)


def test_non_finite_steps_are_skipped_not_spun_on(capfd):
    """A NaN direction under an anisotropic medium makes the reference's photon loop spin forever (the absorption budget
    becomes NaN and never drops below EPSILON); on a GPU that is a hang.  Such steps propagate nothing, their streams
    stay untouched, the rest of the bunch is unaffected."""
    cfg = common.config("lea")
    steps = common.steps_for(cfg, 1024, seed=19)
    n = len(steps)
    bad = steps.copy()
    bad["theta"][3] = np.nan
    bad["x"][100] = np.inf
    bad["beta"][777] = -np.inf
    x, a = common.streams(n)
    T = common.oracle_tables(cfg)
    good = bad.copy()
    good["num"][[3, 100, 777]] = 0
    ph_o, cnt_o, x_o, _ = capi.propagate(T, good, x, a, threads=8)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    conv = common.product_converter(cfg, n)
    conv.EnqueueSteps(bad, 1)
    _, ph_p = conv.GetConversionResult()
    assert len(ph_p) == cnt_o and common.sort_photons(ph_o).tobytes() == common.sort_photons(ph_p).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)
    assert "3 steps of bunch 1 have non-finite" in capfd.readouterr().err
    # a source type without a wavelength generator (the reference's generateWavelength() returns 0 for it)
    cfg = common.config("flasher")
    steps = common.steps_for(cfg, 512, seed=20)
    n = len(steps)
    bad = steps.copy()
    bad["sourceType"][7] = 9
    good = bad.copy()
    good["num"][7] = 0
    x, a = common.streams(n)
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, good, x, a, threads=8)
    conv = common.product_converter(cfg, n)
    conv.EnqueueSteps(bad, 2)
    _, ph_p = conv.GetConversionResult()
    assert len(ph_p) == cnt_o and np.array_equal(conv.GetRNGState(n), x_o)
    assert "1 steps of bunch 2" in capfd.readouterr().err


def test_detector_of_576_strings_runs_with_fewer_workgroups_per_cu():
    """A table image beyond the budget of seven workgroups per CU (tables.cpp: compile_tables): both kernels run with
    what fits and stay bit-identical to the oracle."""
    import os
    cfg = common.config("mie")
    cfg["geom"] = S.large_detector_geometry()
    steps = S.cascade_steps(4096, seed=17, radius=1400.0)
    x, a = common.streams(len(steps))
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    ph_o = capi.replace_indices_with_ids(ph_o, T.geo)
    assert cnt_o > 100
    for kernel in ("classic", "pool"):
        os.environ["CLSIMHIP_KERNEL"] = kernel
        try:
            conv = common.product_converter(cfg, len(steps))
        finally:
            del os.environ["CLSIMHIP_KERNEL"]
        assert conv.GetTable("lds_bytes_per_workgroup")[0] > 160 * 1024 / 7
        conv.EnqueueSteps(steps, 5)
        ident, ph_p = conv.GetConversionResult()
        assert ident == 5 and len(ph_p) == cnt_o, kernel
        assert common.sort_photons(ph_p).tobytes() == common.sort_photons(ph_o).tobytes(), kernel
        assert np.array_equal(conv.GetRNGState(len(steps)), x_o), kernel


def test_result_buffers_follow_the_hit_counts_and_survive_a_caller_that_keeps_them(monkeypatch):
    """The page-locked result pool (converter.cpp: take_result_buffer): buffers are sized by the photons that arrive and at
    most six exist.  A caller that keeps nine results before releasing any gets the last ones from plain vectors; a bunch with
    many more photons than its predecessors gets a larger buffer; every result equals the oracle's."""
    monkeypatch.setenv("CLSIMHIP_RESULT_MIN_RECORDS", "64")        # (default 65 536 records: these bunches would never outgrow it)
    cfg = common.config("flasher")
    small, large = common.steps_for(cfg, 256, seed=5), common.steps_for(cfg, 4096, seed=6)
    x, a = common.streams(4096)
    T = common.oracle_tables(cfg)
    conv = common.product_converter(cfg, 4096, double_buffering=True)
    bunches = [small] * 4 + [large] + [small] * 4 + [large, small]
    xo, expected = x, []
    for steps in bunches:
        ph_o, cnt_o, x_next, _ = capi.propagate(T, steps, xo[:len(steps)], a[:len(steps)], threads=8)
        xo = np.concatenate([x_next, xo[len(steps):]])
        expected.append(common.sort_photons(capi.replace_indices_with_ids(ph_o, T.geo)).tobytes())
    assert len(expected[4]) > 8 * len(expected[0]) > 0, "the large bunch must need a larger buffer"
    held = []
    import threading                                    # (the input queue holds five bunches: a producer thread, like a real caller)
    producer = threading.Thread(target=lambda: [conv.EnqueueSteps(steps, k) for k, steps in enumerate(bunches[:9])])
    producer.start()
    for k in range(9):
        ident, view, release = conv.GetConversionResultInPlace()
        assert ident == k
        held.append((view, release))
    producer.join()
    for k, (view, _) in enumerate(held):                # all nine still readable, none overwritten by a later download
        assert common.sort_photons(np.array(view)).tobytes() == expected[k], k
    for _, release in held:
        release()
    for k in (9, 10):                                   # and the pool serves the next bunches from what came back
        conv.EnqueueSteps(bunches[k], k)
        ident, ph = conv.GetConversionResult()
        assert ident == k and common.sort_photons(ph).tobytes() == expected[k]
