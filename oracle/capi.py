"""
oracle/capi.py -- TEST INFRASTRUCTURE, not product code.

ctypes front end of liboracle.so (oracle/clsim_oracle.c): packs the constants
made by oracle/builders.py into `struct oracle_tables` and runs the CPU
restatement of propKernel.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import builders as B

HERE = os.path.dirname(os.path.abspath(__file__))

STEP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("t", "<f4"),
                       ("theta", "<f4"), ("phi", "<f4"), ("length", "<f4"), ("beta", "<f4"),
                       ("num", "<u4"), ("weight", "<f4"), ("id", "<u4"),
                       ("sourceType", "u1"), ("dummy1", "u1"), ("dummy2", "<u2")])
PHOTON_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("t", "<f4"),
                         ("theta", "<f4"), ("phi", "<f4"), ("wavelength", "<f4"), ("cherenkovDist", "<f4"),
                         ("numScatters", "<u4"), ("weight", "<f4"), ("id", "<u4"),
                         ("stringID", "<i2"), ("omID", "<u2"),
                         ("sx", "<f4"), ("sy", "<f4"), ("sz", "<f4"), ("st", "<f4"),
                         ("stheta", "<f4"), ("sphi", "<f4"), ("groupVelocity", "<f4"), ("distInAbsLens", "<f4")])
assert STEP_DTYPE.itemsize == 48 and PHOTON_DTYPE.itemsize == 80

MAX_GEN = 8
MAX_SUBDET = 9
FP = C.POINTER(C.c_float)


class WlenGen(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("first", C.c_float), ("spacing", C.c_float),
                ("yv", FP), ("ycum", FP), ("value", C.c_float), ("xv", FP)]


class Tables(C.Structure):
    _fields_ = [
        ("stop_detected", C.c_int32), ("has_pancake", C.c_int32), ("pancake", C.c_float),
        ("num_layers", C.c_int32), ("layer_bottom", C.c_float), ("layer_thickness", C.c_float),
        ("len_mode", C.c_int32), ("abs_const", FP), ("sca_const", FP),
        ("aDust400", FP), ("deltaTau", FP), ("b400", FP),
        ("kappa", C.c_float), ("A", C.c_float), ("B", C.c_float), ("D", C.c_float), ("E", C.c_float),
        ("alpha", C.c_float), ("ref_wlen_recip", C.c_float), ("nanometer", C.c_float),
        ("n", C.c_float * 5), ("g", C.c_float * 5), ("micrometer", C.c_float), ("c_light", C.c_float),
        ("scat_kind", C.c_int32), ("mix_frac", C.c_float), ("mix_frac_rest", C.c_float),
        ("liu_beta", C.c_float), ("hg_g", C.c_float), ("hg_g2", C.c_float),
        ("has_abs_corr", C.c_int32), ("abs_corr_const", C.c_float),
        ("an_l", C.c_float * 3), ("an_rl", C.c_float * 3), ("an_azx", C.c_float), ("an_azy", C.c_float),
        ("an_mazy", C.c_float), ("an_B2", C.c_float),
        ("has_pre", C.c_int32), ("pre_renorm", C.c_int32), ("has_post", C.c_int32), ("post_renorm", C.c_int32),
        ("pre", C.c_float * 9), ("post", C.c_float * 9),
        ("has_tilt", C.c_int32), ("tilt_const", C.c_float), ("tilt_nd", C.c_int32), ("tilt_nz", C.c_int32),
        ("tilt_first_z", C.c_float), ("tilt_dz", C.c_float), ("tilt_lnx", C.c_float), ("tilt_lny", C.c_float),
        ("tilt_dist", FP), ("tilt_zcorr", FP),
        ("num_gen", C.c_int32), ("gen", WlenGen * MAX_GEN),
        ("bias_kind", C.c_int32), ("bias_n", C.c_int32), ("bias_start", C.c_float), ("bias_step", C.c_float),
        ("bias_value", C.c_float), ("bias_data", FP),
        ("num_strings", C.c_int32), ("om_radius", C.c_float), ("string_max_radius", C.c_float),
        ("str_x", FP), ("str_y", FP), ("str_minz", FP), ("str_maxz", FP),
        ("str_set", C.POINTER(C.c_uint8)),
        ("num_sets", C.c_int32), ("max_layers", C.c_int32),
        ("set_nlayers", C.POINTER(C.c_uint16)), ("set_startz", FP), ("set_height", FP),
        ("layer_to_om", C.POINTER(C.c_uint16)),
        ("num_subdet", C.c_int32),
        ("cell_nx", C.c_int32 * MAX_SUBDET), ("cell_ny", C.c_int32 * MAX_SUBDET),
        ("cell_wx", C.c_float * MAX_SUBDET), ("cell_wy", C.c_float * MAX_SUBDET),
        ("cell_sx", C.c_float * MAX_SUBDET), ("cell_sy", C.c_float * MAX_SUBDET),
        ("cell_index", C.POINTER(C.c_uint16) * MAX_SUBDET),
        ("dom_mul_x", C.c_float), ("dom_mul_y", C.c_float),
        ("dom_tx", C.POINTER(C.c_int16)), ("dom_ty", C.POINTER(C.c_int16)), ("dom_tz", FP),
        ("dom_start", C.POINTER(C.c_uint32)), ("dom_meanx", FP), ("dom_meany", FP),
        ("tab_n", C.c_int32), ("tab_start", C.c_float), ("tab_step", C.c_float), ("tab_store16", C.c_int32),
        ("abs_q", C.POINTER(C.c_uint16)), ("sca_q", C.POINTER(C.c_uint16)),
        ("abs_lo", FP), ("abs_hi", FP), ("sca_lo", FP), ("sca_hi", FP), ("abs_f", FP), ("sca_f", FP),
        ("phase_mode", C.c_int32), ("group_mode", C.c_int32), ("phase_n", C.c_int32), ("group_n", C.c_int32),
        ("phase_start", C.c_float), ("phase_step", C.c_float), ("group_start", C.c_float), ("group_step", C.c_float),
        ("phase_data", FP), ("group_data", FP),
        ("has_fixed_abs", C.c_int32), ("fixed_abs", C.c_float), ("history_n", C.c_int32),
        ("tab_axes_kind", C.c_int32), ("tab_full_azimuth", C.c_int32),
        ("tab_ndim", C.c_int32),
        ("tab_scale", C.c_float * 5), ("tab_offset", C.c_float * 5), ("tab_inverse", C.c_int32 * 5), ("tab_inv_exp", C.c_float * 5),
        ("tab_nbins", C.c_int32 * 5), ("tab_stride", C.c_uint32 * 5),
        ("tab_max0", C.c_float), ("tab_max3", C.c_float), ("tab_min_inv_groupvel", C.c_float), ("tab_tan_thetac", C.c_float),
        ("tab_volume_step", C.c_float), ("tab_entries_per_stream", C.c_uint32),
        ("ang_n", C.c_int32), ("ang_coeff", C.c_float * 16), ("ang_has_min", C.c_int32), ("ang_has_max", C.c_int32),
        ("ang_min", C.c_float), ("ang_max", C.c_float), ("ang_underflow", C.c_float), ("ang_overflow", C.c_float),
    ]


_lib = None


def build():
    """Compile liboracle.so (gcc).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


def use_variant(name):
    """ANALYSIS ONLY (tools/math_sensitivity.py): switch to another build of the restatement -- "liboracle_libm.so"
    (glibc math instead of the deterministic header) or "liboracle_libm_mad.so" (+ fused multiply-adds) -- or back to
    None = the checker.  The parity tests never call this."""
    global _lib, _variant
    if name is not None:
        subprocess.check_call(["make", "-s", "-C", HERE, name])
    _variant = name
    _lib = None


_variant = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, _variant or "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.oracle_sizeof_tables.restype = C.c_size_t
        assert _lib.oracle_sizeof_tables() == C.sizeof(Tables), "oracle_tables layout mismatch"
        _lib.oracle_propagate.restype = C.c_uint32
        _lib.oracle_propagate_mt.restype = C.c_uint32
    return _lib


class OracleTables:
    """Owns the ctypes struct and the numpy arrays it points into."""

    def __init__(self):
        self.t = Tables()
        self._keep = []
        self.arrays = {}

    def _ptr(self, name, arr, ctype):
        arr = np.ascontiguousarray(arr)
        self._keep.append(arr)
        self.arrays[name] = arr
        return arr.ctypes.data_as(C.POINTER(ctype))

    def scalars(self):
        out = {}
        for name, ctype in Tables._fields_:
            v = getattr(self.t, name)
            if isinstance(v, (int, float)):
                out[name] = v
            elif hasattr(v, "_length_") and ctype._type_ in (C.c_float, C.c_int32):
                out[name] = list(v)
        return out


def make_tables(medium, geometry, generators, bias, pancake=5.0, stop_detected=True, fixed_abs_lengths=None, history_entries=0,
                tabulator=None):
    """The oracle's counterpart of Compile() (OpenCL.cxx:485-533): converts the
    descriptions (doubles) into the literals of the generated OpenCL program."""
    fl = B.float_literal
    fls = B.float_literals
    O = OracleTables()
    t = O.t
    t.stop_detected = 1 if stop_detected else 0
    if fixed_abs_lengths is not None and not np.isnan(fixed_abs_lengths):    # OpenCL.cxx:425-431
        t.has_fixed_abs = 1
        t.fixed_abs = fl(fixed_abs_lengths)
    t.history_n = int(history_entries)                                       # OpenCL.cxx:416-419
    if tabulator is not None:                                                # StepToTableConverter.cxx:178-207
        tb = tabulator
        t.has_fixed_abs, t.fixed_abs = 1, fl(42.0)                           # PROPAGATE_FOR_FIXED_NUMBER_OF_ABSORPTION_LENGTHS 42
        t.tab_axes_kind = 0 if tb["kind"] == "spherical" else 1
        t.tab_full_azimuth = 1 if tb["full_azimuth"] else 0
        t.tab_ndim = len(tb["axes"])                                         # > 4: TABULATE_IMPACT_ANGLE
        for k in range(t.tab_ndim):
            t.tab_scale[k], t.tab_offset[k] = tb["scale"][k], tb["offset"][k]
            t.tab_inverse[k], t.tab_nbins[k], t.tab_stride[k] = tb["inverse"][k], tb["axes"][k]["n_bins"], tb["strides"][k]
            t.tab_inv_exp[k] = tb["inv_exp"][k]
        t.tab_max0, t.tab_max3 = tb["max0"], tb["max3"]
        t.tab_min_inv_groupvel, t.tab_tan_thetac = tb["min_inv_groupvel"], tb["tan_thetac"]
        t.tab_volume_step, t.tab_entries_per_stream = tb["volume_step"], tb["entries_per_stream"]
        t.ang_n = len(tb["angular"])
        for k, c in enumerate(tb["angular"]):
            t.ang_coeff[k] = fl(c)
    t.has_pancake = 1 if pancake != 1.0 else 0          # OpenCL.cxx:432
    t.pancake = fl(pancake)
    m = medium
    t.num_layers = m["num_layers"]
    t.layer_bottom = fl(m["layers_z_start"])
    t.layer_thickness = fl(m["layers_height"])
    if m["len_mode"] == "constant":
        t.len_mode = 0
        t.abs_const = O._ptr("abs_const", fls(m["abs_const"]), C.c_float)
        t.sca_const = O._ptr("sca_const", fls(m["sca_const"]), C.c_float)
    elif m["len_mode"] == "table":
        tb = m["table"]
        t.len_mode = 2
        t.tab_n, t.tab_start, t.tab_step = tb["n"], fl(tb["start"]), fl(tb["step"])
        t.tab_store16 = 1 if tb["store16"] else 0
        if tb["store16"]:
            qa = [B.quantize_table(row) for row in tb["abs"]]
            qs = [B.quantize_table(row) for row in tb["sca"]]
            t.abs_q = O._ptr("abs_q", np.array([q[2] for q in qa], dtype=np.uint16), C.c_uint16)
            t.sca_q = O._ptr("sca_q", np.array([q[2] for q in qs], dtype=np.uint16), C.c_uint16)
            t.abs_lo = O._ptr("abs_lo", np.array([q[0] for q in qa], dtype=np.float32), C.c_float)
            t.abs_hi = O._ptr("abs_hi", np.array([q[1] for q in qa], dtype=np.float32), C.c_float)
            t.sca_lo = O._ptr("sca_lo", np.array([q[0] for q in qs], dtype=np.float32), C.c_float)
            t.sca_hi = O._ptr("sca_hi", np.array([q[1] for q in qs], dtype=np.float32), C.c_float)
        else:
            t.abs_f = O._ptr("abs_f", fls(tb["abs"]), C.c_float)
            t.sca_f = O._ptr("sca_f", fls(tb["sca"]), C.c_float)
    else:
        t.len_mode = 1
        t.aDust400 = O._ptr("aDust400", fls(m["aDust400"]), C.c_float)
        t.deltaTau = O._ptr("deltaTau", fls(m["deltaTau"]), C.c_float)
        t.b400 = O._ptr("b400", fls(m["b400"]), C.c_float)
        t.kappa, t.A, t.B, t.D, t.E = fl(m["kappa"]), fl(m["A"]), fl(m["B"]), fl(m["D"]), fl(m["E"])
        t.alpha = fl(m["alpha"])
        t.ref_wlen_recip = fl(1.0 / (400.0 * B.NANOMETER))
    t.nanometer = fl(B.NANOMETER)
    t.micrometer = fl(B.MICROMETER)
    t.c_light = fl(B.C_LIGHT)
    for i in range(5):
        t.n[i] = fl(m["n"][i])
        t.g[i] = fl(m["g"][i])
    for key in ("phase", "group"):                      # tabulated refractive indices (float data)
        if key + "_table" in m:
            ft = m[key + "_table"]
            setattr(t, key + "_mode", 1)
            setattr(t, key + "_n", len(ft["values"]))
            setattr(t, key + "_start", fl(ft["start"]))
            setattr(t, key + "_step", fl(ft["step"]))
            setattr(t, key + "_data", O._ptr(key + "_data", fls(ft["values"]), C.c_float))
    if m.get("group_from_dispersion"):                  # no group refractive index override (MediumPropertiesSource.cxx:274-300)
        assert "phase_table" not in m and "group_table" not in m, "FromTable has no derivative"
        t.group_mode = 2
    sc = m["scat"]
    g = sc["mean_cos"]
    t.liu_beta = fl((1.0 - g) / (1.0 + g))
    t.hg_g = fl(g)
    t.hg_g2 = fl(g * g)
    if sc["kind"] == "mixed":
        t.scat_kind = 2
        t.mix_frac = fl(sc["fraction"])
        t.mix_frac_rest = fl(1.0 - sc["fraction"])
    elif sc["kind"] == "hg":
        t.scat_kind = 0
    else:
        t.scat_kind = 1
    if "aniso" in m:
        c = B.aniso_constants(m["aniso"])
        t.has_abs_corr = 1
        for i in range(3):
            t.an_l[i] = fl(c["l"][i])
            t.an_rl[i] = fl(c["rl"][i])
        t.an_azx, t.an_azy, t.an_mazy, t.an_B2 = fl(c["azx"]), fl(c["azy"]), fl(-c["azy"]), fl(c["B2"])
    else:
        t.has_abs_corr = 0
        t.abs_corr_const = fl(1.0)
    for key in ("pre", "post"):
        if key in m:
            setattr(t, "has_" + key, 1)
            setattr(t, key + "_renorm", 1 if m[key]["renormalize"] else 0)
            arr = getattr(t, key)
            mat = np.asarray(m[key]["matrix"], dtype=np.float64)
            for i in range(3):
                for j in range(3):
                    arr[3 * i + j] = fl(mat[i, j])
    if "tilt" in m:
        tl = m["tilt"]
        first_z, dz = B.tilt_spacing(tl["zcoords"])
        t.has_tilt = 1
        t.tilt_nd = len(tl["distances"])
        t.tilt_nz = len(tl["zcoords"])
        t.tilt_first_z = fl(first_z)
        t.tilt_dz = fl(dz)
        t.tilt_lnx = fl(np.cos(tl["azimuth"]))
        t.tilt_lny = fl(np.sin(tl["azimuth"]))
        t.tilt_dist = O._ptr("tilt_dist", fls(tl["distances"]), C.c_float)
        t.tilt_zcorr = O._ptr("tilt_zcorr", fls(np.asarray(tl["zcorr"]).ravel()), C.c_float)
    else:
        t.has_tilt = 0
        t.tilt_const = fl(0.0)
    assert len(generators) <= MAX_GEN
    t.num_gen = len(generators)
    for k, gdesc in enumerate(generators):
        if gdesc["kind"] == "interp":
            yv, ycum = B.interp_dist_tables(gdesc)
            t.gen[k].kind = 0
            t.gen[k].n = len(yv)
            t.gen[k].first = fl(gdesc["first"])
            t.gen[k].spacing = fl(gdesc["spacing"])
            t.gen[k].yv = O._ptr("gen%d_yv" % k, yv, C.c_float)
            t.gen[k].ycum = O._ptr("gen%d_ycum" % k, ycum, C.c_float)
        elif gdesc["kind"] == "interp_x":                     # its own x values (InterpolatedDistribution.cxx:41-55)
            yv, ycum = B.interp_dist_tables(gdesc)
            t.gen[k].kind = 3
            t.gen[k].n = len(yv)
            t.gen[k].yv = O._ptr("gen%d_yv" % k, yv, C.c_float)
            t.gen[k].ycum = O._ptr("gen%d_ycum" % k, ycum, C.c_float)
            t.gen[k].xv = O._ptr("gen%d_xv" % k, B.float_literals(gdesc["x"]), C.c_float)
        elif gdesc["kind"] == "nodispersion":                # WlenCherenkovNoDispersion.cxx:72-77
            min_val = 1.0 / gdesc["to"]
            t.gen[k].kind = 2
            t.gen[k].first = fl(min_val)
            t.gen[k].spacing = fl((1.0 / gdesc["from"]) - min_val)
        else:
            t.gen[k].kind = 1
            t.gen[k].value = fl(gdesc["value"])
    if bias["kind"] == "table":
        t.bias_kind = 0
        t.bias_n = len(bias["values"])
        t.bias_start = fl(bias["start"])
        t.bias_step = fl(bias["step"])
        t.bias_data = O._ptr("bias_data", fls(bias["values"]), C.c_float)
    else:
        t.bias_kind = 1
        t.bias_value = fl(bias["value"])
    geo = geometry
    O.geo = geo
    t.num_strings = geo["num_strings"]
    t.om_radius = geo["om_radius"]
    t.string_max_radius = geo["string_max_radius"]
    for name in ("str_x", "str_y", "str_minz", "str_maxz", "set_startz", "set_height", "dom_tz", "dom_meanx", "dom_meany"):
        setattr(t, name, O._ptr(name, geo[name], C.c_float))
    t.str_set = O._ptr("str_set", geo["str_set"], C.c_uint8)
    t.num_sets = geo["num_sets"]
    t.max_layers = geo["max_layers"]
    t.set_nlayers = O._ptr("set_nlayers", geo["set_nlayers"], C.c_uint16)
    t.layer_to_om = O._ptr("layer_to_om", geo["layer_to_om"], C.c_uint16)
    assert len(geo["cells"]) <= MAX_SUBDET
    t.num_subdet = len(geo["cells"])
    for k, cell in enumerate(geo["cells"]):
        t.cell_nx[k], t.cell_ny[k] = cell["nx"], cell["ny"]
        t.cell_wx[k], t.cell_wy[k] = cell["width_x"], cell["width_y"]
        t.cell_sx[k], t.cell_sy[k] = cell["start_x"], cell["start_y"]
        t.cell_index[k] = O._ptr("cell_index_%d" % k, cell["index"], C.c_uint16)
    t.dom_mul_x, t.dom_mul_y = geo["dom_mul_x"], geo["dom_mul_y"]
    t.dom_tx = O._ptr("dom_tx", geo["dom_tx"], C.c_int16)
    t.dom_ty = O._ptr("dom_ty", geo["dom_ty"], C.c_int16)
    t.dom_start = O._ptr("dom_start", geo["dom_start"], C.c_uint32)
    return O


def propagate(tables, steps, x, a, max_hits=None, threads=1, history=False):
    """Runs the restated propKernel on steps (STEP_DTYPE) with RNG streams (x,a).
    Returns (photons[:min(count,max_hits)], count, x_after, iterations).
    String / DOM fields hold INDICES, like the kernel's raw output.
    history=True (tables.t.history_n > 0): single-threaded, returns the raw photonHistory buffer
    ([hits, history_n, 4] float32) as a fifth value."""
    L = lib()
    steps = np.ascontiguousarray(steps, dtype=STEP_DTYPE)
    n = len(steps)
    x = np.array(x[:n], dtype=np.uint64, copy=True)
    a = np.ascontiguousarray(a[:n], dtype=np.uint32)
    if max_hits is None:
        max_hits = int(steps["num"].sum()) + 1
    out = np.zeros(max_hits, dtype=PHOTON_DTYPE)
    it = C.c_uint64(0)
    args = [C.byref(tables.t), steps.ctypes.data_as(C.c_void_p), C.c_uint32(n), x.ctypes.data_as(C.c_void_p),
            a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_uint32(max_hits)]
    if history:
        nh = int(tables.t.history_n)
        raw = np.zeros((max_hits, max(nh, 1), 4), dtype=np.float32)
        L.oracle_propagate_hist.restype = C.c_uint32
        cnt = L.oracle_propagate_hist(*args, C.byref(it), raw.ctypes.data_as(C.c_void_p))
        k = min(cnt, max_hits)
        return out[:k], cnt, x, it.value, raw[:k]
    if threads == 1:
        cnt = L.oracle_propagate(*args, C.byref(it))
    else:
        cnt = L.oracle_propagate_mt(*args, C.c_int(threads), C.byref(it))
    return out[:min(cnt, max_hits)], cnt, x, it.value


# ---- the counting build (count_ops.hpp): the enumerations oc_op / oc_region / oc_event, in their order ----
COUNT_OPS = ["add", "mul", "div", "cmp", "neg", "cvt", "sqrt", "rsqrt", "log", "exp", "powr", "powr_unit", "sin", "cos", "sincos", "acos", "atan2",
             "rng_draw", "floor_trunc", "fabs", "other_math"]
COUNT_REGIONS = ["other", "create", "wavelength", "medium_per_photon", "tilt", "layer_lengths", "walk", "aniso", "scatter_angle", "rotate",
                 "transform", "search_cells", "search_string", "search_dom", "hit_record", "advance", "rng_internal", "per_step"]
COUNT_EVENTS = ["photons", "trips", "scatters", "layer_crossings", "liu", "hg", "search_calls", "cells", "strings", "dom_tests", "hits",
                "layer_length_evals", "steps", "crossing_trips"]


def count_ops(tables, steps, x, a, threads=1):
    """MEASUREMENT ONLY: the same propagation through liboracle_count.so, whose float operators count themselves.
    Returns (propagate()'s tuple, ops[region][op] as a dict of dicts, events as a dict)."""
    use_variant("liboracle_count.so")
    try:
        L = lib()
        shape = [C.c_int32(), C.c_int32(), C.c_int32()]
        L.oracle_count_shape(*[C.byref(v) for v in shape])
        assert [v.value for v in shape] == [len(COUNT_REGIONS), len(COUNT_OPS), len(COUNT_EVENTS)], "count_ops.hpp enumerations changed"
        L.oracle_count_reset()
        result = propagate(tables, steps, x, a, threads=threads)
        ops = np.zeros((len(COUNT_REGIONS), len(COUNT_OPS)), dtype=np.uint64)
        events = np.zeros(len(COUNT_EVENTS), dtype=np.uint64)
        L.oracle_count_get(ops.ctypes.data_as(C.c_void_p), events.ctypes.data_as(C.c_void_p))
    finally:
        use_variant(None)
    by_region = {r: {o: int(ops[i, j]) for j, o in enumerate(COUNT_OPS) if ops[i, j]} for i, r in enumerate(COUNT_REGIONS) if ops[i].any()}
    return result, by_region, {e: int(events[k]) for k, e in enumerate(COUNT_EVENTS)}


REQUEST_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("time", "<f4"), ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"),
                          ("length", "<f4"), ("pa", "<f4"), ("pb", "<f4"), ("kind", "<u4"), ("identifier", "<u4"),
                          ("photons_per_step", "<u4"), ("num_photons_in_last_step", "<u4"), ("num_steps", "<u8")])
assert REQUEST_DTYPE.itemsize == 64


def plan_generated_steps(requests, granularity):
    """first output step of every request, number of real steps, padded number (Async.cxx:240-257)."""
    first = np.zeros(len(requests) + 1, dtype=np.uint64)
    for i, q in enumerate(requests):
        first[i + 1] = first[i] + np.uint64(int(q["num_steps"]) + (1 if q["num_photons_in_last_step"] > 0 else 0))
    real = int(first[-1])
    return first, real, ((real + granularity - 1) // granularity) * granularity


def generate_steps(requests, seed, granularity=1):
    """oracle_generate_steps (stepgen_oracle.c)."""
    L = lib()
    req = np.ascontiguousarray(requests, dtype=REQUEST_DTYPE)
    first, real, padded = plan_generated_steps(req, granularity)
    out = np.zeros(padded, dtype=STEP_DTYPE)
    L.oracle_generate_steps(req.ctypes.data_as(C.c_void_p), first.ctypes.data_as(C.c_void_p), C.c_uint32(max(len(req), 1)),
                            C.c_uint64(real), C.c_uint64(padded), C.c_uint64(int(seed)), out.ctypes.data_as(C.c_void_p))
    return out


FLASHER_REQUEST_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("time", "<f4"), ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"),
                                  ("sigma_polar", "<f4"), ("sigma_azimuthal", "<f4"), ("pulse_width", "<f4"), ("identifier", "<u4"),
                                  ("source_type", "<u4"), ("num_photons_with_bias", "<u8")])
FLASHER_PLAN_DTYPE = np.dtype([("first_out", "<u8"), ("n_real", "<u8"), ("last_real", "<u4"), ("profile", "<u4")])
assert FLASHER_REQUEST_DTYPE.itemsize == 56 and FLASHER_PLAN_DTYPE.itemsize == 24
DIST_KINDS = {"constant": 0, "normal": 1, "uniform": 2, "flasher_time_profile": 3}


class FlasherConfig(C.Structure):
    _fields_ = [("polar_kind", C.c_int32), ("polar_value", C.c_float), ("azimuthal_kind", C.c_int32), ("azimuthal_value", C.c_float),
                ("time_kind", C.c_int32), ("time_value", C.c_float), ("polar_coordinates", C.c_int32),
                ("photons_per_step", C.c_uint32), ("max_bunch_size", C.c_uint32), ("bunch_size_granularity", C.c_uint32)]


def flasher_config(polar, azimuthal, time_delay, polar_coordinates=False, photons_per_step=400, max_bunch_size=512000, granularity=512):
    """(kind, value) pairs as in include/clsimhip.h; the defaults are the reference's (Flasher.cxx:46-48)."""
    c = FlasherConfig()
    c.polar_kind, c.polar_value = DIST_KINDS[polar[0]], polar[1]
    c.azimuthal_kind, c.azimuthal_value = DIST_KINDS[azimuthal[0]], azimuthal[1]
    c.time_kind, c.time_value = DIST_KINDS[time_delay[0]], time_delay[1]
    c.polar_coordinates = 1 if polar_coordinates else 0
    c.photons_per_step, c.max_bunch_size, c.bunch_size_granularity = photons_per_step, max_bunch_size, granularity
    return c


def plan_flasher_steps(cfg, requests):
    """Bunch plan from the literal MakeSteps model (builders.flasher_make_steps_model) and the time profile tables."""
    from . import builders as B
    req = np.ascontiguousarray(requests, dtype=FLASHER_REQUEST_DTYPE)
    plan = np.zeros(len(req), dtype=FLASHER_PLAN_DTYPE)
    widths, out, counts = [], 0, []
    for i, q in enumerate(req):
        photons = B.flasher_make_steps_model(int(q["num_photons_with_bias"]), cfg.photons_per_step, cfg.max_bunch_size, cfg.bunch_size_granularity)
        real = [p for p in photons if p > 0]
        # the kernel lays a pulse out as its real steps followed by its dummy steps; the reference interleaves dummy steps
        # only at the end of a result, and only the last result of a pulse has any
        assert all(p > 0 for p in photons[:len(real)]) and all(p == cfg.photons_per_step for p in real[:-1])
        plan[i] = (out, len(real), real[-1] if real else cfg.photons_per_step, 0)
        if cfg.time_kind == 3:
            w = float(q["pulse_width"])
            if w not in widths:
                widths.append(w)
            plan[i]["profile"] = widths.index(w)
        out += len(photons)
        counts.append(photons)
    profiles = np.zeros((max(len(widths), 1), 2, 240), dtype=np.float32)
    for k, w in enumerate(widths):
        profiles[k, 0], profiles[k, 1] = B.interpolated_distribution_tables(0.5, B.flasher_time_profile(w))
    return plan, out, profiles, counts


def generate_flasher_steps(cfg, requests, seed):
    """oracle_generate_flasher_steps (stepgen_oracle.c)."""
    L = lib()
    req = np.ascontiguousarray(requests, dtype=FLASHER_REQUEST_DTYPE)
    plan, total, profiles, _ = plan_flasher_steps(cfg, req)
    out = np.zeros(total, dtype=STEP_DTYPE)
    L.oracle_generate_flasher_steps(C.byref(cfg), req.ctypes.data_as(C.c_void_p), plan.ctypes.data_as(C.c_void_p), C.c_uint32(len(req)),
                                    C.c_uint64(total), C.c_uint64(int(seed)), profiles.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


ENTRY_DTYPE = np.dtype([("index", np.uint32), ("weight", np.float32)])


def tabulate(tables, steps, x, a, reference, threads=8):
    """One launch of the TABULATE kernel (oracle_tabulate).  Returns (entries [n, EPS], num_entries [n],
    photons_left [n], x_after)."""
    L = lib()
    steps = np.ascontiguousarray(steps, dtype=STEP_DTYPE)
    n = len(steps)
    eps = int(tables.t.tab_entries_per_stream)
    x = np.array(x[:n], dtype=np.uint64, copy=True)
    a = np.ascontiguousarray(a[:n], dtype=np.uint32)
    ref = np.ascontiguousarray(reference, dtype=np.float32)
    assert ref.shape == (12,)
    entries = np.zeros((n, eps), dtype=ENTRY_DTYPE)
    num = np.zeros(n, dtype=np.uint32)
    left = np.zeros(n, dtype=np.uint32)
    L.oracle_tabulate(C.byref(tables.t), steps.ctypes.data_as(C.c_void_p), C.c_uint32(n), x.ctypes.data_as(C.c_void_p),
                      a.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p), entries.ctypes.data_as(C.c_void_p),
                      num.ctypes.data_as(C.c_void_p), left.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return entries, num, left, x


def tabulate_accumulate(tables, steps, x, a, reference, bins=None, photons_per_call=8, threads=8):
    """The table maker's host loop around the kernel (oracle_tabulate_accumulate; StepToTableConverter.cxx:399-460,
    495-507) for bunches whose entries would not fit in memory: every step runs `photons_per_call` photons per kernel
    call, the calls' entries are added to `bins` (float64 array of n_bins, in place; None: not kept).  Returns
    (per-step sum of weights [n] float64, per-step number of entries [n] uint64, x_after).  Raises if a call runs out of
    entry space (tables.t.tab_entries_per_stream slots for photons_per_call photons)."""
    L = lib()
    steps = np.ascontiguousarray(steps, dtype=STEP_DTYPE)
    n = len(steps)
    x = np.array(x[:n], dtype=np.uint64, copy=True)
    a = np.ascontiguousarray(a[:n], dtype=np.uint32)
    ref = np.ascontiguousarray(reference, dtype=np.float32)
    assert ref.shape == (12,)
    if bins is not None:
        n_bins = int(tables.t.tab_stride[0]) * (int(tables.t.tab_nbins[0]) + 2)            # Axes.cxx:51-64
        assert bins.dtype == np.float64 and bins.flags["C_CONTIGUOUS"] and bins.size == n_bins
    sums = np.zeros(n, dtype=np.float64)
    counts = np.zeros(n, dtype=np.uint64)
    L.oracle_tabulate_accumulate.restype = C.c_int
    rc = L.oracle_tabulate_accumulate(C.byref(tables.t), steps.ctypes.data_as(C.c_void_p), C.c_uint32(n), x.ctypes.data_as(C.c_void_p),
                                      a.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p),
                                      bins.ctypes.data_as(C.c_void_p) if bins is not None else None,
                                      sums.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                      C.c_uint32(photons_per_call), C.c_int(threads))
    if rc != 0:
        raise RuntimeError("oracle_tabulate_accumulate: %d (-1: %d photons did not fit into %d entries)"
                           % (rc, photons_per_call, int(tables.t.tab_entries_per_stream)))
    return sums, counts, x


def accumulate_entries(entries, num, n_bins, dtype=np.float32):
    """The host loop of StepToTableConverter.cxx:495-507: binContent_[index] += weight, stream by stream in entry
    order (float accumulators in the reference; float64 gives the order-independent sum)."""
    bins = np.zeros(n_bins, dtype=dtype)
    for i in range(len(num)):
        k = int(num[i])
        if k:
            np.add.at(bins, entries["index"][i, :k], entries["weight"][i, :k].astype(dtype))
    return bins


def eval_tabulator(tables, reference, pos_and_time):
    L = lib()
    p = np.ascontiguousarray(pos_and_time, dtype=np.float32).reshape(-1, 4)
    ref = np.ascontiguousarray(reference, dtype=np.float32)
    coords = np.zeros_like(p)
    index = np.zeros(len(p), dtype=np.uint32)
    oob = np.zeros(len(p), dtype=np.int32)
    L.oracle_eval_tabulator(C.byref(tables.t), ref.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), C.c_int(len(p)),
                            coords.ctypes.data_as(C.c_void_p), index.ctypes.data_as(C.c_void_p), oob.ctypes.data_as(C.c_void_p))
    return coords, index, oob


def convert_photon_histories(raw, photons, entries):
    """ConvertPhotonHistories (OpenCL.cxx:940-989): the ring buffer of each photon in forward order (most recent
    scatter last); only min(numScatters, entries) points exist.  Returns a list of [k, 4] arrays."""
    out = []
    for i in range(len(photons)):
        ns = int(photons["numScatters"][i])
        if ns == 0 or entries == 0:
            out.append(np.zeros((0, 4), dtype=np.float32))
            continue
        k = min(ns, entries)
        cur = 0 if ns <= entries else ns % entries
        rows = []
        for _ in range(k):
            rows.append(raw[i, cur])
            cur += 1
            if cur >= entries:
                cur = 0
        out.append(np.array(rows, dtype=np.float32))
    return out


def replace_indices_with_ids(photons, geo):
    """ReplaceStringDOMIndexWithStringDOMIDs (OpenCL.cxx:1565-1600)."""
    out = photons.copy()
    for k in range(len(out)):
        s = int(out["stringID"][k])
        d = int(out["omID"][k])
        out["stringID"][k] = geo["string_index_to_id"][s]
        out["omID"][k] = geo["dom_index_to_id"][s][d]
    return out


def sort_photons(ph):
    """Canonical order for multiset comparison."""
    raw = np.ascontiguousarray(ph).view(np.uint32).reshape(len(ph), 20)
    keys = [raw[:, c] for c in range(19, -1, -1)]
    keys += [ph["numScatters"], ph["omID"], ph["stringID"], ph["id"]]
    return ph[np.lexsort(keys)]


def eval_math(what, xs, ys=None):
    L = lib()
    xs = np.ascontiguousarray(xs, dtype=np.float32)
    out = np.empty_like(xs)
    yp = None
    if ys is not None:
        ys = np.ascontiguousarray(ys, dtype=np.float32)
        yp = ys.ctypes.data_as(C.c_void_p)
    L.oracle_eval_math(C.c_int(what), xs.ctypes.data_as(C.c_void_p), yp, C.c_int(len(xs)), out.ctypes.data_as(C.c_void_p))
    return out


def eval_medium(tables, what, wlens, layer=0):
    L = lib()
    w = np.ascontiguousarray(wlens, dtype=np.float32)
    out = np.empty_like(w)
    L.oracle_eval_medium(C.byref(tables.t), C.c_int(what), w.ctypes.data_as(C.c_void_p), C.c_int(len(w)),
                         C.c_int(layer), out.ctypes.data_as(C.c_void_p))
    return out


def eval_field(tables, what, xyz):
    L = lib()
    v = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.empty(len(v) * (3 if what >= 2 else 1), dtype=np.float32)
    L.oracle_eval_field(C.byref(tables.t), C.c_int(what), v.ctypes.data_as(C.c_void_p), C.c_int(len(v)),
                        out.ctypes.data_as(C.c_void_p))
    return out.reshape(-1, 3) if what >= 2 else out


def eval_rng(x, a, n):
    L = lib()
    xs = C.c_uint64(int(x))
    out = np.empty(n, dtype=np.float32)
    L.oracle_eval_rng(C.byref(xs), C.c_uint32(int(a)), C.c_int(n), out.ctypes.data_as(C.c_void_p))
    return out, xs.value


def generate_wavelengths(tables, generator, n, seed=1):
    """n draws of generateWavelength_<generator> from one MWC stream (oracle_eval_wlen)"""
    L = lib()
    xs = C.c_uint64(0x9E3779B97F4A7C15 ^ int(seed) * 0x100000001B3 & 0xFFFFFFFFFFFFFFFF | 1)
    out = np.empty(n, dtype=np.float32)
    L.oracle_eval_wlen(C.byref(tables.t), C.c_int(int(generator)), C.byref(xs), C.c_uint32(4294967118), C.c_int(n), out.ctypes.data_as(C.c_void_p))
    return out


def sample(tables, what, x, a, draws, generator=0):
    """`draws` values from every stream (x[i], a[i]): what = 'uniform' (rand_MWC_co), 'wavelength' (generateWavelength_<generator>),
    'scattering_cosine' (makeScatteringCosAngle).  Returns (values (n, draws), final states) -- oracle_eval_rng / _wlen / _scatcos."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.uint64)
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.empty((len(x), draws), dtype=np.float32)
    x_out = np.empty_like(x)
    row = np.empty(draws, dtype=np.float32)
    for i in range(len(x)):
        xs = C.c_uint64(int(x[i]))
        p = row.ctypes.data_as(C.c_void_p)
        if what == "uniform":
            L.oracle_eval_rng(C.byref(xs), C.c_uint32(int(a[i])), C.c_int(draws), p)
        elif what == "wavelength":
            L.oracle_eval_wlen(C.byref(tables.t), C.c_int(int(generator)), C.byref(xs), C.c_uint32(int(a[i])), C.c_int(draws), p)
        elif what == "scattering_cosine":
            L.oracle_eval_scatcos(C.byref(tables.t), C.byref(xs), C.c_uint32(int(a[i])), C.c_int(draws), p)
        else:
            raise ValueError(what)
        out[i] = row
        x_out[i] = xs.value
    return out, x_out
