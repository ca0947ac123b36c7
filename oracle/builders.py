"""
oracle/builders.py -- TEST INFRASTRUCTURE, not product code.

numpy / plain-Python restatement of the reference's host-side "code generators":
everything that turns a medium / geometry / spectrum description into the float
constants the OpenCL kernel is compiled with.  The product has its own C++
implementation of the same steps (clsim_amd/csrc/*.cpp); tests compare the two
table by table.

Each function cites the reference file:line it follows (paths relative to the
reference root).  All arithmetic is IEEE binary64 in the reference's operation
order; values reach the kernel through `float_literal` = what an OpenCL compiler
makes of the "%.10e" text the reference prints
(private/clsim/I3CLSimHelperToFloatString.h:36-59).
"""
import ctypes
import math
import os

import numpy as np

_libc = ctypes.CDLL(None)
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]

# I3Units / I3Constants (icetray): m = ns = 1
NANOMETER = 1e-9
MICROMETER = 1e-6
DEG = math.pi / 180.0
C_LIGHT = 0.299792458


def float_literal(v):
    """binary32 value of the literal ToFloatString(v) (ToFloatString.h:36-59):
    scientific notation, precision digits10+4 = 10, parsed by the compiler
    with correct rounding (strtof)."""
    return np.float32(_libc.strtof(("%.10e" % float(v)).encode(), None))


def float_literals(vs):
    return np.array([float_literal(v) for v in np.asarray(vs, dtype=np.float64).ravel()],
                    dtype=np.float32).reshape(np.shape(vs))


# --------------------------------------------------------------------------
# DOM acceptance, wavelength generator
# --------------------------------------------------------------------------

# python/GetIceCubeDOMAcceptance.py:62-105 (photonics efficiency.h table, m^2)
DOM2007A_EFF_AREA = [
    0.0000064522, 0.0000064522, 0.0000064522, 0.0000064522, 0.0000021980, 0.0001339040,
    0.0005556810, 0.0016953000, 0.0035997000, 0.0061340900, 0.0074592700, 0.0090579800,
    0.0099246700, 0.0105769000, 0.0110961000, 0.0114214000, 0.0114425000, 0.0111527000,
    0.0108086000, 0.0104458000, 0.0099763100, 0.0093102500, 0.0087516600, 0.0083225800,
    0.0079767200, 0.0075625100, 0.0066377000, 0.0053335800, 0.0043789400, 0.0037583500,
    0.0033279800, 0.0029212500, 0.0025334900, 0.0021115400, 0.0017363300, 0.0013552700,
    0.0010546600, 0.0007201020, 0.0004843820, 0.0002911110, 0.0001782310, 0.0001144300,
    0.0000509155]


def icecube_dom_acceptance(dom_radius=0.16510, efficiency=1.0):
    """python/GetIceCubeDOMAcceptance.py:35-115 (highQE=False).  Returns the
    FunctionFromTable description (start, step, values) in doubles."""
    area = np.array(DOM2007A_EFF_AREA, dtype=np.float64) * 1.0
    dom_area = math.pi * dom_radius ** 2.0
    eff = efficiency * (area / dom_area)
    return dict(kind="table", start=260.0 * NANOMETER, step=10.0 * NANOMETER, values=eff)


REFINDEX_N = (1.55749, -1.57988, 3.99993, -4.68271, 2.09354)       # RefIndexIceCube.cxx:36-40
REFINDEX_G = (1.227106, -0.954648, 1.42568, -0.711832, 0.0)        # RefIndexIceCube.cxx:43-47


def phase_ref_index_host(wlen, n=REFINDEX_N):
    """I3CLSimFunctionRefIndexIceCube::GetValue, mode 'phase' (RefIndexIceCube.cxx:84-101)."""
    x = wlen / MICROMETER
    return n[0] + x * (n[1] + x * (n[2] + x * (n[3] + x * n[4])))


def from_table_host(tab, wlen):
    """I3CLSimFunctionFromTable::GetValue, equal spacing (FromTable.cxx:105-122)."""
    q = (wlen - tab["start"]) / tab["step"]
    fraction, fbin = math.modf(q)
    ibin = int(fbin)
    n = len(tab["values"])
    if ibin < 0 or (ibin == 0 and fraction < 0):
        ibin, fraction = 0, 0.0
    elif ibin >= n - 1:
        ibin, fraction = n - 2, 1.0
    v = tab["values"]
    return v[ibin] + (v[ibin + 1] - v[ibin]) * fraction


def from_table_x_host(tab, wlen):
    """I3CLSimFunctionFromTable::GetValue with the table's own wavelengths (FromTable.cxx:123-145)."""
    w, v = tab["wavelengths"], tab["values"]
    if wlen <= w[0]:
        return v[0]
    for i in range(1, len(w)):
        if wlen <= w[i]:
            fraction = (wlen - w[i - 1]) / (w[i] - w[i - 1])
            return v[i - 1] + (v[i] - v[i - 1]) * fraction
    return v[len(w) - 1]


FLASHER_LED_SPECTRA = {
    # python/GetIceCubeFlasherSpectrum.py:38-60: file under resources/flasher_data/, normalisation constant
    "LED340nm": ("flasher_led_340nm_emission_spectrum_cw_measured_20mA_pulseCurrent.txt", 24.306508),
    "LED370nm": ("flasher_led_370nm_emission_spectrum_cw_measured.txt", 15.7001863),
    "LED405nm": ("flasher_led_405nm_emission_spectrum_datasheet.txt", 8541585.10324),
    "LED450nm": ("flasher_led_450nm_emission_spectrum_datasheet.txt", 21.9792812618),
    "LED505nm": ("flasher_led_505nm_emission_spectrum_cw_measured.txt", 38.1881),
}


def flasher_spectrum(kind, data_dir):
    """python/GetIceCubeFlasherSpectrum.py:38-82: the LED's emission spectrum as an I3CLSimFunctionFromTable with its own
    wavelengths (numpy.loadtxt, wavelengths * nanometer, values / constant); SC1 / SC2: a delta peak at 337 nm."""
    if kind in ("SC1", "SC2"):
        return dict(kind="delta", value=337.0 * NANOMETER)
    name, norm = FLASHER_LED_SPECTRA[kind]
    data = np.loadtxt(os.path.join(data_dir, name), unpack=True)
    w = data[0] * NANOMETER
    v = data[1] / norm
    return dict(kind="table_x", wavelengths=w, values=v)


def make_wavelength_generator(spectrum, bias, medium):
    """I3CLSimModuleHelper::makeWavelengthGenerator (I3CLSimModuleHelper.cxx:73-171) for the spectrum classes its callers
    pass: a delta peak -> RandomValueConstant; a table -> InterpolatedDistribution on the table's own binning, every
    entry multiplied by the bias at its wavelength (no clipping to the medium's range, :100-106)."""
    if spectrum["kind"] == "delta":
        return dict(kind="const", value=spectrum["value"])
    bias_at = (lambda wl: from_table_host(bias, wl)) if bias["kind"] == "table" else (lambda wl: bias["value"])
    if spectrum["kind"] == "table_x":
        w = np.asarray(spectrum["wavelengths"], dtype=np.float64)
        y = np.array([bias_at(w[i]) * spectrum["values"][i] for i in range(len(w))])
        return dict(kind="interp_x", x=w, y=y)
    assert spectrum["kind"] == "table"
    n = len(spectrum["values"])
    y = np.array([bias_at(spectrum["start"] + float(i) * spectrum["step"]) * spectrum["values"][i] for i in range(n)])
    return dict(kind="interp", first=spectrum["start"], spacing=spectrum["step"], y=y)


def cherenkov_wlen_generator(bias, medium, beta=1.0):
    """I3CLSimModuleHelper::makeCherenkovWavelengthGenerator with a tabulated
    bias and dispersion on (I3CLSimModuleHelper.cxx:175-263, 52-63).  Returns an
    InterpolatedDistribution description (first, spacing, y)."""
    assert bias["kind"] == "table"
    n = len(bias["values"])
    y = np.empty(n, dtype=np.float64)
    for i in range(n):
        wl = bias["start"] + float(i) * bias["step"]          # FromTable.cxx:88-90
        if "phase_table" in medium:
            nphase = from_table_host(medium["phase_table"], wl)
        else:
            nphase = phase_ref_index_host(wl, medium["n"])
        y[i] = bias["values"][i] * ((2.0 * math.pi / (137.0 * (wl * wl))) * (1.0 - 1.0 / (math.pow(beta * nphase, 2.0))))
    return dict(kind="interp", first=bias["start"], spacing=bias["step"], y=y)


def interp_dist_tables(gen):
    """I3CLSimRandomValueInterpolatedDistribution::InitTables + WriteTableCode
    (InterpolatedDistribution.cxx:134-175, 177-234): -> (yv, ycum) literals."""
    y = np.asarray(gen["y"], dtype=np.float64)
    n = len(y)
    acu = np.zeros(n, dtype=np.float64)
    if gen["kind"] == "interp_x":                         # :148-154
        x = np.asarray(gen["x"], dtype=np.float64)
        for j in range(1, n):
            acu[j] = acu[j - 1] + (x[j] - x[j - 1]) * (y[j] + y[j - 1]) / 2.0
    else:
        for j in range(1, n):
            acu[j] = acu[j - 1] + (gen["spacing"]) * (y[j] + y[j - 1]) / 2.0
    total = acu[n - 1]
    beta = np.empty(n)
    for j in range(n):
        beta[j] = y[j] / total
        acu[j] = acu[j] / total
    return float_literals(beta), float_literals(acu)


# --------------------------------------------------------------------------
# ice model loader (PPC tables)
# --------------------------------------------------------------------------

def _loadtxt(path):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.split("#")[0].strip()
            if line:
                rows.append([float(t) for t in line.split()])
    return rows


def load_ppc_ice(directory, detector_center_depth=1948.07, use_tilt_if_available=True):
    """python/MakeIceCubeMediumProperties.py:49-256 (+ util/GetIceTiltZShift.py:40-62,
    util/GetSpiceLeaAnisotropyTransforms.py:39-101).  Returns a medium description
    in doubles (the values the reference hands to its C++ function objects)."""
    use_tilt = False
    if use_tilt_if_available:
        has_par = os.path.isfile(os.path.join(directory, "tilt.par"))
        has_dat = os.path.isfile(os.path.join(directory, "tilt.dat"))
        if has_par != has_dat:
            raise RuntimeError("tilt.par / tilt.dat: one of the two is missing")
        use_tilt = has_par and has_dat
    dat = np.array(_loadtxt(os.path.join(directory, "icemodel.dat")), dtype=np.float64).T
    par = np.array(_loadtxt(os.path.join(directory, "icemodel.par")), dtype=np.float64)
    cfg = np.array([r[0] for r in _loadtxt(os.path.join(directory, "cfg.txt"))], dtype=np.float64)
    if len(par) == 6:
        alpha, kappa, A, B, D, E = (par[i][0] for i in range(6))
    elif len(par) == 4:
        alpha, kappa, A, B = (par[i][0] for i in range(4))
        D = 400.0 ** kappa
        E = 0.0
    else:
        raise RuntimeError("icemodel.par needs 4 or 6 rows")
    if len(cfg) < 4:
        raise RuntimeError("cfg.txt needs at least 4 lines")
    liu_fraction = cfg[2]
    mean_cos = cfg[3]
    has_aniso = False
    if 4 < len(cfg) < 7:
        raise RuntimeError("cfg.txt: anisotropy needs 7 lines")
    elif len(cfg) > 4:
        has_aniso = True
        an_az = cfg[4] * DEG
        an_k1 = cfg[5]
        an_k2 = cfg[6]
    depth = dat[0] * 1.0
    b_e400, a_dust400, delta_tau = dat[1], dat[2], dat[3]
    layer_height = depth[1] - depth[0]
    if layer_height <= 0:
        raise RuntimeError("ice layer depths are not in increasing order")
    for i in range(len(depth) - 1):
        if abs((depth[i + 1] - depth[i]) - layer_height) > 1e-5:
            raise RuntimeError("ice layers are not spaced evenly")
    depth = depth[::-1]
    b_e400 = b_e400[::-1]
    a_dust400 = a_dust400[::-1]
    delta_tau = delta_tau[::-1]
    b_400 = b_e400 / (1.0 - mean_cos)
    depth = depth - layer_height / 2.0
    depth_bottom = depth + layer_height
    layer_z_start = detector_center_depth - depth_bottom
    med = dict(
        num_layers=len(layer_z_start), layers_z_start=float(layer_z_start[0]), layers_height=float(layer_height),
        min_wlen=265.0 * NANOMETER, max_wlen=675.0 * NANOMETER,
        len_mode="icecube", alpha=alpha, kappa=kappa, A=A, B=B, D=D, E=E,
        aDust400=np.array(a_dust400), deltaTau=np.array(delta_tau), b400=np.array(b_400),
        n=REFINDEX_N, g=REFINDEX_G,
        scat=dict(kind="mixed", fraction=float(liu_fraction), mean_cos=float(mean_cos)),
    )
    if has_aniso:
        med["aniso"] = dict(azimuth=an_az, k1=an_k1, k2=an_k2)
        k1 = np.exp(an_k1); k2 = np.exp(an_k2); kz = 1.0 / (k1 * k2)
        Am = np.array([[k1, 0., 0.], [0., k2, 0.], [0., 0., kz]])
        sa = np.sin(an_az); ca = np.cos(an_az)
        T = np.array([[ca, sa, 0.], [-sa, ca, 0.], [0., 0., 1.]])
        med["pre"] = dict(matrix=np.dot(np.dot(T.T, Am), T), renormalize=True)
        med["post"] = dict(matrix=np.dot(np.dot(T.T, np.linalg.inv(Am)), T), renormalize=True)
    if use_tilt:
        tpar = np.array(_loadtxt(os.path.join(directory, "tilt.par")), dtype=np.float64).T
        tdat = np.array(_loadtxt(os.path.join(directory, "tilt.dat")), dtype=np.float64).T
        dist = tpar[1] * 1.0
        zcoords = (detector_center_depth - tdat[0])[::-1]
        zshift = np.array([tdat[i + 1][::-1] for i in range(len(dist))])
        med["tilt"] = dict(distances=dist, zcoords=zcoords, zcorr=zshift, azimuth=225.0 * DEG)
    return med


def load_photonics_ice(table_file, detector_center_depth=1948.07):
    """python/MakeIceCubeMediumPropertiesPhotonics.py:47-227: photonics ice table -> medium with one
    FromTable function per layer for the absorption / scattering length (stored in 16 bits), tabulated phase and
    group refractive index (one function for all layers), Henyey-Greenstein scattering, no tilt, no anisotropy."""
    with open(table_file) as f:
        raw = f.readlines()
    parsed = [line.split() for line in raw if line.strip() and line.lstrip()[0] != "#"]
    nlayer = [l for l in parsed if l[0].upper() == "NLAYER"]
    nwvl = [l for l in parsed if l[0].upper() == "NWVL"]
    if len(nlayer) != 1 or len(nwvl) != 1:
        raise RuntimeError("the ice table needs exactly one NLAYER and one NWVL entry")
    n_layers = int(nlayer[0][1])
    n_wlen = int(nwvl[0][1])
    start = float(nwvl[0][2]) * NANOMETER
    step = float(nwvl[0][3]) * NANOMETER
    start += step / 2.0
    parsed = [l for l in parsed if l[0].upper() not in ("NLAYER", "NWVL")]
    if len(parsed) != n_layers * 6:
        raise RuntimeError("expected %d lines in the ice table, found %d" % (n_layers * 6, len(parsed)))
    if parsed[0][0].upper() != "LAYER":
        raise RuntimeError("layer definitions should start with the LAYER keyword")
    layers, cur = [], {}
    for line in parsed:
        key = line[0].upper()
        if key == "LAYER":
            if cur:
                layers.append(cur)
            cur = {}
        elif key in cur:
            raise RuntimeError("keyword %s is used twice for one layer" % key)
        cur[key] = [float(t) * 1.0 for t in line[1:]]
    if cur:
        layers.append(cur)
    if not layers:
        raise RuntimeError("at least one layer is required")
    height = abs(layers[0]["LAYER"][1] - layers[0]["LAYER"][0])
    by_z = {}
    for layer in layers:
        bottom, top = layer["LAYER"][0], layer["LAYER"][1]
        if bottom > top:
            bottom, top = top, bottom
        if abs((top - bottom) - height) > 0.0001:
            raise RuntimeError("differing layer heights")
        by_z[bottom] = layer
    layers, end_z = [], None
    for _, layer in sorted(by_z.items()):
        start_z = layer["LAYER"][0]
        if end_z is not None and abs(end_z - start_z) > 0.0001:
            raise RuntimeError("your layers have holes")
        end_z = layer["LAYER"][1]
        layers.append(layer)
    mean_cos = None
    for layer in layers:
        if mean_cos is None:
            mean_cos = layer["COS"][0]
        for c in layer["COS"]:
            if abs(c - mean_cos) > 0.0001:
                raise RuntimeError("only a constant mean cosine is supported")
        for key in ("COS", "ABS", "SCAT", "N_GROUP", "N_PHASE"):
            if len(layer[key]) != n_wlen:
                raise RuntimeError("expected %d %s values, got %d" % (n_wlen, key, len(layer[key])))
        for i in range(n_wlen):
            if abs(layer["N_GROUP"][i] - layers[0]["N_GROUP"][i]) > 0.0001 or abs(layer["N_PHASE"][i] - layers[0]["N_PHASE"][i]) > 0.0001:
                raise RuntimeError("N_GROUP / N_PHASE may not differ between layers")
    abs_tab = np.array([[1.0 / a for a in layer["ABS"]] for layer in layers])
    sca_tab = np.array([[(1.0 / s) * (1.0 - mean_cos) for s in layer["SCAT"]] for layer in layers])
    last = start + step * float(n_wlen - 1)                     # FromTable.cxx:157-164 (GetMaxWlen)
    return dict(num_layers=len(layers), layers_z_start=layers[0]["LAYER"][0], layers_height=height,
                min_wlen=start, max_wlen=last,                  # MediumProperties.cxx:85-153, nothing forced
                len_mode="table", table=dict(start=start, step=step, n=n_wlen, store16=True, abs=abs_tab, sca=sca_tab),
                phase_table=dict(kind="table", start=start, step=step, values=np.array(layers[0]["N_PHASE"])),
                group_table=dict(kind="table", start=start, step=step, values=np.array(layers[0]["N_GROUP"])),
                n=(0.0,) * 5, g=(0.0,) * 5,
                scat=dict(kind="hg", mean_cos=float(mean_cos)))


def quantize_table(values):
    """I3CLSimFunctionFromTable::GetOpenCLFunction, 16-bit storage (FromTable.cxx:183-207): -> (smallest literal,
    largest literal, uint16 data)."""
    v = np.asarray(values, dtype=np.float64)
    lo, hi = float(v[0]), float(v[0])
    for x in v[1:]:
        if x < lo: lo = float(x)
        if x > hi: hi = float(x)
    q = np.array([int(65535.0 * (float(x) - lo) / (hi - lo)) for x in v], dtype=np.uint16)   # static_cast<uint16_t>: truncation
    return float_literal(lo), float_literal(hi), q


def homogeneous_medium(abs_len=100.0, sca_len=25.0, z_start=-1000.0, height=2000.0, mean_cos=0.9, liu_fraction=0.45):
    """BASELINE config C1 (SURVEY.md 9.7 option i): one layer, FunctionConstant
    absorption / scattering length, IceCube refractive index, Mixed(Liu,HG)."""
    return dict(num_layers=1, layers_z_start=z_start, layers_height=height,
                min_wlen=265.0 * NANOMETER, max_wlen=675.0 * NANOMETER,
                len_mode="constant", abs_const=np.array([abs_len]), sca_const=np.array([sca_len]),
                n=REFINDEX_N, g=REFINDEX_G,
                scat=dict(kind="mixed", fraction=liu_fraction, mean_cos=mean_cos))


def tilt_spacing(zcoords):
    """ScalarFieldIceTiltZShift.cxx:62-89: mean z spacing and first coordinate."""
    z = np.asarray(zcoords, dtype=np.float64)
    mean = 0.0
    for i in range(len(z) - 1):
        mean += z[i + 1] - z[i]
    mean /= float(len(z) - 1)
    return float(z[0]), mean


def aniso_constants(an):
    """ScalarFieldAnisotropyAbsLenScaling.cxx:96-108: host constants."""
    azx = math.cos(an["azimuth"]); azy = math.sin(an["azimuth"])
    k1 = math.exp(an["k1"]); k2 = math.exp(an["k2"]); kz = 1.0 / (k1 * k2)
    l1 = k1 * k1; l2 = k2 * k2; l3 = kz * kz
    B2 = 1.0 / l1 + 1.0 / l2 + 1.0 / l3
    return dict(azx=azx, azy=azy, l=(l1, l2, l3), rl=(1.0 / l1, 1.0 / l2, 1.0 / l3), B2=B2)


# --------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------

def _cell_contains(lo, hi, cmin, cmax):
    c = False
    if lo <= cmin and hi >= cmin: c = True
    if lo <= cmax and hi >= cmax: c = True
    if lo >= cmin and hi <= cmax: c = True
    return c


def _divide_into_cells(strings, sd, n):
    """GeometrySource.cxx:135-271."""
    sel = [s for s in strings if s["sd"] == sd]
    min_x = min_y = max_x = max_y = float("nan")
    for s in sel:
        if (s["meanX"] - s["maxR"] < min_x) or math.isnan(min_x): min_x = s["meanX"] - s["maxR"]
        if (s["meanY"] - s["maxR"] < min_y) or math.isnan(min_y): min_y = s["meanY"] - s["maxR"]
        if (s["meanX"] + s["maxR"] > max_x) or math.isnan(max_x): max_x = s["meanX"] + s["maxR"]
        if (s["meanY"] + s["maxR"] > max_y) or math.isnan(max_y): max_y = s["meanY"] + s["maxR"]
    start_x, start_y = min_x, min_y
    wx = (max_x - min_x) / float(n)
    wy = (max_y - min_y) / float(n)
    cells = [0xFFFF] * (n * n)
    for i in range(n):
        cx0 = start_x + float(i) * wx
        cx1 = start_x + float(i + 1) * wx
        xs = [(k, s) for k, s in enumerate(strings) if s["sd"] == sd and
              _cell_contains(s["meanX"] - s["maxR"], s["meanX"] + s["maxR"], cx0, cx1)]
        for j in range(n):
            cy0 = start_y + float(j) * wy
            cy1 = start_y + float(j + 1) * wy
            found = 0xFFFF
            for k, s in xs:
                if _cell_contains(s["meanY"] - s["maxR"], s["meanY"] + s["maxR"], cy0, cy1):
                    if found != 0xFFFF:
                        return None
                    found = k
            cells[j * n + i] = found
    return start_x, start_y, wx, wy, cells


def _layer_contains(z, r, zmin, zmax):
    return _cell_contains(z - r, z + r, zmin, zmax)


def _divide_into_layers(s, n, r, min_hint, max_hint):
    """GeometrySource.cxx:375-446."""
    if n == 0 or r < 0: return None
    table = [0xFFFF] * n
    min_z, max_z = min_hint, max_hint
    if (s["minZ"] - r < min_z) or math.isnan(min_z): min_z = s["minZ"] - r
    if (s["maxZ"] + r > max_z) or math.isnan(max_z): max_z = s["maxZ"] + r
    start = min_z
    height = (max_z - min_z) / float(n)
    for i in range(n):
        z0 = start + float(i) * height
        z1 = start + float(i + 1) * height
        for d, dom in enumerate(s["doms"]):
            if _layer_contains(dom[3], r, z0, z1):
                if table[i] != 0xFFFF:
                    return None
                table[i] = d
    return start, height, table


def _does_match_layering(s, start, height, n, r, table):
    """GeometrySource.cxx:273-342."""
    if n == 0 or r < 0: return False
    assigned = 0
    for i in range(n):
        z0 = start + float(i) * height
        z1 = start + float(i + 1) * height
        should = 0xFFFF
        for d, dom in enumerate(s["doms"]):
            if _layer_contains(dom[3], r, z0, z1):
                if should != 0xFFFF:
                    return False
                should = d
                assigned += 1
        if table[i] != should:
            return False
    return assigned == len(s["doms"])


def _c_short(v):
    """static_cast<short>(double) as x86-64 gcc does it: truncate toward zero;
    NaN (0/0 for perfectly straight strings) -> 0."""
    if math.isnan(v):
        return 0
    return int(v)        # python int() truncates toward zero


def geometry_from_text_file(filename, om_radius, string_min=1, string_max=2 ** 31 - 1, dom_min=1, dom_max=60):
    """I3CLSimSimpleGeometryTextFile (private/clsim/I3CLSimSimpleGeometryTextFile.cxx:43-100)."""
    tok = open(filename).read().split()
    sid, did, xs, ys, zs = [], [], [], [], []
    for k in range(0, len(tok) - 4, 5):
        try:
            s, d = int(tok[k]), int(tok[k + 1])
            x, y, z = float(tok[k + 2]), float(tok[k + 3]), float(tok[k + 4])
        except ValueError:
            break                                   # operator>> stops at the first malformed record
        if s < string_min or s > string_max or d < dom_min or d > dom_max:
            continue
        sid.append(s); did.append(d); xs.append(x); ys.append(y); zs.append(z)
    return dict(string_ids=np.array(sid, dtype=np.int32), dom_ids=np.array(did, dtype=np.uint32), x=np.array(xs),
                y=np.array(ys), z=np.array(zs), subdetectors=["default"] * len(sid), om_radius=om_radius)


def build_geometry(string_ids, dom_ids, pos_x, pos_y, pos_z, subdetectors, om_radius):
    """I3CLSimHelper::write_geometry_code_and_fill_buffer + generate_get_dom_position_code
    (GeometrySource.cxx:712-1275, 499-709).  Returns every constant the kernel
    sees as numpy arrays of the kernel's types plus the index->ID maps."""
    n = len(string_ids)
    assert n > 0 and om_radius >= 0
    keys = sorted(set((int(string_ids[i]), str(subdetectors[i])) for i in range(n)))
    subdet_names = sorted(set(str(s) for s in subdetectors))
    subdet_id = {name: i for i, name in enumerate(subdet_names)}
    by_key = {}
    for i in range(n):
        by_key.setdefault((int(string_ids[i]), str(subdetectors[i])), []).append(i)
    strings = []
    string_max_r = float("nan")
    for sid, sname in keys:
        s = dict(id=sid, sd=subdet_id[sname], doms=[], maxZ=float("nan"), minZ=float("nan"),
                 meanX=0.0, meanY=0.0, maxR=float("nan"))
        last_z = last_dz = float("nan")
        numdz = 0
        meandz = 0.0
        for i in by_key[(sid, sname)]:
            s["meanX"] += float(pos_x[i]); s["meanY"] += float(pos_y[i])
            z = float(pos_z[i])
            if (z > s["maxZ"]) or math.isnan(s["maxZ"]): s["maxZ"] = z
            if (z < s["minZ"]) or math.isnan(s["minZ"]): s["minZ"] = z
            if math.isnan(last_z):
                last_z = z
            else:
                dz = abs(last_z - z)
                last_z = z
                if not math.isnan(last_dz):
                    if dz < 1.75 * meandz / float(numdz):
                        meandz += dz; numdz += 1; last_dz = dz
                else:
                    last_dz = dz; meandz += dz; numdz += 1
            s["doms"].append((int(dom_ids[i]), float(pos_x[i]), float(pos_y[i]), z))
        nd = len(s["doms"])
        s["meanX"] /= float(nd); s["meanY"] /= float(nd)
        s["meandZ"] = meandz / float(numdz) if numdz else float("nan")
        for (_, x, y, _z) in s["doms"]:
            dx = s["meanX"] - x; dy = s["meanY"] - y
            r = math.sqrt(dx * dx + dy * dy) + om_radius
            if (r > s["maxR"]) or math.isnan(s["maxR"]): s["maxR"] = r
            if (r > string_max_r) or math.isnan(string_max_r): string_max_r = r
        strings.append(s)
    ns = len(strings)
    assert ns < 0xFFFF - 1

    # xy cells per subdetector (GeometrySource.cxx:913-949)
    cells = []
    for sd in range(len(subdet_names)):
        g = 1
        while True:
            res = _divide_into_cells(strings, sd, g)
            if res is not None:
                break
            g += 1
            if g >= 1000:
                raise RuntimeError("no x-y cell division")
        sx, sy, wx, wy, idx = res
        cells.append(dict(nx=g, ny=g, start_x=float_literal(sx), start_y=float_literal(sy),
                          width_x=float_literal(wx), width_y=float_literal(wy),
                          index=np.array(idx, dtype=np.uint16)))

    # z layers / string sets (GeometrySource.cxx:956-1091)
    set_n, set_start, set_height, set_table, in_set = [], [], [], [], []
    max_layers = 0
    for s in strings:
        match = None
        for k in range(len(set_n)):
            if _does_match_layering(s, set_start[k], set_height[k], set_n[k], om_radius, set_table[k]):
                match = k
                break
        if match is not None:
            in_set.append(match)
            continue
        in_set.append(len(set_n))
        if len(set_n) + 1 >= 0xFF:
            raise RuntimeError("more than 255 string sets")
        lo_hint = s["minZ"] - s["meandZ"] / 2.0
        hi_hint = s["maxZ"] + s["meandZ"] / 2.0
        n0 = int((s["maxZ"] - s["minZ"] + s["meandZ"]) / s["meandZ"])
        res = _divide_into_layers(s, n0, om_radius, lo_hint, hi_hint)
        nl = n0
        if res is None:
            nl = n0 + 1
            res = _divide_into_layers(s, nl, om_radius, lo_hint, hi_hint)
        if res is None:
            nl = 1
            while True:
                res = _divide_into_layers(s, nl, om_radius, lo_hint, hi_hint)
                if res is not None:
                    break
                nl += 1
                if nl >= 1000:
                    raise RuntimeError("no layer division for string")
        set_n.append(nl); set_start.append(res[0]); set_height.append(res[1]); set_table.append(res[2])
        if nl > max_layers: max_layers = nl
    nsets = len(set_n)
    flat = [0xFFFF] * (max_layers * nsets)
    for j in range(nsets):
        for i in range(set_n[j]):
            flat[j * max_layers + i] = set_table[j][i]
    bufsize = ((nsets * max_layers) // 64 + 1) * 64
    layer_to_om = np.full(bufsize, 0xFFFF, dtype=np.uint16)
    layer_to_om[:nsets * max_layers] = flat

    # DOM position templates (GeometrySource.cxx:499-709)
    mean_x = []; mean_y = []
    for s in strings:
        sx = sy = 0.0
        for (_, x, y, _z) in s["doms"]:
            sx += x; sy += y
        mean_x.append(sx / float(len(s["doms"]))); mean_y.append(sy / float(len(s["doms"])))
    eps = 1e-1 * 1e-3
    templates = []
    in_template = []
    for i, s in enumerate(strings):
        found = None
        for t, tpl in enumerate(templates):
            if len(tpl) != len(s["doms"]):
                continue
            ok = True
            for j, (_, x, y, z) in enumerate(s["doms"]):
                if abs(tpl[j][0] - (x - mean_x[i])) > eps: ok = False; break
                if abs(tpl[j][1] - (y - mean_y[i])) > eps: ok = False; break
                if abs(tpl[j][2] - z) > eps: ok = False; break
            if ok:
                found = t
                break
        if found is None:
            templates.append([(x - mean_x[i], y - mean_y[i], z) for (_, x, y, z) in s["doms"]])
            found = len(templates) - 1
        in_template.append(found)
    max_abs_x = max_abs_y = float("nan")
    flat_x, flat_y, flat_z, tpl_start = [], [], [], []
    for tpl in templates:
        tpl_start.append(len(flat_x))
        for (x, y, z) in tpl:
            flat_x.append(x); flat_y.append(y); flat_z.append(z)
            if (abs(x) > max_abs_x) or math.isnan(max_abs_x): max_abs_x = abs(x)
            if (abs(y) > max_abs_y) or math.isnan(max_abs_y): max_abs_y = abs(y)

    def _q(v, m):
        d = m / 32767.0
        return _c_short(v / d if d != 0.0 else float("nan"))

    geo = dict(
        num_strings=ns, om_radius=float_literal(om_radius), string_max_radius=float_literal(string_max_r),
        str_x=float_literals([s["meanX"] for s in strings]), str_y=float_literals([s["meanY"] for s in strings]),
        str_radius=float_literals([s["maxR"] for s in strings]),
        str_minz=float_literals([s["minZ"] for s in strings]), str_maxz=float_literals([s["maxZ"] for s in strings]),
        str_set=np.array(in_set, dtype=np.uint8),
        num_sets=nsets, max_layers=max_layers, set_nlayers=np.array(set_n, dtype=np.uint16),
        set_startz=float_literals(set_start), set_height=float_literals(set_height),
        layer_to_om=layer_to_om, cells=cells, subdetectors=subdet_names,
        max_dom_index=max(len(s["doms"]) for s in strings),
        dom_mul_x=float_literal(max_abs_x / 32767.0), dom_mul_y=float_literal(max_abs_y / 32767.0),
        dom_tx=np.array([_q(v, max_abs_x) for v in flat_x], dtype=np.int16),
        dom_ty=np.array([_q(v, max_abs_y) for v in flat_y], dtype=np.int16),
        dom_tz=float_literals(flat_z),
        dom_start=np.array([tpl_start[t] for t in in_template], dtype=np.uint32),
        dom_meanx=float_literals(mean_x), dom_meany=float_literals(mean_y),
        string_index_to_id=np.array([s["id"] for s in strings], dtype=np.int32),
        dom_index_to_id=[np.array([d[0] for d in s["doms"]], dtype=np.uint32) for s in strings],
    )
    return geo


# --------------------------------------------------------------------------
# RNG: safeprime multipliers + stream seeding
# --------------------------------------------------------------------------

_MR_BASES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)


def _is_prime(n):
    """Deterministic Miller-Rabin for n < 3.3e24 (the reference uses GMP's
    mpz_probab_prime_p, private/make_safeprimes/main.cxx:13-29)."""
    if n < 2:
        return False
    for p in _MR_BASES:
        if n % p == 0:
            return n == p
    d = n - 1
    r = 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in _MR_BASES:
        x = pow(a, d, n)
        if x == 1 or x == n - 1:
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _small_primes(limit):
    sieve = np.ones(limit + 1, dtype=bool)
    sieve[:2] = False
    for p in range(2, int(limit ** 0.5) + 1):
        if sieve[p]:
            sieve[p * p::p] = False
    return np.nonzero(sieve)[0]


def mwc_multipliers(count, start=4294967118):
    """private/make_safeprimes/main.cxx:31-104: multipliers a, descending from
    4294967118, with a*2^32-1 and (a*2^32-2)/2 both prime."""
    out = []
    primes = [int(p) for p in _small_primes(20000) if p > 2]
    # residues of a that make n2 = a*2^32-1 or n1 = a*2^31-1 divisible by p
    bad = []
    for p in primes:
        bad.append((p, pow(pow(2, 32, p), -1, p), pow(pow(2, 31, p), -1, p)))
    hi = start
    seg = 1 << 20
    while len(out) < count:
        lo = hi - seg + 1
        alive = np.ones(seg, dtype=bool)         # index k <-> a = hi - k
        for p, r2, r1 in bad:
            for r in (r2, r1):
                k0 = (hi - r) % p                 # smallest k with (hi-k) % p == r
                alive[k0::p] = False
        for k in np.nonzero(alive)[0]:
            a = hi - int(k)
            n2 = (a << 32) - 1
            if not _is_prime(n2):
                continue
            if not _is_prime((n2 - 1) >> 1):
                continue
            out.append(a)
            if len(out) >= count:
                break
        hi = lo - 1
    return np.array(out, dtype=np.uint32)


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def seed_streams(a, seed=12345):
    """init_MWC_RNG's state loop (private/opencl/mwcrng_init.h:105-113) with
    I3RandomService::Integer(0xffffffff) replaced by splitmix64: each call is
    (next() >> 32) % 0xffffffff (the random service is external to clsim)."""
    st = seed & 0xFFFFFFFFFFFFFFFF
    x = np.zeros(len(a), dtype=np.uint64)
    for i in range(len(a)):
        xi = 0
        ai = int(a[i])
        while (xi == 0) or ((xi >> 32) >= (ai - 1)) or ((xi & 0xFFFFFFFF) >= 0xFFFFFFFF):
            st, v = splitmix64(st)
            h = (v >> 32) % 0xFFFFFFFF
            st, v = splitmix64(st)
            l = (v >> 32) % 0xFFFFFFFF
            xi = (h << 32) + l
        x[i] = xi
    return x


# --------------------------------------------------------------------------
# photon tabulator (private/clsim/tabulator/)
# --------------------------------------------------------------------------

def linear_axis(lo, hi, n_bins):
    return dict(kind="linear", min=float(lo), max=float(hi), n_bins=int(n_bins), power=1)


def power_axis(lo, hi, n_bins, power=1):
    return dict(kind="power", min=float(lo), max=float(hi), n_bins=int(n_bins), power=int(power))


def axis_transform(ax, v):
    """Axis.cxx:93-97, 131-135."""
    return v if ax["kind"] == "linear" else math.pow(v, ax["power"])


def axis_inverse(ax, v):
    """Axis.cxx:99-103, 137-141."""
    return v if ax["kind"] == "linear" else math.pow(v, 1.0 / ax["power"])


def axis_index_literals(ax):
    """Axis::GetIndexCode (Axis.cxx:45-60): (scale, offset) as the float literals of the generated code."""
    scale = ax["n_bins"] / (axis_inverse(ax, ax["max"]) - axis_inverse(ax, ax["min"]))
    offset = scale * axis_inverse(ax, ax["min"])
    return float_literal(scale), float_literal(offset)


def axis_bin_edge(ax, i):
    """Axis::GetBinEdge (Axis.cxx:76-83)."""
    imin = axis_inverse(ax, ax["min"])
    imax = axis_inverse(ax, ax["max"])
    istep = (imax - imin) / ax["n_bins"]
    return axis_transform(ax, imin + i * istep)


def axis_bin_edges(ax):
    return np.array([axis_bin_edge(ax, i) for i in range(ax["n_bins"] + 1)])


def axes_layout(axes):
    """Axes::Axes (Axes.cxx:51-64): every axis has an under- and an overflow bin."""
    n = len(axes)
    shape = [0] * n
    strides = [0] * n
    shape[n - 1] = axes[n - 1]["n_bins"] + 2
    strides[n - 1] = 1
    for i in range(n - 2, -1, -1):
        shape[i] = axes[i]["n_bins"] + 2
        strides[i] = strides[i + 1] * shape[i + 1]
    return shape, strides, strides[0] * shape[0]


def bin_volume(kind, axes, idxs):
    """SphericalAxes / CylindricalAxes::GetBinVolume (Axes.cxx:118-133, 153-164)."""
    e = lambda k, i: axis_bin_edge(axes[k], i)
    if kind == "spherical":
        scalefactor = 1 if axes[1]["max"] > 180.0 else 2
        return ((math.pow(e(0, idxs[0] + 1), 3) - math.pow(e(0, idxs[0]), 3)) / 3.0) \
            * scalefactor * DEG * (e(1, idxs[1] + 1) - e(1, idxs[1])) * (e(2, idxs[2] + 1) - e(2, idxs[2]))
    return ((math.pow(e(0, idxs[0] + 1), 2) - math.pow(e(0, idxs[0]), 2)) / 2.0) \
        * 2 * (e(1, idxs[1] + 1) - e(1, idxs[1])) * (e(2, idxs[2] + 1) - e(2, idxs[2]))


def normalize_table(bins, kind, axes, step_length, dom_area):
    """I3CLSimStepToTableConverter::Normalize (StepToTableConverter.cxx:512-543): float bins divided by a double norm."""
    shape, strides, n = axes_layout(axes)
    out = np.array(bins, dtype=np.float32, copy=True)
    spatial = strides[2]
    for offset in range(0, n, spatial):
        idxs = [min(max((offset // strides[j]) % shape[j] - 1, 0), shape[j] - 3) for j in range(len(axes))]
        norm = bin_volume(kind, axes, idxs) / (step_length * dom_area)
        out[offset:offset + spatial] = (out[offset:offset + spatial].astype(np.float64) / norm).astype(np.float32)
    return out


def group_ref_index_host(medium, wlen):
    if "group_table" in medium:
        return from_table_host(medium["group_table"], wlen)
    x = wlen / MICROMETER                                   # RefIndexIceCube.cxx:84-101 ("group")
    n, g = medium["n"], medium["g"]
    np_ = n[0] + x * (n[1] + x * (n[2] + x * (n[3] + x * n[4])))
    corr = g[0] + x * (g[1] + x * (g[2] + x * (g[3] + x * g[4])))
    return np_ * corr


def phase_ref_index_of(medium, wlen):
    if "phase_table" in medium:
        return from_table_host(medium["phase_table"], wlen)
    return phase_ref_index_host(wlen, medium["n"])


def minimum_refractive_index(medium):
    """GetMinimumRefractiveIndex (StepToTableConverter.cxx:96-120), as written: the scan variable is
    wmin + i*(wmax-wmin) for i = 0..999, NOT divided by the number of points; (n_group, n_phase) of the smallest
    group index above 1."""
    best = (float("inf"), float("inf"))
    if medium.get("group_from_dispersion"):                 # :103-104
        raise ValueError("Medium properties don't know how to calculate group refractive indices")
    tabulated = "group_table" in medium
    gmin = medium["group_table"]["start"] if tabulated else -float("inf")
    gmax = (medium["group_table"]["start"] + medium["group_table"]["step"] * float(len(medium["group_table"]["values"]) - 1)) if tabulated else float("inf")
    wmin = max(medium["min_wlen"], gmin)
    wmax = min(medium["max_wlen"], gmax)
    for i in range(1000):                                   # identical for every layer
        w = wmin + i * (wmax - wmin)
        n = group_ref_index_host(medium, w)
        if n > 1 and n < best[0]:
            best = (n, phase_ref_index_of(medium, w))
    return best


def reference_particle(pos, time, direction):
    """I3CLSimReferenceParticle (StepToTableConverter.cxx:64-93): 12 floats posAndTime, dir, perpDir."""
    dx, dy, dz = (float(v) for v in direction)
    perpz = math.hypot(dx, dy)
    if perpz > 0.0:
        perp = (-dx * dz / perpz, -dy * dz / perpz, perpz)
        # I3Direction(x, y, z) normalises its arguments
        norm = math.sqrt(perp[0] ** 2 + perp[1] ** 2 + perp[2] ** 2)
        perp = tuple(v / norm for v in perp)
    else:
        perp = (1.0, 0.0, 0.0)
    return np.array([pos[0], pos[1], pos[2], time, dx, dy, dz, 0.0, perp[0], perp[1], perp[2], 0.0], dtype=np.float32)


def tabulator_config(kind, axes, medium, angular_coefficients, step_length=1.0, entries_per_stream=5000):
    """What tabulator/I3CLSimStepToTableConverter.cxx:178-207 and Axes::GenerateBinningCode put into the program."""
    assert kind in ("spherical", "cylindrical") and len(axes) in (4, 5)     # 5: TABULATE_IMPACT_ANGLE (:187-188)
    shape, strides, n_bins = axes_layout(axes)
    n_group, n_phase = minimum_refractive_index(medium)
    lit = [axis_index_literals(ax) for ax in axes]
    for ax in axes:
        assert ax["kind"] == "linear" or ax["power"] >= 1
    return dict(kind=kind, axes=axes, shape=shape, strides=strides, n_bins=n_bins,
                full_azimuth=(kind == "spherical" and axes[1]["max"] > 180.0),
                scale=[l[0] for l in lit], offset=[l[1] for l in lit],
                inverse=[(ax["power"] if ax["kind"] == "power" else 0) for ax in axes],
                inv_exp=[(float_literal(1.0 / ax["power"]) if ax["kind"] == "power" else 1.0) for ax in axes],
                max0=float_literal(axes[0]["max"]), max3=float_literal(axes[3]["max"]),
                min_inv_groupvel=float_literal(n_group / C_LIGHT),
                tan_thetac=float_literal(math.sqrt(n_phase * n_phase - 1.0)),
                n_group=n_group, n_phase=n_phase,
                volume_step=float_literal(step_length), entries_per_stream=int(entries_per_stream),
                angular=[float(c) for c in angular_coefficients])


# ---- step store and bunching (public/clsim/I3CLSimStepStore.h, I3CLSimLightSourceToStepConverterAsync.cxx) ----
class StepStoreModel:
    """I3CLSimStepStore restated with plain Python containers (StepStore.h:66-320)."""

    def __init__(self):
        self.bins = {}                      # photon count -> list (FIFO); the reference's vector of deques
        self.pending = {}
        self.n = 0

    def insert_copy(self, step):
        self.bins.setdefault(int(step["num"]), []).append(step.copy())           # :96-123
        self.pending[int(step["id"])] = self.pending.get(int(step["id"]), 0) + 1  # :276-281
        self.n += 1

    def size(self):
        return self.n

    def count(self, identifier):
        return self.pending.get(int(identifier), 0)                                # :308-312

    def pop_bunch_to_vector(self, size, fill=None):
        real = min(size, self.n)                                                   # :163-198
        out = []
        for key in sorted(self.bins):
            q = self.bins[key]
            while q and len(out) < real:
                s = q.pop(0)
                out.append(s)
                self.pending[int(s["id"])] -= 1                                    # :286-296
                if self.pending[int(s["id"])] == 0:
                    del self.pending[int(s["id"])]
            if len(out) >= real:
                break
        self.n -= len(out)
        if fill is not None:                                                       # :209-222, 298-306
            out.extend(fill.copy() for _ in range(size - len(out)))
        return out


def bunch_steps_model(sources, max_bunch_size, granularity, no_op):
    """The feeder thread's flushStepStore/emitStep (Async.cxx:209-273) for a list of (identifier, steps) light sources
    followed by a barrier: list of (steps, finished identifiers, last)."""
    store, markers, out = StepStoreModel(), [], []

    def full():
        while store.size() >= max_bunch_size:                                      # :212-232
            steps = store.pop_bunch_to_vector(max_bunch_size)
            finished = []
            while markers and store.count(markers[0]) == 0:
                finished.append(markers.pop(0))
            out.append((steps, finished, False))

    for identifier, steps in sources:
        markers.append(identifier)
        for s in steps:
            store.insert_copy(s)                                                   # emitStep :268-271
            full()
    full()
    n_fill = ((store.size() // granularity) + 1) * granularity if granularity > 1 else store.size()   # :256
    steps = store.pop_bunch_to_vector(n_fill, fill=no_op)
    assert store.size() == 0
    out.append((steps, list(markers), True))                                       # :262-269
    return out


def feeder_model(items, max_bunch_size, granularity, no_op):
    """The feeder thread's main loop with a parameterisation (Async.cxx:340-392 with getStepsFromParameterization :282-315 and
    flushStepStore :209-273).  items: (identifier, steps) light sources and None for a barrier.  The parameterisation hands
    its steps over in bunches of at most max_bunch_size; each is inserted whole, then the store is flushed; the marker of a
    light source is pushed AFTER its conversion (:388).  Returns the list of (steps, finished identifiers, last)."""
    store, markers, out = StepStoreModel(), [], []

    def flush(reset_barrier):
        while store.size() >= max_bunch_size:
            steps = store.pop_bunch_to_vector(max_bunch_size)
            finished = []
            while markers and store.count(markers[0]) == 0:
                finished.append(markers.pop(0))
            out.append((steps, finished, False))
        if reset_barrier:
            n_fill = ((store.size() // granularity) + 1) * granularity if granularity > 1 else store.size()
            steps = store.pop_bunch_to_vector(n_fill, fill=no_op)
            assert store.size() == 0
            out.append((steps, list(markers), True))
            del markers[:]

    for item in items:
        flush(item is None)
        if item is None:
            continue
        identifier, steps = item
        for lo in range(0, len(steps), max_bunch_size):
            for s in steps[lo:lo + max_bunch_size]:
                store.insert_copy(s)
            flush(False)
        markers.append(identifier)
    return out


# ---- flasher step producer: host logic (I3CLSimLightSourceToStepConverterFlasher.cxx:329-440,
#      python/I3CLSimRandomValueIceCubeFlasherTimeProfile.py) ----
FLASHER_PULSE_WIDTH15 = np.array([      # :52-88, the measured LED profile the reference tabulates (1 ns steps)
    1.18e-03, 2.769e-02, 1.2517e-01, 2.1484e-01, 3.2089e-01, 4.3239e-01, 4.6437e-01, 5.0023e-01, 4.3161e-01, 3.1621e-01, 2.2965e-01,
    1.3764e-01, 8.774e-02, 7.214e-02, 5.966e-02, 4.797e-02, 4.095e-02, 2.925e-02, 3.081e-02, 2.847e-02, 2.613e-02, 1.834e-02,
    1.834e-02, 1.99e-02, 1.288e-02, 1.288e-02, 1.288e-02, 1.6e-02, 1.444e-02, 1.678e-02, 7.42e-03, 6.64e-03, 9.76e-03, 1.132e-02,
    7.42e-03, 9.76e-03, 4.3e-03, 5.86e-03, 7.42e-03, 4.3e-03, 8.2e-03, 5.86e-03, 3.52e-03, 1.96e-03, 2.74e-03, 4.3e-03, 5.08e-03,
    2.74e-03, 3.52e-03, 4.3e-03, 2.74e-03])


def _pulse_narrow(x):
    """interp1d(kind='linear', bounds_error=False, fill_value=0.) of the adjusted table (:90-91)."""
    y = (FLASHER_PULSE_WIDTH15 - 0.00118) / 0.49905
    return np.interp(x, np.arange(51.0), y, left=0.0, right=0.0)


def flasher_time_profile(width_ns):
    """_the_pulse(numpy.linspace(0, 120, 240, endpoint=False), width*2) (:118-155)."""
    x = np.linspace(0.0, 120.0, 240, endpoint=False)
    fb = width_ns * 2.0
    if fb <= 15:
        return _pulse_narrow(x * (15.0 / fb))
    plateau = (fb - 15.0) * 59.5 / (124.0 - 15.0)
    rising = np.log(fb - 12.0) * 1.91 + 5.0
    rise = _pulse_narrow(np.clip(7.0 * x / rising, 0.0, 7.0))
    fall = _pulse_narrow(np.maximum(x - rising - plateau + 7.0, 7.0))
    return np.where(x <= rising, rise, np.where(x <= rising + plateau, 1.0, fall))


def interpolated_distribution_tables(spacing, y):
    """I3CLSimRandomValueInterpolatedDistribution::InitTables (InterpolatedDistribution.cxx:134-175) as float literals."""
    y = np.asarray(y, dtype=np.float64)
    acu = np.zeros(len(y))
    for j in range(1, len(y)):
        acu[j] = acu[j - 1] + spacing * (y[j] + y[j - 1]) / 2.0
    total = acu[-1]
    return (np.array([float_literal(v / total) for v in y], dtype=np.float32),
            np.array([float_literal(v / total) for v in acu], dtype=np.float32))


def flasher_make_steps_model(num_photons_with_bias, photons_per_step, max_bunch_size, granularity):
    """MakeSteps (Flasher.cxx:329-440) called until the pulse is used up: list of photon counts per output step
    (0 = dummy step), literally as the reference loops."""
    out = []
    left = int(num_photons_with_bias)
    while True:
        max_per_result = max_bunch_size * photons_per_step
        dummies = 0
        if left >= max_per_result:
            n_steps, in_last = max_bunch_size, photons_per_step
            left -= n_steps * photons_per_step
            done = (left == 0)
        else:
            if left <= photons_per_step:
                n_steps, in_last = 1, left
            else:
                n_steps, in_last = left // photons_per_step, left % photons_per_step
                if in_last > 0:
                    n_steps += 1
            done, left = True, 0
            modulo = n_steps % granularity
            if modulo > 0:
                dummies = granularity - modulo
        steps = []
        for i in range(n_steps):
            k = in_last if i == n_steps - 1 else photons_per_step
            if k == 0:
                dummies += 1
                continue
            steps.append(k)
        out.extend(steps + [0] * dummies)
        if done:
            return out
