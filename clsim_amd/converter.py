"""Host-side mirror of the reference's converter interface over the C ABI.

Names, argument meaning and error behaviour follow
  public/clsim/I3CLSimStepToPhotonConverter.h:67-192 and
  public/clsim/I3CLSimStepToPhotonConverterOpenCL.h:78-258;
the helpers follow python/MakeIceCubeMediumProperties.py,
python/GetIceCubeDOMAcceptance.py and I3CLSimModuleHelper.cxx:175-372.
Everything numerical happens inside libclsimhip.so (C++/HIP); this module only
marshals numpy arrays.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _lib
from .synthetic import PHOTON_DTYPE, STEP_DTYPE

NANOMETER = 1e-9


class I3CLSimStepToPhotonConverter_exception(RuntimeError):
    """public/clsim/I3CLSimStepToPhotonConverter.h:57-65"""

    def __init__(self, msg, code=0):
        RuntimeError.__init__(self, msg)
        self.code = code


def _check(rc, handle=None):
    if rc != _lib.OK:
        msg = _lib.load().clsimhip_last_error(handle)
        raise I3CLSimStepToPhotonConverter_exception((msg or b"").decode() or ("status %d" % rc), rc)


def _dp(a):
    return a.ctypes.data_as(_lib.DP)


class I3CLSimFunctionFromTable:
    """private/clsim/function/I3CLSimFunctionFromTable.cxx: (startWlen, wlenStep, values) :70-90, equal spacing, or
    (wlens, values) :57-70, the table's own wavelengths -- host side only, like the reference (:169-170): an emission
    spectrum for makeWavelengthGenerator."""

    def __init__(self, *args):
        if len(args) == 3:
            self.startWlen, self.wlenStep = float(args[0]), float(args[1])
            self.values = np.ascontiguousarray(args[2], dtype=np.float64)
            self.wlens = None
        elif len(args) == 2:
            self.wlens = np.ascontiguousarray(args[0], dtype=np.float64)
            self.values = np.ascontiguousarray(args[1], dtype=np.float64)
            if len(self.wlens) < 2:
                raise I3CLSimStepToPhotonConverter_exception("wlens must contain at least 2 elements!")
            if len(self.wlens) != len(self.values):
                raise I3CLSimStepToPhotonConverter_exception("wlens and values must have the same size!")
        else:
            raise TypeError("I3CLSimFunctionFromTable(startWlen, wlenStep, values) or (wlens, values)")

    def GetInEqualSpacingMode(self):
        return self.wlens is None

    def _desc(self):
        if self.wlens is None:
            return _lib.Function(0, len(self.values), self.startWlen, self.wlenStep, _dp(self.values), 0.0, None)
        return _lib.Function(2, len(self.values), 0.0, 0.0, _dp(self.values), 0.0, _dp(self.wlens))


class I3CLSimFunctionDeltaPeak:
    """function/I3CLSimFunctionDeltaPeak: a single-wavelength emission spectrum (the standard candles)"""

    def __init__(self, peakPosition):
        self.peakPosition = float(peakPosition)

    def GetPeakPosition(self):
        return self.peakPosition

    def _desc(self):
        return _lib.Function(3, 0, 0.0, 0.0, None, self.peakPosition, None)


class I3CLSimFunctionConstant:
    def __init__(self, value):
        self.value = float(value)

    def _desc(self):
        return _lib.Function(1, 0, 0.0, 0.0, None, self.value, None)


class I3CLSimRandomValueInterpolatedDistribution:
    """...InterpolatedDistribution.cxx: (xFirst, xSpacing, y) :57-74, or (x, y) :40-55."""

    def __init__(self, *args):
        if len(args) == 3:
            self.first, self.spacing = float(args[0]), float(args[1])
            self.y = np.ascontiguousarray(args[2], dtype=np.float64)
            self.x = None
        elif len(args) == 2:
            self.x = np.ascontiguousarray(args[0], dtype=np.float64)
            self.y = np.ascontiguousarray(args[1], dtype=np.float64)
            if len(self.x) != len(self.y):
                raise I3CLSimStepToPhotonConverter_exception('The "x" and "y" vectors must have the same size!')
        else:
            raise TypeError("I3CLSimRandomValueInterpolatedDistribution(xFirst, xSpacing, y) or (x, y)")

    def _desc(self):
        if self.x is None:
            return _lib.RandomValue(0, len(self.y), self.first, self.spacing, _dp(self.y), 0.0, None)
        return _lib.RandomValue(3, len(self.y), 0.0, 0.0, _dp(self.y), 0.0, _dp(self.x))


class I3CLSimRandomValueWlenCherenkovNoDispersion:
    """random_value/I3CLSimRandomValueWlenCherenkovNoDispersion.cxx:40-98: 1/lambda uniform between 1/toWlen and 1/fromWlen."""

    def __init__(self, fromWlen, toWlen):
        self.fromWlen, self.toWlen = float(fromWlen), float(toWlen)

    def _desc(self):
        return _lib.RandomValue(2, 0, self.fromWlen, self.toWlen, None, 0.0, None)


class I3CLSimRandomValueConstant:
    def __init__(self, value):
        self.value = float(value)

    def _desc(self):
        return _lib.RandomValue(1, 0, 0.0, 0.0, None, self.value, None)


class I3CLSimMediumProperties:
    """Opaque medium object living in the library (clsimhip_medium)."""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep

    def __del__(self):
        try:
            if self._h:
                _lib.load().clsimhip_medium_destroy(self._h)
        except Exception:
            pass

    def describe(self):
        d = _lib.MediumDesc()
        _check(_lib.load().clsimhip_medium_describe(self._h, C.byref(d)))
        nl = d.num_layers

        def arr(p, n):
            return np.ctypeslib.as_array(p, shape=(n,)).copy() if (p and n) else np.zeros(0)
        out = {k: getattr(d, k) for k in ("num_layers", "layers_z_start", "layers_height", "min_wavelength",
                                          "max_wavelength", "lengths_kind", "alpha", "kappa", "A", "B", "D", "E",
                                          "scatter_kind", "liu_fraction", "mean_cosine", "has_anisotropy",
                                          "aniso_azimuth", "aniso_k1", "aniso_k2", "has_pre_transform",
                                          "pre_renormalize", "has_post_transform", "post_renormalize", "has_tilt",
                                          "tilt_azimuth")}
        out["n"] = list(d.n); out["g"] = list(d.g)
        out["pre_matrix"] = np.array(list(d.pre_matrix)).reshape(3, 3)
        out["post_matrix"] = np.array(list(d.post_matrix)).reshape(3, 3)
        out["phase_index_kind"], out["group_index_kind"] = d.phase_index_kind, d.group_index_kind
        for key in ("phase_index_table", "group_index_table"):
            f = getattr(d, key)
            if getattr(d, key.replace("_table", "_kind")) == 1:
                out[key] = dict(start=f.start, step=f.step, values=arr(f.values, f.n))
        if d.lengths_kind == 0:
            out["abs_length"] = arr(d.abs_length, nl); out["sca_length"] = arr(d.sca_length, nl)
        elif d.lengths_kind == 2:
            nw = d.table_num_wavelengths
            out.update(table_num_wavelengths=nw, table_start_wavelength=d.table_start_wavelength,
                       table_wavelength_step=d.table_wavelength_step, table_store_as_16bit=bool(d.table_store_as_16bit))
            out["abs_length_table"] = arr(d.abs_length_table, nl * nw).reshape(nl, nw)
            out["sca_length_table"] = arr(d.sca_length_table, nl * nw).reshape(nl, nw)
        else:
            out["a_dust400"] = arr(d.a_dust400, nl); out["delta_tau"] = arr(d.delta_tau, nl); out["b400"] = arr(d.b400, nl)
        if d.has_tilt:
            nd, nz = d.tilt_num_distances, d.tilt_num_z
            out["tilt_distances"] = arr(d.tilt_distances, nd)
            out["tilt_z_coordinates"] = arr(d.tilt_z_coordinates, nz)
            out["tilt_z_corrections"] = arr(d.tilt_z_corrections, nd * nz).reshape(nd, nz)
        return out


def MakeIceCubeMediumProperties(detectorCenterDepth=1948.07, iceDataDirectory=None, useTiltIfAvailable=True):
    """python/MakeIceCubeMediumProperties.py:49-256 (PPC ice tables -> medium)."""
    h = C.c_void_p()
    _check(_lib.load().clsimhip_medium_create_from_ppc(str(iceDataDirectory).encode(), float(detectorCenterDepth),
                                                        1 if useTiltIfAvailable else 0, C.byref(h)))
    return I3CLSimMediumProperties(h)


def MakeIceCubeMediumPropertiesPhotonics(tableFile, detectorCenterDepth=1948.07):
    """python/MakeIceCubeMediumPropertiesPhotonics.py:47-227 (photonics ice table -> medium)."""
    h = C.c_void_p()
    _check(_lib.load().clsimhip_medium_create_from_photonics(str(tableFile).encode(), float(detectorCenterDepth), C.byref(h)))
    return I3CLSimMediumProperties(h)


def MakeHomogeneousMediumProperties(absLen=100.0, scaLen=25.0, zStart=-1000.0, height=2000.0, meanCosine=0.9,
                                    liuFraction=0.45):
    """BASELINE config C1: one layer with I3CLSimFunctionConstant absorption /
    scattering lengths, IceCube refractive index (SURVEY.md 9.7 option i)."""
    d = _lib.MediumDesc()
    d.num_layers = 1
    d.layers_z_start, d.layers_height = zStart, height
    d.min_wavelength, d.max_wavelength = 265.0 * NANOMETER, 675.0 * NANOMETER
    d.lengths_kind = 0
    a = np.array([absLen], dtype=np.float64); s = np.array([scaLen], dtype=np.float64)
    d.abs_length, d.sca_length = _dp(a), _dp(s)
    for i, v in enumerate((1.55749, -1.57988, 3.99993, -4.68271, 2.09354)):
        d.n[i] = v
    for i, v in enumerate((1.227106, -0.954648, 1.42568, -0.711832, 0.0)):
        d.g[i] = v
    d.scatter_kind = 2
    d.liu_fraction, d.mean_cosine = liuFraction, meanCosine
    h = C.c_void_p()
    _check(_lib.load().clsimhip_medium_create(C.byref(d), C.byref(h)))
    return I3CLSimMediumProperties(h)


def GetIceCubeDOMAcceptance(domRadius=0.16510, efficiency=1.0):
    """python/GetIceCubeDOMAcceptance.py:35-115."""
    vals = np.zeros(43, dtype=np.float64)
    start, step = C.c_double(), C.c_double()
    _check(_lib.load().clsimhip_icecube_dom_acceptance(domRadius, efficiency, _dp(vals), C.byref(start), C.byref(step)))
    return I3CLSimFunctionFromTable(start.value, step.value, vals)


def makeCherenkovWavelengthGenerator(wavelengthGenerationBias, mediumProperties):
    """I3CLSimModuleHelper::makeCherenkovWavelengthGenerator (ModuleHelper.cxx:175-263)."""
    y = np.zeros(len(wavelengthGenerationBias.values), dtype=np.float64)
    first, spacing = C.c_double(), C.c_double()
    desc = wavelengthGenerationBias._desc()
    _check(_lib.load().clsimhip_make_cherenkov_wlen_generator(C.byref(desc), mediumProperties._h, _dp(y),
                                                               C.byref(first), C.byref(spacing)))
    return I3CLSimRandomValueInterpolatedDistribution(first.value, spacing.value, y)


def makeWavelengthGenerator(unbiasedSpectrum, wavelengthGenerationBias, mediumProperties):
    """I3CLSimModuleHelper::makeWavelengthGenerator (ModuleHelper.cxx:73-171): a delta peak becomes a constant, a tabulated
    spectrum an InterpolatedDistribution on the table's own binning with the bias folded in."""
    n = max(len(getattr(unbiasedSpectrum, "values", ())), 1)
    x = np.zeros(n, dtype=np.float64)
    y = np.zeros(n, dtype=np.float64)
    out = _lib.RandomValue()
    spectrum, bias = unbiasedSpectrum._desc(), wavelengthGenerationBias._desc()
    _check(_lib.load().clsimhip_make_wlen_generator(C.byref(spectrum), C.byref(bias), mediumProperties._h, C.byref(out), _dp(x), _dp(y), n))
    if out.kind == 1:
        return I3CLSimRandomValueConstant(out.value)
    if out.kind == 3:
        return I3CLSimRandomValueInterpolatedDistribution(x[:out.n].copy(), y[:out.n].copy())
    return I3CLSimRandomValueInterpolatedDistribution(out.first, out.spacing, y[:out.n].copy())


FLASHER_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "flasher_data")
_FLASHER_LED_SPECTRA = {
    # python/GetIceCubeFlasherSpectrum.py:38-60: file under resources/flasher_data/, normalisation constant
    "LED340nm": ("flasher_led_340nm_emission_spectrum_cw_measured_20mA_pulseCurrent.txt", 24.306508),
    "LED370nm": ("flasher_led_370nm_emission_spectrum_cw_measured.txt", 15.7001863),
    "LED405nm": ("flasher_led_405nm_emission_spectrum_datasheet.txt", 8541585.10324),
    "LED450nm": ("flasher_led_450nm_emission_spectrum_datasheet.txt", 21.9792812618),
    "LED505nm": ("flasher_led_505nm_emission_spectrum_cw_measured.txt", 38.1881),
}


def GetIceCubeFlasherSpectrumData(spectrumType):
    """python/GetIceCubeFlasherSpectrum.py:38-70: (wavelengths [m], values) of an LED's emission spectrum"""
    if spectrumType not in _FLASHER_LED_SPECTRA:
        raise RuntimeError("invalid spectrumType")
    name, norm = _FLASHER_LED_SPECTRA[spectrumType]
    data = np.loadtxt(os.path.join(FLASHER_DATA, name), unpack=True)
    data[0] *= NANOMETER
    data[1] /= norm
    return data


def GetIceCubeFlasherSpectrum(spectrumType="LED405nm"):
    """python/GetIceCubeFlasherSpectrum.py:72-82; spectrumType: 'LED340nm' ... 'LED505nm', 'SC1', 'SC2'
    (I3CLSimFlasherPulse::FlasherPulseType)"""
    if spectrumType in ("SC1", "SC2"):
        return I3CLSimFunctionDeltaPeak(337.0 * NANOMETER)
    data = GetIceCubeFlasherSpectrumData(spectrumType)
    return I3CLSimFunctionFromTable(data[0], data[1])


def mwc_multipliers(count):
    a = np.zeros(count, dtype=np.uint32)
    _check(_lib.load().clsimhip_mwc_multipliers(a.ctypes.data_as(C.c_void_p), count))
    return a


def seed_streams(a, seed=12345):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    x = np.zeros(len(a), dtype=np.uint64)
    _check(_lib.load().clsimhip_seed_streams(a.ctypes.data_as(C.c_void_p), len(a), seed, x.ctypes.data_as(C.c_void_p)))
    return x


class I3CLSimSimpleGeometry:
    """public/clsim/I3CLSimSimpleGeometry.h: parallel per-DOM arrays."""

    def __init__(self, string_ids, dom_ids, x, y, z, subdetectors, om_radius):
        self.string_ids = np.ascontiguousarray(string_ids, dtype=np.int32)
        self.dom_ids = np.ascontiguousarray(dom_ids, dtype=np.uint32)
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self.z = np.ascontiguousarray(z, dtype=np.float64)
        self.subdetectors = [str(s) for s in subdetectors]
        self.om_radius = float(om_radius)

    @classmethod
    def from_text_file(cls, OMRadius, filename, ignoreStringIDsSmallerThan=1, ignoreStringIDsLargerThan=2 ** 31 - 1,
                       ignoreDomIDsSmallerThan=1, ignoreDomIDsLargerThan=60):
        """I3CLSimSimpleGeometryTextFile (private/clsim/I3CLSimSimpleGeometryTextFile.cxx:43-100); parsing happens in
        the library when the geometry is set (clsimhip_set_geometry_from_text_file)."""
        g = cls([], [], [], [], [], [], OMRadius)
        g.text_file = (str(filename), int(ignoreStringIDsSmallerThan), int(ignoreStringIDsLargerThan),
                       int(ignoreDomIDsSmallerThan), int(ignoreDomIDsLargerThan))
        return g

    @classmethod
    def from_dict(cls, g):
        return cls(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])


class I3CLSimStepToPhotonConverterHIP:
    """MI355X implementation of I3CLSimStepToPhotonConverter."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        _check(self._lib.clsimhip_create(int(device), C.byref(self._h)))
        self._history_entries = 0

    def __del__(self):
        try:
            if self._h:
                self._lib.clsimhip_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _call(self, name, *args):
        _check(getattr(self._lib, name)(self._h, *args), self._h)

    # ---- configuration ----
    def SetWlenGenerators(self, wlenGenerators):
        descs = (_lib.RandomValue * len(wlenGenerators))(*[g._desc() for g in wlenGenerators])
        self._call("clsimhip_set_wlen_generators", descs, len(wlenGenerators))

    def SetWlenBias(self, wlenBias):
        d = wlenBias._desc()
        self._call("clsimhip_set_wlen_bias", C.byref(d))

    def SetMediumProperties(self, mediumProperties):
        self._call("clsimhip_set_medium_properties", mediumProperties._h)

    def SetGeometry(self, geometry):
        g = geometry
        if getattr(g, "text_file", None):
            fn, smin, smax, dmin, dmax = g.text_file
            self._call("clsimhip_set_geometry_from_text_file", fn.encode(), g.om_radius, smin, smax, dmin, dmax)
            return
        names = (C.c_char_p * len(g.subdetectors))(*[s.encode() for s in g.subdetectors])
        self._call("clsimhip_set_geometry", len(g.string_ids), g.string_ids.ctypes.data_as(C.c_void_p),
                   g.dom_ids.ctypes.data_as(C.c_void_p), g.x.ctypes.data_as(C.c_void_p),
                   g.y.ctypes.data_as(C.c_void_p), g.z.ctypes.data_as(C.c_void_p), names, g.om_radius)

    def SetEnableDoubleBuffering(self, v): self._call("clsimhip_set_enable_double_buffering", int(bool(v)))
    def SetDoublePrecision(self, v): self._call("clsimhip_set_double_precision", int(bool(v)))
    def SetStopDetectedPhotons(self, v): self._call("clsimhip_set_stop_detected_photons", int(bool(v)))
    def SetSaveAllPhotons(self, v): self._call("clsimhip_set_save_all_photons", int(bool(v)))
    def SetSaveAllPhotonsPrescale(self, v): self._call("clsimhip_set_save_all_photons_prescale", float(v))
    def SetFixedNumberOfAbsorptionLengths(self, v): self._call("clsimhip_set_fixed_number_of_absorption_lengths", float(v))
    def SetDOMPancakeFactor(self, v): self._call("clsimhip_set_dom_pancake_factor", float(v))
    def SetPhotonHistoryEntries(self, v):
        self._call("clsimhip_set_photon_history_entries", int(v))
        self._history_entries = int(v)
    def SetWorkgroupSize(self, v): self._call("clsimhip_set_workgroup_size", int(v))
    def SetMaxNumWorkitems(self, v): self._call("clsimhip_set_max_num_workitems", int(v))

    def Compile(self): self._call("clsimhip_compile")

    def GetMaxWorkgroupSize(self):
        v = C.c_size_t()
        self._call("clsimhip_get_max_workgroup_size", C.byref(v))
        return v.value

    def Initialize(self, seed=12345):
        self._call("clsimhip_initialize", int(seed))

    def InitializeWithStreams(self, x, a):
        x = np.ascontiguousarray(x, dtype=np.uint64); a = np.ascontiguousarray(a, dtype=np.uint32)
        self._call("clsimhip_initialize_with_streams", x.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), len(x))

    def IsInitialized(self):
        return bool(self._lib.clsimhip_is_initialized(self._h))

    # ---- steady state ----
    def EnqueueSteps(self, steps, identifier):
        if steps is None:
            raise I3CLSimStepToPhotonConverter_exception("Steps pointer is (null)!", _lib.ERR_ARGUMENT)
        steps = np.ascontiguousarray(steps, dtype=STEP_DTYPE)
        self._call("clsimhip_enqueue_steps", steps.ctypes.data_as(C.c_void_p), len(steps), int(identifier))

    def GetConversionResult(self, with_histories=False, out=None):
        """ConversionResult_t (I3CLSimStepToPhotonConverter.h:70-90): (identifier, photons), plus with
        with_histories=True the photonHistories as a list of [k_i, 4] arrays (k_i = min(numScatters_i,
        PhotonHistoryEntries); None when no histories are recorded).  The photons are copied out of the library's buffer
        (like the C++ adapter copies them into the I3CLSimPhotonSeries it hands to the caller): into `out`, a PHOTON_DTYPE
        array the caller recycles, when it is given and large enough -- a view of it is returned."""
        # a recycled buffer is written through its raw address: it must be exactly what the records are -- checked before a
        # result is taken, whatever that result holds (the contract does not depend on the data)
        if out is not None and not (isinstance(out, np.ndarray) and out.dtype == PHOTON_DTYPE and out.ndim == 1 and out.flags.c_contiguous
                                    and out.flags.writeable):
            raise ValueError("GetConversionResult(out=...): a writeable, C-contiguous one-dimensional array of PHOTON_DTYPE (80-byte records) is required")
        ident, ptr, n = C.c_uint32(), C.c_void_p(), C.c_size_t()
        self._call("clsimhip_get_conversion_result", C.byref(ident), C.byref(ptr), C.byref(n))
        histories = None
        if n.value:
            buf = (C.c_char * (n.value * 80)).from_address(ptr.value)
            if out is not None and len(out) >= n.value:
                C.memmove(out.ctypes.data, ptr.value, n.value * 80)
                photons = out[:n.value]
            else:
                photons = np.frombuffer(buf, dtype=PHOTON_DTYPE).copy()
            if with_histories:
                hp, entries = C.POINTER(C.c_float)(), C.c_uint32()
                self._call("clsimhip_get_result_histories", ptr, C.byref(hp), C.byref(entries))
                if hp and entries.value:
                    flat = np.ctypeslib.as_array(hp, shape=(n.value, entries.value, 4)).copy()
                    histories = [flat[i, :min(int(photons["numScatters"][i]), entries.value)] for i in range(n.value)]
            self._call("clsimhip_release_result", ptr)
        else:
            photons = np.zeros(0, dtype=PHOTON_DTYPE)
            if with_histories and self._history_entries:
                histories = []
        return (ident.value, photons, histories) if with_histories else (ident.value, photons)

    def GetConversionResultInPlace(self):
        """(identifier, photons, release): `photons` is a read-only view of the library's page-locked result buffer -- what
        a C or C++ consumer that works on the records where they are gets from clsimhip_get_conversion_result -- valid until
        `release()` is called (clsimhip_release_result), which the caller must do"""
        ident, ptr, n = C.c_uint32(), C.c_void_p(), C.c_size_t()
        self._call("clsimhip_get_conversion_result", C.byref(ident), C.byref(ptr), C.byref(n))
        if not n.value:
            return ident.value, np.zeros(0, dtype=PHOTON_DTYPE), (lambda: None)
        buf = (C.c_char * (n.value * 80)).from_address(ptr.value)
        view = np.frombuffer(buf, dtype=PHOTON_DTYPE)
        view.flags.writeable = False
        return ident.value, view, (lambda: self._call("clsimhip_release_result", ptr))

    def _size(self, name):
        v = C.c_size_t()
        self._call(name, C.byref(v))
        return v.value

    def GetWorkgroupSize(self): return self._size("clsimhip_get_workgroup_size")
    def GetMaxNumWorkitems(self): return self._size("clsimhip_get_max_num_workitems")
    def QueueSize(self): return self._size("clsimhip_queue_size")

    def MorePhotonsAvailable(self):
        v = C.c_int()
        self._call("clsimhip_more_photons_available", C.byref(v))
        return bool(v.value)

    def GetStatistics(self):
        out = (C.c_double * 8)()
        self._call("clsimhip_get_statistics", out)
        keys = ["TotalDeviceTime", "TotalHostTime", "NumKernelCalls", "TotalNumPhotonsGenerated",
                "TotalNumPhotonsAtDOMs", "AverageDeviceTimePerPhoton", "AverageHostTimePerPhoton", "DeviceUtilization"]
        return dict(zip(keys, list(out)))

    # the accessors of the concrete class (OpenCL.h:138-258, :377-381; times in nanoseconds)
    def GetTotalDeviceTime(self): return self.GetStatistics()["TotalDeviceTime"]
    def GetTotalHostTime(self): return self.GetStatistics()["TotalHostTime"]
    def GetNumKernelCalls(self): return int(self.GetStatistics()["NumKernelCalls"])
    def GetTotalNumPhotonsGenerated(self): return int(self.GetStatistics()["TotalNumPhotonsGenerated"])
    def GetTotalNumPhotonsAtDOMs(self): return int(self.GetStatistics()["TotalNumPhotonsAtDOMs"])

    # ---- tuning (include/clsimhip.h: clsimhip_set_tuning; no result depends on it) ----
    _KERNELS = {"auto": 0, "pool": 1, "classic": 2}

    def SetTuning(self, key, value):
        """clsimhip_set_tuning(key, value); "kernel" also takes "auto" | "pool" | "classic"."""
        if key == "kernel" and isinstance(value, str):
            value = self._KERNELS[value]
        self._call("clsimhip_set_tuning", key.encode(), int(value))

    def GetTuning(self, key):
        v = C.c_longlong()
        self._call("clsimhip_get_tuning", key.encode(), C.byref(v))
        return v.value

    def _option(self, which):
        v = C.c_double()
        self._call("clsimhip_get_option", int(which), C.byref(v))
        return v.value

    def GetEnableDoubleBuffering(self): return self._option(0) != 0.0
    def GetDoublePrecision(self): return self._option(1) != 0.0
    def GetStopDetectedPhotons(self): return self._option(2) != 0.0
    def GetSaveAllPhotons(self): return self._option(3) != 0.0
    def GetSaveAllPhotonsPrescale(self): return self._option(4)
    def GetFixedNumberOfAbsorptionLengths(self): return self._option(5)
    def GetDOMPancakeFactor(self): return self._option(6)
    def GetPhotonHistoryEntries(self): return int(self._option(7))

    # ---- device-resident path / introspection ----
    def PropagateDevice(self, d_steps, n, d_photons, capacity, d_hit_count, stream=0, rng_offset=0):
        self._call("clsimhip_propagate_device", C.c_void_p(d_steps), int(n), int(rng_offset), C.c_void_p(d_photons),
                   int(capacity), C.c_void_p(d_hit_count), C.c_void_p(stream))

    def SetConcurrentDeviceLaunches(self, k):
        """k device-path launches in flight on k streams (disjoint rng_offset ranges): each sizes its grid for 1/k of the chip"""
        self._call("clsimhip_set_concurrent_device_launches", int(k))

    def ReplaceIndicesWithIDs(self, photons):
        photons = np.ascontiguousarray(photons, dtype=PHOTON_DTYPE)
        self._call("clsimhip_replace_indices_with_ids", photons.ctypes.data_as(C.c_void_p), len(photons))
        return photons

    def KernelTimeMs(self, reset=False):
        total, launches = C.c_double(), C.c_uint64()
        self._call("clsimhip_kernel_time_ms", int(bool(reset)), C.byref(total), C.byref(launches))
        return total.value, launches.value

    def GetTable(self, name):
        n = self._lib.clsimhip_get_table(self._h, name.encode(), None, 0)
        if n < 0:
            _check(int(n), self._h)
        out = np.zeros(n, dtype=np.float64)
        self._lib.clsimhip_get_table(self._h, name.encode(), _dp(out), n)
        return out

    def KernelForBunch(self, n_steps):
        """'pool' or 'classic': the scheduling the propagation kernel runs with for a bunch of n_steps steps"""
        v = C.c_int32()
        self._call("clsimhip_kernel_for_bunch", int(n_steps), C.byref(v))
        return "pool" if v.value else "classic"

    def UsesPooledKernel(self):
        v = C.c_int32()
        self._call("clsimhip_uses_pooled_kernel", C.byref(v))
        return bool(v.value)

    # ---- the reference's tester classes (private/test/I3CLSim*Tester): single functions evaluated on the device ----
    EVAL = {"lengths": 0, "refraction": 1, "wavelength_bias": 2, "tilt": 3, "abs_len_scaling": 4, "pre_scatter_transform": 5,
            "post_scatter_transform": 6}
    EVAL_RANDOM = {"uniform": 0, "wavelength": 1, "scattering_cosine": 2}

    def EvaluateOnDevice(self, what, values, layer=0, fast=False):
        """what: 'lengths' (values = wavelengths -> columns absorption, scattering length of `layer`), 'refraction' (-> phase
        index, group velocity), 'wavelength_bias', 'tilt' (values = positions (n, 3)), 'abs_len_scaling', 'pre_scatter_transform',
        'post_scatter_transform' (values = directions (n, 3) -> (n, 3)).  Returns float32 (n, 4)."""
        v = np.asarray(values, dtype=np.float32)
        inp = np.zeros((len(v), 4), dtype=np.float32)
        if v.ndim == 1:
            inp[:, 0] = v
        else:
            inp[:, :v.shape[1]] = v
        out = np.zeros_like(inp)
        self._call("clsimhip_eval_device_function", self.EVAL[what], int(layer), int(bool(fast)), inp.ctypes.data_as(C.c_void_p), len(inp),
                   out.ctypes.data_as(C.c_void_p))
        return out

    def SampleOnDevice(self, what, x, a, draws, generator=0, fast=False):
        """what: 'uniform', 'wavelength' (of `generator`), 'scattering_cosine'; one work item per stream (x[i], a[i]), `draws`
        values each.  Returns (values (n_streams, draws), final stream states)."""
        xs = np.ascontiguousarray(x, dtype=np.uint64).copy()
        a32 = np.ascontiguousarray(a, dtype=np.uint32)
        out = np.zeros((len(xs), int(draws)), dtype=np.float32)
        self._call("clsimhip_eval_device_random", self.EVAL_RANDOM[what], int(generator), int(bool(fast)), xs.ctypes.data_as(C.c_void_p),
                   a32.ctypes.data_as(C.c_void_p), len(xs), int(draws), out.ctypes.data_as(C.c_void_p))
        return out, xs

    def GetRNGState(self, count):
        x = np.zeros(count, dtype=np.uint64)
        self._call("clsimhip_get_rng_state", x.ctypes.data_as(C.c_void_p), count)
        return x


def initializeHIP(device, geometry, medium, wavelengthGenerationBias, wavelengthGenerators,
                  enableDoubleBuffering=False, doublePrecision=False, stopDetectedPhotons=True, saveAllPhotons=False,
                  saveAllPhotonsPrescale=0.01, fixedNumberOfAbsorptionLengths=float("nan"), pancakeFactor=1.0,
                  photonHistoryEntries=0, limitWorkgroupSize=0, approximateNumberOfWorkItems=262144,
                  seed=12345, streams=None, tuning=None):
    """Canonical configuration sequence, I3CLSimModuleHelper::initializeOpenCL
    (ModuleHelper.cxx:303-372).  tuning: {key: value} for clsimhip_set_tuning, applied before Compile()."""
    conv = I3CLSimStepToPhotonConverterHIP(device)
    for key, value in (tuning or {}).items():
        conv.SetTuning(key, value)
    conv.SetWlenGenerators(wavelengthGenerators)
    conv.SetWlenBias(wavelengthGenerationBias)
    conv.SetMediumProperties(medium)
    conv.SetGeometry(geometry)
    conv.SetEnableDoubleBuffering(enableDoubleBuffering)
    conv.SetDoublePrecision(doublePrecision)
    conv.SetStopDetectedPhotons(stopDetectedPhotons)
    conv.SetSaveAllPhotons(saveAllPhotons)
    conv.SetSaveAllPhotonsPrescale(saveAllPhotonsPrescale)
    conv.SetFixedNumberOfAbsorptionLengths(fixedNumberOfAbsorptionLengths)
    conv.SetDOMPancakeFactor(pancakeFactor)
    conv.SetPhotonHistoryEntries(photonHistoryEntries)
    conv.Compile()
    max_wg = conv.GetMaxWorkgroupSize()
    if limitWorkgroupSize:
        max_wg = min(limitWorkgroupSize, max_wg)
    conv.SetWorkgroupSize(max_wg)
    wg = max_wg
    max_items = (int(approximateNumberOfWorkItems) // wg) * wg
    if max_items == 0:
        max_items = wg
    conv.SetMaxNumWorkitems(max_items)
    if streams is not None:
        conv.InitializeWithStreams(streams[0][:max_items], streams[1][:max_items])
    else:
        conv.Initialize(seed)
    return conv


# ---- step producer on the GPU (clsimhip_generate_steps*, csrc/steps_kernel.hip) ----
REQUEST_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("time", "<f4"), ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"),
                          ("length", "<f4"), ("pa", "<f4"), ("pb", "<f4"), ("kind", "<u4"), ("identifier", "<u4"),
                          ("photons_per_step", "<u4"), ("num_photons_in_last_step", "<u4"), ("num_steps", "<u8")])
STEPS_CASCADE, STEPS_MUON_CASCADE, STEPS_MUON = 0, 1, 2


def CountGeneratedSteps(requests, granularity=1):
    req = np.ascontiguousarray(requests, dtype=REQUEST_DTYPE)
    steps, padded = C.c_size_t(), C.c_size_t()
    _check(_lib.load().clsimhip_count_generated_steps(req.ctypes.data_as(C.POINTER(_lib.StepRequest)), len(req), int(granularity),
                                                      C.byref(steps), C.byref(padded)))
    return steps.value, padded.value


def GenerateSteps(requests, seed, granularity=1, device=0):
    """Steps of a list of step requests (the reference's CascadeStepData_t / MuonStepData_t queue entries,
    I3CLSimLightSourceToStepConverterPPC.cxx:524-551, 785-842), generated on the GPU, as a host array."""
    req = np.ascontiguousarray(requests, dtype=REQUEST_DTYPE)
    _, padded = CountGeneratedSteps(req, granularity)
    out = np.zeros(padded, dtype=STEP_DTYPE)
    got = C.c_size_t()
    _check(_lib.load().clsimhip_generate_steps(int(device), req.ctypes.data_as(C.POINTER(_lib.StepRequest)), len(req), int(seed),
                                               int(granularity), out.ctypes.data_as(C.c_void_p), len(out), C.byref(got)))
    return out


def GenerateStepsDevice(requests, seed, d_steps, capacity, granularity=1, device=0, stream=0):
    """The same into device memory (address d_steps, room for `capacity` steps); returns the padded step count."""
    req = np.ascontiguousarray(requests, dtype=REQUEST_DTYPE)
    got = C.c_size_t()
    _check(_lib.load().clsimhip_generate_steps_device(int(device), req.ctypes.data_as(C.POINTER(_lib.StepRequest)), len(req), int(seed),
                                                      int(granularity), C.c_void_p(d_steps), int(capacity), C.c_void_p(stream), C.byref(got)))
    return got.value


# ---- particle -> step requests (I3CLSimLightSourceToStepConverterPPC front end, csrc/lightsource.cpp) ----
PARTICLE_DTYPE = np.dtype([("type", "<i4"), ("shape", "<i4"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("time", "<f8"),
                           ("dx", "<f8"), ("dy", "<f8"), ("dz", "<f8"), ("energy", "<f8"), ("length", "<f8"),
                           ("identifier", "<u4"), ("reserved", "<u4")])
assert PARTICLE_DTYPE.itemsize == 88


class ParticleType:
    """I3Particle::ParticleType values (dataclasses): PDG codes and IceCube's codes for stochastic losses"""
    Gamma, EMinus, EPlus, MuMinus, MuPlus, TauMinus, TauPlus = 22, 11, -11, 13, -13, 15, -15
    Pi0, PiPlus, PiMinus, K0_Long, KPlus, KMinus, K0_Short = 111, 211, -211, 130, 321, -321, 310
    PPlus, PMinus, Neutron = 2212, -2212, 2112
    Brems, DeltaE, PairProd, NuclInt, Hadrons = -2000001001, -2000001002, -2000001003, -2000001004, -2000001006


SHAPE_OTHER, SHAPE_CASCADE_SEGMENT = 0, 1


class I3CLSimLightSourceToStepConverterPPC:
    """Front end of the reference's converter (private/clsim/I3CLSimLightSourceToStepConverterPPC.cxx:51-132, 188-470): particles in,
    step requests out; GenerateSteps / GenerateStepsDevice make the steps on the GPU."""

    def __init__(self, photonsPerStep=200, highPhotonsPerStep=2000, useHighPhotonsPerStepStartingFromNumPhotons=1.0e9):
        if photonsPerStep <= 0 or highPhotonsPerStep <= 0:
            raise I3CLSimStepToPhotonConverter_exception("photonsPerStep may not be <= 0!")
        self._cfg = _lib.PPCConfig(int(photonsPerStep), int(highPhotonsPerStep), float(useHighPhotonsPerStepStartingFromNumPhotons), 1, 0, 0.9216, 0)
        self._bias = self._medium = self._h = None
        self._lib = _lib.load()

    def SetUseCascadeExtension(self, v):
        """may be called after Initialize(), as resources/tests/testCascadeExtension.py does (the library object is rebuilt)"""
        self._cfg.use_cascade_extension = int(bool(v))
        if self._h is not None:
            self._lib.clsimhip_ppc_destroy(self._h)
            self._h = None
            self.Initialize()

    def SetWlenBias(self, wlenBias):
        self._bias = wlenBias

    def SetMediumProperties(self, mediumProperties, density=0.9216):
        self._medium = mediumProperties
        self._cfg.medium_density = float(density)

    def SetRandomSeed(self, seed):
        self._cfg.seed = int(seed)

    def Initialize(self):
        if self._bias is None:
            raise I3CLSimStepToPhotonConverter_exception("WlenBias not set!")
        if self._medium is None:
            raise I3CLSimStepToPhotonConverter_exception("MediumProperties not set!")
        h = C.c_void_p()
        d = self._bias._desc()
        _check(self._lib.clsimhip_ppc_create(self._medium._h, C.byref(d), C.byref(self._cfg), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                self._lib.clsimhip_ppc_destroy(self._h)
        except Exception:
            pass

    def IsInitialized(self):
        return self._h is not None

    def MeanPhotonsPerMeter(self, layer=0):
        v = C.c_double()
        _check(self._lib.clsimhip_ppc_photons_per_meter(self._h, int(layer), C.byref(v)))
        return v.value

    def EnqueueLightSources(self, particles):
        """particles: array of PARTICLE_DTYPE -> array of REQUEST_DTYPE (one per cascade, two per muon / tau)"""
        if self._h is None:
            raise I3CLSimStepToPhotonConverter_exception("I3CLSimLightSourceToStepConverterPPC is not initialized!")
        p = np.ascontiguousarray(particles, dtype=PARTICLE_DTYPE)
        out = np.zeros(2 * len(p), dtype=REQUEST_DTYPE)
        n = C.c_size_t()
        _check(self._lib.clsimhip_ppc_enqueue(self._h, p.ctypes.data_as(C.c_void_p), len(p), out.ctypes.data_as(C.POINTER(_lib.StepRequest)),
                                              len(out), C.byref(n)))
        return out[:n.value]


def ShowerParameters(particleType, energy, density=0.9216):
    """(a, b [m], emScale, emScaleSigma) of I3SimConstants::ShowerParameters as restated in csrc/lightsource.cpp"""
    out = (C.c_double * 4)()
    _check(_lib.load().clsimhip_shower_parameters(int(particleType), float(energy), float(density), out))
    return tuple(out)


# ---- flasher step producer (I3CLSimLightSourceToStepConverterFlasher) ----
FLASHER_REQUEST_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("time", "<f4"), ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"),
                                  ("sigma_polar", "<f4"), ("sigma_azimuthal", "<f4"), ("pulse_width", "<f4"), ("identifier", "<u4"),
                                  ("source_type", "<u4"), ("num_photons_with_bias", "<u8")])
assert FLASHER_REQUEST_DTYPE.itemsize == 56
DIST_CONSTANT, DIST_NORMAL, DIST_UNIFORM, DIST_FLASHER_TIME_PROFILE = 0, 1, 2, 3


FLASHER_PULSE_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("time", "<f4"), ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"),
                                ("sigma_polar", "<f4"), ("sigma_azimuthal", "<f4"), ("pulse_width", "<f4"), ("identifier", "<u4"),
                                ("source_type", "<u4"), ("num_photons_no_bias", "<f8")])
assert FLASHER_PULSE_DTYPE.itemsize == 56


def FlasherPhotonNumberCorrectionFactor(wlenBias, spectrumNoBias=None, peakWavelength=None, fromWlen=265e-9, toWlen=675e-9):
    """PhotonNumberCorrectionFactorAfterBias (ConverterUtils.cxx:113-214): spectrumNoBias = I3CLSimFunctionFromTable, or None with
    peakWavelength for an I3CLSimFunctionDeltaPeak"""
    v = C.c_double()
    b = wlenBias._desc()
    if spectrumNoBias is None:
        _check(_lib.load().clsimhip_flasher_correction_factor(None, float(peakWavelength), C.byref(b), float(fromWlen), float(toWlen), C.byref(v)))
    else:
        sp = spectrumNoBias._desc()
        _check(_lib.load().clsimhip_flasher_correction_factor(C.byref(sp), 0.0, C.byref(b), float(fromWlen), float(toWlen), C.byref(v)))
    return v.value


def EnqueueFlasherPulses(pulses, correctionFactor, seed=0):
    """I3CLSimLightSourceToStepConverterFlasher::EnqueueLightSource (Flasher.cxx:214-265) for an array of FLASHER_PULSE_DTYPE:
    the converter's queue entries (FLASHER_REQUEST_DTYPE) with the photon numbers after bias drawn"""
    p = np.ascontiguousarray(pulses, dtype=FLASHER_PULSE_DTYPE)
    out = np.zeros(len(p), dtype=FLASHER_REQUEST_DTYPE)
    n = C.c_size_t()
    _check(_lib.load().clsimhip_flasher_enqueue(float(correctionFactor), int(seed), p.ctypes.data_as(C.c_void_p), len(p),
                                                out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
    return out[:n.value]


def FlasherStepConverterConfig(angularProfileDistributionPolar, angularProfileDistributionAzimuthal, timeDelayDistribution,
                               interpretAngularDistributionsInPolarCoordinates=False, photonsPerStep=400, maxBunchSize=512000,
                               bunchSizeGranularity=512):
    """Constructor arguments of I3CLSimLightSourceToStepConverterFlasher (Flasher.h; defaults Flasher.cxx:46-48); a
    distribution is (kind, value): (DIST_NORMAL, mean), (DIST_CONSTANT, 0), (DIST_UNIFORM, from), (DIST_FLASHER_TIME_PROFILE, 0).
    python/GetFlasherParameterizationList.py: LEDs = normal(0) / normal(0) / time profile, not polar; standard candles =
    constant / uniform(0) / normal(2 ns), polar."""
    c = _lib.FlasherConfig()
    for name, d in (("polar", angularProfileDistributionPolar), ("azimuthal", angularProfileDistributionAzimuthal), ("time_delay", timeDelayDistribution)):
        f = getattr(c, name)
        f.kind, f.value = int(d[0]), float(d[1])
    c.interpret_in_polar_coordinates = 1 if interpretAngularDistributionsInPolarCoordinates else 0
    c.photons_per_step, c.max_bunch_size, c.bunch_size_granularity = int(photonsPerStep), int(maxBunchSize), int(bunchSizeGranularity)
    return c


def CountFlasherSteps(config, requests):
    req = np.ascontiguousarray(requests, dtype=FLASHER_REQUEST_DTYPE)
    total, real = C.c_size_t(), C.c_size_t()
    _check(_lib.load().clsimhip_count_flasher_steps(C.byref(config), req.ctypes.data_as(C.c_void_p), len(req), C.byref(total), C.byref(real)))
    return total.value, real.value


def GenerateFlasherSteps(config, requests, seed, device=0):
    """All steps of the given flasher pulses (MakeSteps called until every pulse is used up), made on the GPU."""
    req = np.ascontiguousarray(requests, dtype=FLASHER_REQUEST_DTYPE)
    total, _ = CountFlasherSteps(config, req)
    out = np.zeros(total, dtype=STEP_DTYPE)
    n = C.c_size_t()
    _check(_lib.load().clsimhip_generate_flasher_steps(int(device), C.byref(config), req.ctypes.data_as(C.c_void_p), len(req), int(seed),
                                                       out.ctypes.data_as(C.c_void_p), total, C.byref(n)))
    return out[:n.value]


def GenerateFlasherStepsDevice(config, requests, seed, d_steps, capacity, device=0, stream=0):
    req = np.ascontiguousarray(requests, dtype=FLASHER_REQUEST_DTYPE)
    n = C.c_size_t()
    _check(_lib.load().clsimhip_generate_flasher_steps_device(int(device), C.byref(config), req.ctypes.data_as(C.c_void_p), len(req), int(seed),
                                                              C.c_void_p(int(d_steps)), int(capacity), C.c_void_p(int(stream)), C.byref(n)))
    return n.value


def FlasherTimeProfile(pulseWidthNs):
    """(density, cumulative) tables of the time delay distribution of one pulse width (240 points at 0.5 ns)."""
    d, c = np.zeros(240, dtype=np.float32), np.zeros(240, dtype=np.float32)
    _check(_lib.load().clsimhip_flasher_time_profile(float(pulseWidthNs), d.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)))
    return d, c
