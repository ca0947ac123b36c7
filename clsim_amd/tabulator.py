"""Python mirror of the photon table maker (clsim::tabulator, I3CLSimStepToTableConverter):
same names and argument meaning as the reference's pybindings
(private/pybindings/tabulator/*.cxx, python/tablemaker/tabulator.py:621-641), over the C ABI of
libclsimhip.so.  No CPU fallback: the table is filled by the HIP kernel only."""
import ctypes as C

import numpy as np

from . import _lib
from .converter import I3CLSimStepToPhotonConverter_exception
from .synthetic import STEP_DTYPE


class LinearAxis:
    """clsim::tabulator::LinearAxis(min, max, n_bins) (tabulator/Axis.h:71-80)."""
    kind = 0

    def __init__(self, min, max, n_bins):
        self.min, self.max, self.n_bins, self.power = float(min), float(max), int(n_bins), 1


class PowerAxis(LinearAxis):
    """clsim::tabulator::PowerAxis(min, max, n_bins, power) (tabulator/Axis.h:82-95)."""
    kind = 1

    def __init__(self, min, max, n_bins, power=1):
        LinearAxis.__init__(self, min, max, n_bins)
        self.power = int(power)


class _Axes:
    def __init__(self, axes):
        self.axes = list(axes)

    def __len__(self):
        return len(self.axes)


class SphericalAxes(_Axes):
    kind = 0


class CylindricalAxes(_Axes):
    kind = 1


class I3CLSimFunctionPolynomial:
    """I3CLSimFunctionPolynomial(coeffs[, rangemin, rangemax[, underflow, overflow]]) (Polynomial.cxx:35-85)."""

    def __init__(self, coefficients, rangemin=-np.inf, rangemax=np.inf, underflow=None, overflow=None):
        self.coefficients = np.ascontiguousarray(coefficients, dtype=np.float64)
        self.rangemin, self.rangemax = float(rangemin), float(rangemax)
        bounded = np.isfinite(self.rangemin) or np.isfinite(self.rangemax)
        self.underflow = float(underflow) if underflow is not None else (self.GetValue(self.rangemin) if bounded else np.nan)
        self.overflow = float(overflow) if overflow is not None else (self.GetValue(self.rangemax) if bounded else np.nan)

    def GetValue(self, x):                      # Polynomial.cxx:81-94
        if len(self.coefficients) == 0:
            return 0.0
        s, m = self.coefficients[0], 1.0
        for c in self.coefficients[1:]:
            m *= x
            s += c * m
        return float(s)


class I3CLSimStepToTableConverterHIP:
    """I3CLSimStepToTableConverter(device, axes, entriesPerStream, storeSquaredWeights, mediumProperties, spectrumTable,
    referenceArea, wavelengthAcceptance, angularAcceptance, rng) (tabulator/I3CLSimStepToTableConverter.h:47-52).
    entriesPerStream has no meaning here (samples go straight into the bins); streams=(x, a) stands in for rng."""

    def __init__(self, device, axes, storeSquaredWeights, mediumProperties, referenceArea, wavelengthAcceptance,
                 angularAcceptance, streams, stepLength=1.0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        ax = (_lib.Axis * len(axes))()
        for i, a in enumerate(axes.axes):
            ax[i].kind, ax[i].min, ax[i].max, ax[i].n_bins, ax[i].power = a.kind, a.min, a.max, a.n_bins, a.power
        poly = _lib.Polynomial()
        self._coeff = angularAcceptance.coefficients
        poly.n = len(self._coeff)
        poly.coefficients = self._coeff.ctypes.data_as(_lib.DP)
        poly.range_min, poly.range_max = angularAcceptance.rangemin, angularAcceptance.rangemax
        poly.underflow, poly.overflow = angularAcceptance.underflow, angularAcceptance.overflow
        x = np.ascontiguousarray(streams[0], dtype=np.uint64)
        a_ = np.ascontiguousarray(streams[1], dtype=np.uint32)
        f = wavelengthAcceptance._desc()
        self._acceptance = wavelengthAcceptance
        rc = self._lib.clsimhip_tabulator_create(int(device), axes.kind, ax, len(axes), 1 if storeSquaredWeights else 0,
                                                  mediumProperties._h, C.byref(f), C.byref(poly), float(referenceArea),
                                                  float(stepLength), x.ctypes.data_as(C.c_void_p), a_.ctypes.data_as(C.c_void_p),
                                                  len(x), C.byref(self._h))
        if rc != _lib.OK:
            raise I3CLSimStepToPhotonConverter_exception((self._lib.clsimhip_tabulator_last_error(None) or b"").decode(), rc)
        self._medium = mediumProperties
        n, nd, shape = C.c_size_t(), C.c_size_t(), (C.c_size_t * 5)()
        self._call("clsimhip_tabulator_get_shape", C.byref(n), C.byref(nd), shape)
        self.n_bins, self.shape = n.value, tuple(shape)[:nd.value]

    def __del__(self):
        try:
            if self._h:
                self._lib.clsimhip_tabulator_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _call(self, name, *args):
        rc = getattr(self._lib, name)(self._h, *args)
        if rc != _lib.OK:
            raise I3CLSimStepToPhotonConverter_exception((self._lib.clsimhip_tabulator_last_error(self._h) or b"").decode(), rc)

    def EnqueueSteps(self, steps, reference):
        """reference: (x, y, z, time, dx, dy, dz) of the source I3Particle."""
        steps = np.ascontiguousarray(steps, dtype=STEP_DTYPE)
        ref = (C.c_double * 7)(*[float(v) for v in reference])
        self._call("clsimhip_tabulator_enqueue_steps", steps.ctypes.data_as(C.c_void_p), len(steps), ref)

    def SetTuning(self, key, value):
        """clsimhip_tabulator_set_tuning: "fast_kernels" 0 | 1, "grid" (no result depends on either)"""
        self._call("clsimhip_tabulator_set_tuning", key.encode(), int(value))

    def Finish(self):
        self._call("clsimhip_tabulator_finish")

    def GetBinContent(self, squared=False, normalized=False):
        out = np.zeros(self.n_bins, dtype=np.float32)
        self._call("clsimhip_tabulator_get_bin_content", out.ctypes.data_as(C.c_void_p), self.n_bins, int(squared), int(normalized))
        return out.reshape(self.shape)

    def GetBinSums(self, squared=False):
        out = np.zeros(self.n_bins, dtype=np.float64)
        self._call("clsimhip_tabulator_get_bin_sums", out.ctypes.data_as(C.c_void_p), self.n_bins, int(squared))
        return out

    def GetBinEdges(self, axis):
        out = np.zeros(self.shape[axis] - 1, dtype=np.float64)
        self._call("clsimhip_tabulator_get_bin_edges", int(axis), out.ctypes.data_as(_lib.DP), len(out))
        return out

    def GetStatistics(self):
        out = (C.c_double * 8)()
        self._call("clsimhip_tabulator_get_statistics", out)
        keys = ["NumPhotons", "SumOfPhotonWeights", "n_group", "n_phase", "KernelTimeMs", "NumKernelCalls", "NumBins"]
        return dict(zip(keys, list(out)))

    def WriteFITSFile(self, path, tableHeader=None):
        """WriteFITSFile(path, tableHeader) (StepToTableConverter.cxx:595-686): ints and floats of the dict become
        "HIERARCH _i3_<key>" keywords (other value types are skipped, as in the reference)"""
        items = [(k, v) for k, v in (tableHeader or {}).items() if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool)]
        n = len(items)
        keys = (C.c_char_p * max(n, 1))(*[k.encode() for k, _ in items])
        is_int = (C.c_int32 * max(n, 1))(*[1 if isinstance(v, (int, np.integer)) else 0 for _, v in items])
        ints = (C.c_int64 * max(n, 1))(*[int(v) if isinstance(v, (int, np.integer)) else 0 for _, v in items])
        dbls = (C.c_double * max(n, 1))(*[float(v) for _, v in items])
        rc = self._lib.clsimhip_tabulator_write_fits_file(self._h, str(path).encode(), keys, is_int, ints, dbls, n)
        if rc != _lib.OK:
            raise I3CLSimStepToPhotonConverter_exception((self._lib.clsimhip_tabulator_last_error(self._h) or b"").decode(), rc)

    def GetRNGState(self, count):
        out = np.zeros(count, dtype=np.uint64)
        self._call("clsimhip_tabulator_get_rng_state", out.ctypes.data_as(C.c_void_p), int(count))
        return out

    def GetTable(self, name):
        n = self._lib.clsimhip_tabulator_get_table(self._h, name.encode(), None, 0)
        if n < 0:
            raise I3CLSimStepToPhotonConverter_exception((self._lib.clsimhip_tabulator_last_error(self._h) or b"").decode(), int(n))
        out = np.zeros(n, dtype=np.float64)
        self._lib.clsimhip_tabulator_get_table(self._h, name.encode(), out.ctypes.data_as(_lib.DP), n)
        return out
