"""Python mirror of the reference's step store and of its bunching rule, over the C ABI
(clsimhip_step_store_*; clsim_amd/csrc/step_store.cpp).

Reference: public/clsim/I3CLSimStepStore.h:44-320 (insert_copy, pop_bunch_to_vector, count) and the feeder thread of
I3CLSimLightSourceToStepConverterAsync (private/clsim/I3CLSimLightSourceToStepConverterAsync.cxx:209-273:
flushStepStore / emitStep): steps leave the store sorted by photon count in bunches of maxBunchSize; the last bunch
before a barrier is padded with no-op steps to the bunch granularity, and every bunch carries the identifiers of the
light sources whose steps have all left the store."""
import ctypes as C
from collections import deque

import numpy as np

from . import _lib
from .converter import I3CLSimStepToPhotonConverter_exception
from .synthetic import STEP_DTYPE


def no_op_step():
    """NoOpStepTemplate (Async.cxx:246-254): position 0, direction (0, 0, -1), no photons, weight 0, beta 1."""
    s = np.zeros(1, dtype=STEP_DTYPE)
    s["theta"] = np.float32(np.pi)          # I3CLSimStep::SetDir(I3Direction(0, 0, -1))
    s["beta"] = 1.0
    return s


class I3CLSimStepStore:
    def __init__(self, initialSize=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self._check(self._lib.clsimhip_step_store_create(int(initialSize), C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                self._lib.clsimhip_step_store_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _check(self, rc):
        if rc != _lib.OK:
            raise I3CLSimStepToPhotonConverter_exception((self._lib.clsimhip_last_error(None) or b"").decode() or ("status %d" % rc), rc)

    def insert_copy(self, steps):
        """insert_copy(step.GetNumPhotons(), step) for every record of `steps`."""
        steps = np.ascontiguousarray(np.atleast_1d(steps), dtype=STEP_DTYPE)
        self._check(self._lib.clsimhip_step_store_insert(self._h, steps.ctypes.data_as(C.c_void_p), len(steps)))

    def size(self):
        n = C.c_size_t()
        self._check(self._lib.clsimhip_step_store_size(self._h, C.byref(n)))
        return n.value

    def empty(self):
        return self.size() == 0

    def count(self, identifier):
        n = C.c_uint32()
        self._check(self._lib.clsimhip_step_store_count(self._h, int(identifier), C.byref(n)))
        return n.value

    def pop_bunch_to_vector(self, size, fill=None):
        out = np.zeros(int(size), dtype=STEP_DTYPE)
        if fill is None:
            n = C.c_size_t()
            self._check(self._lib.clsimhip_step_store_pop_bunch(self._h, int(size), out.ctypes.data_as(C.c_void_p), C.byref(n)))
            return out[:n.value]
        fill = np.ascontiguousarray(fill, dtype=STEP_DTYPE)
        self._check(self._lib.clsimhip_step_store_pop_bunch_filled(self._h, int(size), out.ctypes.data_as(C.c_void_p),
                                                                   fill.ctypes.data_as(C.c_void_p)))
        return out

    def size_with_dummy_fill(self, granularity):
        n = C.c_size_t()
        self._check(self._lib.clsimhip_step_store_size_with_dummy_fill(self._h, int(granularity), C.byref(n)))
        return n.value


class StepBuncher:
    """flushStepStore / emitStep of the reference's feeder thread (Async.cxx:209-273).  `emit` returns the full-sized
    bunches that became available, `flush` additionally the padded last bunch; a bunch is (steps, finished light
    source identifiers, is-last-before-barrier), the tuple the reference puts on queueFromGeant4_."""

    def __init__(self, maxBunchSize, bunchSizeGranularity=1):
        if maxBunchSize % bunchSizeGranularity != 0:        # Async.cxx SetMaxBunchSize / SetBunchSizeGranularity
            raise I3CLSimStepToPhotonConverter_exception("maxBunchSize is not a multiple of the bunch size granularity", _lib.ERR_ARGUMENT)
        self.maxBunchSize, self.granularity = int(maxBunchSize), int(bunchSizeGranularity)
        self.store = I3CLSimStepStore()
        self.markers = deque()

    def begin_light_source(self, identifier):
        self.markers.append(int(identifier))

    def _full_bunches(self):
        out = []
        while self.store.size() >= self.maxBunchSize:
            steps = self.store.pop_bunch_to_vector(self.maxBunchSize)
            finished = []
            while self.markers and self.store.count(self.markers[0]) == 0:
                finished.append(self.markers.popleft())
            out.append((steps, finished, False))
        return out

    def emit(self, steps):
        out = []
        for s in np.atleast_1d(steps):
            self.store.insert_copy(s)
            out.extend(self._full_bunches())
        return out

    def flush(self):
        out = self._full_bunches()
        steps = self.store.pop_bunch_to_vector(self.store.size_with_dummy_fill(self.granularity), fill=no_op_step())
        assert self.store.empty()
        finished = list(self.markers)
        self.markers.clear()
        out.append((steps, finished, True))
        return out


class I3CLSimLightSourceToStepConverterAsync:
    """The reference's asynchronous light-source converter (private/clsim/I3CLSimLightSourceToStepConverterAsync.cxx) over
    the C ABI's feeder (clsimhip_feeder_*, csrc/feeder.cpp): a worker thread turns light sources into steps (PPC front end
    + GPU step producer, or steps the caller supplies), runs them through the step store and hands out bunches."""

    def __init__(self, maxQueueItems=10):
        self._lib = _lib.load()
        self._depth = int(maxQueueItems)
        self._max_bunch, self._granularity, self._ppc, self._seed, self._device = 512000, 512, None, 0, 0
        self._h = None

    def SetBunchSizeGranularity(self, num):
        if num <= 0:
            raise I3CLSimStepToPhotonConverter_exception("BunchSizeGranularity of 0 is invalid!")
        self._granularity = int(num)

    def SetMaxBunchSize(self, num):
        if num <= 0:
            raise I3CLSimStepToPhotonConverter_exception("MaxBunchSize of 0 is invalid!")
        self._max_bunch = int(num)

    def SetLightSourceParameterization(self, ppc, seed=0, device=0):
        """the converter that parameterises particles (clsim_amd.converter.I3CLSimLightSourceToStepConverterPPC, initialised)"""
        self._ppc, self._seed, self._device = ppc, int(seed), int(device)

    def Initialize(self):
        h = C.c_void_p()
        ppc = self._ppc._h if self._ppc is not None else None
        rc = self._lib.clsimhip_feeder_create(ppc, self._device, self._seed, self._max_bunch, self._granularity, self._depth, C.byref(h))
        self._check(rc)
        self._h = h

    def __del__(self):
        try:
            if self._h:
                self._lib.clsimhip_feeder_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _check(self, rc):
        if rc != _lib.OK:
            raise I3CLSimStepToPhotonConverter_exception((self._lib.clsimhip_last_error(None) or b"").decode() or ("status %d" % rc), rc)

    def IsInitialized(self):
        return self._h is not None

    def EnqueueLightSource(self, particle):
        """particle: one record of clsim_amd.converter.PARTICLE_DTYPE (its identifier travels with it)"""
        p = np.ascontiguousarray(particle).reshape(1)
        self._check(self._lib.clsimhip_feeder_enqueue_light_source(self._h, p.ctypes.data_as(C.c_void_p)))

    def EnqueueSteps(self, identifier, steps):
        st = np.ascontiguousarray(steps, dtype=STEP_DTYPE)
        self._check(self._lib.clsimhip_feeder_enqueue_steps(self._h, int(identifier), st.ctypes.data_as(C.c_void_p), len(st)))

    def EnqueueBarrier(self):
        self._check(self._lib.clsimhip_feeder_enqueue_barrier(self._h))

    def BarrierActive(self):
        v = C.c_int()
        self._check(self._lib.clsimhip_feeder_barrier_active(self._h, C.byref(v)))
        return bool(v.value)

    def MoreStepsAvailable(self):
        v = C.c_int()
        self._check(self._lib.clsimhip_feeder_more_steps_available(self._h, C.byref(v)))
        return bool(v.value)

    def GetConversionResultWithBarrierInfoAndMarkers(self, timeout_ms=-1.0):
        """(steps, finished identifiers, barrierWasReset), or None on timeout"""
        got, n, nf, reset = C.c_int(), C.c_size_t(), C.c_size_t(), C.c_int()
        steps, fin = C.c_void_p(), C.c_void_p()
        self._check(self._lib.clsimhip_feeder_get_conversion_result(self._h, float(timeout_ms), C.byref(got), C.byref(steps), C.byref(n),
                                                                    C.byref(fin), C.byref(nf), C.byref(reset)))
        if not got.value:
            return None
        out = np.zeros(n.value, dtype=STEP_DTYPE)
        if n.value:
            C.memmove(out.ctypes.data, steps.value, n.value * STEP_DTYPE.itemsize)
        finished = list(np.ctypeslib.as_array(C.cast(fin, C.POINTER(C.c_uint32)), shape=(nf.value,)).copy()) if nf.value else []
        self._check(self._lib.clsimhip_feeder_release_result(self._h, steps))
        return out, [int(v) for v in finished], bool(reset.value)
