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
