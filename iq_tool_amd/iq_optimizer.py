"""Host-side mirror of the reference's I/Q imbalance optimiser (src/iq_correct.c:154-219, 315-393; thread
src/utility_threads.c:35-47) over the C ABI of include/iqgpu.h.  The optimiser itself is host code in the
reference and here; the GPU chain only supplies the probe block and consumes the factors."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import IqOptimizerStats, RAND_DIR_FN, check

FFT_SIZE = 1024            # IQ_CORRECTION_FFT_SIZE, include/constants.h:157


class IqOptimizer:
    def __init__(self, seed=None, rng=None):
        """seed: private deterministic direction source; rng: callable returning +1 / -1; neither: libc rand()
        as the reference (src/iq_correct.c:391-393)."""
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.iqgpu_iq_optimizer_create(C.byref(h)))
        self._h = h
        self._cb = None
        if rng is not None:
            self._cb = RAND_DIR_FN(lambda _u: float(rng()))
            check(self._lib.iqgpu_iq_optimizer_set_rng(self._h, self._cb, None))
        elif seed is not None:
            check(self._lib.iqgpu_iq_optimizer_seed(self._h, int(seed)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.iqgpu_iq_optimizer_destroy(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def _block(block):
        b = np.ascontiguousarray(block, np.complex64)
        if b.size != FFT_SIZE:
            raise ValueError("the optimiser works on blocks of %d samples" % FFT_SIZE)
        return b

    def set_factors(self, mag, phase):
        check(self._lib.iqgpu_iq_optimizer_set_factors(self._h, mag, phase))

    def factors(self):
        m, p = C.c_float(0), C.c_float(0)
        check(self._lib.iqgpu_iq_optimizer_get_factors(self._h, C.byref(m), C.byref(p)))
        return m.value, p.value

    def metric(self, block, mag, phase):
        """_calculate_imbalance_metric (src/iq_correct.c:339-360)"""
        b = self._block(block)
        return self._lib.iqgpu_iq_optimizer_metric(self._h, b.ctypes.data_as(C.c_void_p), mag, phase)

    def run_optimization(self, block, now_sec=-1.0):
        """iq_correct_run_optimization (src/iq_correct.c:154-219); True if the factors were updated"""
        b = self._block(block)
        upd = C.c_int(0)
        check(self._lib.iqgpu_iq_optimizer_run(self._h, b.ctypes.data_as(C.c_void_p), float(now_sec), C.byref(upd)))
        return bool(upd.value)

    def touch(self, now_sec=-1.0):
        check(self._lib.iqgpu_iq_optimizer_touch(self._h, float(now_sec)))

    def stats(self):
        st = IqOptimizerStats()
        check(self._lib.iqgpu_iq_optimizer_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in IqOptimizerStats._fields_}

    def service(self, chain, now_sec=-1.0):
        """the optimiser thread's loop body (src/utility_threads.c:35-47): probe -> run -> set_iq_factors"""
        upd = C.c_int(0)
        check(self._lib.iqgpu_iq_optimizer_service(self._h, chain._h, float(now_sec), C.byref(upd)))
        return bool(upd.value)
