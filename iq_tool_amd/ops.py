"""Operator-level entry points named after the reference functions they replace.  Each one runs
the same gfx950 kernels as the full chain with a single operator enabled (cf32 in, cf32 out), so
parity tests can be written per operator the way the reference's own API is cut:

    convert_block_to_cf32 / convert_cf32_to_block / get_bytes_per_sample   src/sample_convert.c
    DcBlock.apply / reset                                                  src/dc_block.c:68-86
    iq_correct_apply                                                       src/iq_correct.c:141-152
    FreqShift.apply / reset_nco                                            src/frequency_shift.c:86-107
    Resampler.execute / reset  (create_resampler, resampler_execute)       src/resampler.c:20-53
    Filter.apply / reset       (filter_create, filter_apply)               src/filter.c:138-526
    Agc.apply / reset          (agc_create, agc_apply, "digital" profile)  src/agc.c:21-238
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BYTES_PER_FRAME, FMT, check
from .chain import _NP_VIEW, Chain, _fmt


def get_bytes_per_sample(fmt):
    return _lib.load().iqgpu_get_bytes_per_sample(_fmt(fmt))


def convert_block_to_cf32(raw, input_format, gain=1.0, device=0):
    lib = _lib.load()
    fmt = _fmt(input_format)
    raw = np.ascontiguousarray(raw)
    n = raw.nbytes // BYTES_PER_FRAME[fmt]
    out = np.empty(n, np.complex64)
    check(lib.iqgpu_convert_block_to_cf32(raw.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                          n, fmt, gain, device))
    return out


def convert_cf32_to_block(x, output_format, device=0):
    lib = _lib.load()
    fmt = _fmt(output_format)
    x = np.ascontiguousarray(x, np.complex64)
    out = np.empty(x.size * BYTES_PER_FRAME[fmt], np.uint8)
    check(lib.iqgpu_convert_cf32_to_block(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                          x.size, fmt, device))
    return out.view(_NP_VIEW[fmt])


class _Cf32Op:
    def __init__(self, **kw):
        self.chain = Chain(in_format="cf32", out_format="cf32", **kw)

    def _run(self, x):
        y = self.chain.process(np.ascontiguousarray(x, np.complex64))
        return y.view(np.complex64)

    def reset(self):
        self.chain.reset()


class DcBlock(_Cf32Op):
    def __init__(self, input_rate_hz, **kw):
        super().__init__(input_rate_hz=input_rate_hz, no_resample=True, dc_block=True, **kw)

    apply = _Cf32Op._run


def iq_correct_apply(x, mag, phase, **kw):
    op = _Cf32Op(input_rate_hz=1.0, no_resample=True, iq_correct=True, iq_mag=mag, iq_phase=phase, **kw)
    return op._run(x)


class FreqShift(_Cf32Op):
    def __init__(self, shift_hz, rate_hz, **kw):
        super().__init__(input_rate_hz=rate_hz, no_resample=True, shift_hz=shift_hz, **kw)

    apply = _Cf32Op._run
    reset_nco = _Cf32Op.reset


class Resampler(_Cf32Op):
    """create_resampler(ratio) / resampler_execute / resampler_reset"""

    def __init__(self, resample_ratio, **kw):
        super().__init__(input_rate_hz=1.0, target_rate_hz=0.0, resample_ratio=float(np.float32(resample_ratio)), **kw)

    execute = _Cf32Op._run


class Filter(_Cf32Op):
    """filter_create / filter_apply at one rate (no resampler in the chain)."""

    def __init__(self, filters, rate_hz, **kw):
        super().__init__(input_rate_hz=rate_hz, no_resample=True, filters=tuple(filters), **kw)

    apply = _Cf32Op._run


class Agc(_Cf32Op):
    """agc_create / agc_apply / agc_reset with the "digital" profile.  One apply() call is cut into
    chunks of chunk_frames samples and agc_apply sees them one at a time, as the post-processor
    thread hands them over (src/post_processor.c:55-57)."""

    def __init__(self, sample_rate_hz, target=0.0, chunk_frames=16384, clock="samples", **kw):
        super().__init__(input_rate_hz=sample_rate_hz, no_resample=True, agc=True, agc_target=target,
                         agc_chunk_frames=chunk_frames, agc_clock=clock, **kw)

    apply = _Cf32Op._run

    @property
    def state(self):
        return self.chain.agc_state()
