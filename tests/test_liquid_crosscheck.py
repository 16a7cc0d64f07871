"""Optional live cross-check of the oracle's adopted liquid-dsp semantics (DESIGN.md "SPEC") against a
real libliquid (last row).  liquid-dsp is not in this image and the reference does not pin
a version, so these tests SKIP unless a library is named:

    IQGPU_LIQUID_SO=/usr/lib/x86_64-linux-gnu/libliquid.so python -m pytest tests/test_liquid_crosscheck.py -q

Each test drives the same object the reference creates (file:line in the docstring) with the same
parameters and compares with the oracle on the same input; a failure here means the installed liquid
version implements a SPEC item differently from what this build adopted -- exactly the information a
maintainer needs to re-pin it.  Never a dependency of anything else.
"""
import ctypes as C
import os

import numpy as np
import pytest

from iq_tool_amd import synth

SO = os.environ.get("IQGPU_LIQUID_SO", "")
pytestmark = pytest.mark.skipif(not SO or not os.path.exists(SO), reason="liquid not found -- skipped (set IQGPU_LIQUID_SO)")


@pytest.fixture(scope="module")
def liquid():
    L = C.CDLL(SO)
    vp, u, f = C.c_void_p, C.c_uint, C.c_float
    L.nco_crcf_create.restype = vp; L.nco_crcf_create.argtypes = [C.c_int]
    L.nco_crcf_destroy.argtypes = [vp]
    L.nco_crcf_set_frequency.argtypes = [vp, f]
    L.nco_crcf_mix_block_up.argtypes = [vp, vp, vp, u]
    L.nco_crcf_mix_block_down.argtypes = [vp, vp, vp, u]
    L.msresamp_crcf_create.restype = vp; L.msresamp_crcf_create.argtypes = [f, f]
    L.msresamp_crcf_destroy.argtypes = [vp]
    L.msresamp_crcf_execute.argtypes = [vp, vp, u, vp, C.POINTER(u)]
    L.iirfilt_crcf_create_dc_blocker.restype = vp; L.iirfilt_crcf_create_dc_blocker.argtypes = [f]
    L.iirfilt_crcf_destroy.argtypes = [vp]
    L.iirfilt_crcf_execute_block.argtypes = [vp, vp, u, vp]
    L.liquid_firdes_kaiser.argtypes = [u, f, f, f, vp]
    L.estimate_req_filter_len.restype = u; L.estimate_req_filter_len.argtypes = [f, f]
    L.firfilt_crcf_create.restype = vp; L.firfilt_crcf_create.argtypes = [vp, u]
    L.firfilt_crcf_destroy.argtypes = [vp]
    L.firfilt_crcf_execute_block.argtypes = [vp, vp, u, vp]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("shift,rate", [(200e3, 2.4e6), (-333.3e3, 2.4e6)])
def test_nco_table_oscillator(liquid, oracle, shift, rate):
    """freq_shift_create / freq_shift_apply: nco_crcf_create(LIQUID_NCO), set_frequency, mix_block_up / _down
    (src/frequency_shift.c:24-107); SPEC B.4"""
    x = synth.complex_signal(100000, rate, 71)
    w = np.float32(2 * np.pi * abs(shift) / rate)
    q = liquid.nco_crcf_create(0)                                   # LIQUID_NCO = 0
    liquid.nco_crcf_set_frequency(q, w)
    y = np.empty_like(x)
    (liquid.nco_crcf_mix_block_up if shift >= 0 else liquid.nco_crcf_mix_block_down)(q, _p(x), _p(y), x.size)
    liquid.nco_crcf_destroy(q)
    want = oracle.Nco(w).mix(x, up=shift >= 0)
    assert np.abs(y - want).max() <= 2e-6


def test_dc_blocker(liquid, oracle):
    """dc_block_create / dc_block_apply: iirfilt_crcf_create_dc_blocker(alpha) (src/dc_block.c:32-66); SPEC B.5"""
    x = synth.complex_signal(200000, 2.4e6, 72) + np.complex64(0.05 - 0.02j)
    alpha = np.float32(2 * np.pi * 10.0 / 2.4e6)
    q = liquid.iirfilt_crcf_create_dc_blocker(alpha)
    y = np.empty_like(x)
    liquid.iirfilt_crcf_execute_block(q, _p(x), x.size, _p(y))
    liquid.iirfilt_crcf_destroy(q)
    want = oracle.DcBlock(alpha).apply(x)
    assert np.abs(y - want).max() <= 1e-5


@pytest.mark.parametrize("r", [744187.5 / 2.4e6, 0.24, 0.02422485314, 0.8, 2.5])
def test_msresamp(liquid, oracle, r):
    """create_resampler / resampler_execute: msresamp_crcf_create(r, 60 dB) (src/resampler.c:20-53); SPEC B.6:
    stage count and semi-lengths, 256-arm polyphase with 24-bit phase, per-call output counts"""
    r = np.float32(r)
    x = synth.complex_signal(120000, 2.4e6, 73)
    q = liquid.msresamp_crcf_create(r, np.float32(60.0))
    m = oracle.MsResamp(r)
    pos, got, want = 0, [], []
    for n in (16384, 16384, 1, 40000, 120000 - 72769):
        y = np.empty(int(np.ceil(n * max(1.0, float(r)))) + 4096, np.complex64)
        ny = C.c_uint(0)
        blk = np.ascontiguousarray(x[pos:pos + n])
        liquid.msresamp_crcf_execute(q, _p(blk), n, _p(y), C.byref(ny))
        w = m.execute(blk)
        assert ny.value == w.size, "per-call output count differs (group buffering / phase quantisation)"
        got.append(y[:ny.value].copy()); want.append(w)
        pos += n
    liquid.msresamp_crcf_destroy(q)
    assert np.abs(np.concatenate(got) - np.concatenate(want)).max() <= 1e-5


def test_kaiser_design_and_fir(liquid, oracle):
    """filter_create: estimate_req_filter_len, liquid_firdes_kaiser; filter_apply: firfilt_crcf
    (src/filter.c:180-218, 449-462); SPEC B.1-B.2"""
    n = liquid.estimate_req_filter_len(np.float32(0.02), np.float32(60.0))
    assert n == oracle.lib().orc_estimate_req_filter_len(np.float32(0.02), np.float32(60.0))
    if n % 2 == 0:
        n += 1
    h = np.empty(n, np.float32)
    liquid.liquid_firdes_kaiser(n, np.float32(0.1), np.float32(60.0), np.float32(0.0), _p(h))
    assert np.abs(h - oracle.firdes_kaiser(n, 0.1, 60.0)).max() <= 1e-6
    x = synth.complex_signal(50000, 2.4e6, 74)
    q = liquid.firfilt_crcf_create(_p(h), n)
    y = np.empty_like(x)
    liquid.firfilt_crcf_execute_block(q, _p(x), x.size, _p(y))
    liquid.firfilt_crcf_destroy(q)
    want = np.convolve(x.astype(np.complex128), h.astype(np.float64))[:x.size]
    assert np.abs(y - want).max() <= 1e-5
