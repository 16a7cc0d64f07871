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
    # round 4: the objects of src/agc.c:39-62 and src/filter.c:339-351, and one half-band stage by itself
    L.agc_crcf_create.restype = vp; L.agc_crcf_create.argtypes = []
    L.agc_crcf_destroy.argtypes = [vp]
    L.agc_crcf_set_bandwidth.argtypes = [vp, f]
    L.agc_crcf_set_signal_level.argtypes = [vp, f]
    L.agc_crcf_set_gain.argtypes = [vp, f]
    L.agc_crcf_reset.argtypes = [vp]
    L.agc_crcf_execute_block.argtypes = [vp, vp, u, vp]
    L.agc_crcf_get_gain.restype = f; L.agc_crcf_get_gain.argtypes = [vp]
    L.firfilt_cccf_create.restype = vp; L.firfilt_cccf_create.argtypes = [vp, u]
    L.firfilt_cccf_destroy.argtypes = [vp]
    L.firfilt_cccf_execute_block.argtypes = [vp, vp, u, vp]
    L.fftfilt_cccf_create.restype = vp; L.fftfilt_cccf_create.argtypes = [vp, u, u]
    L.fftfilt_cccf_destroy.argtypes = [vp]
    L.fftfilt_cccf_execute.argtypes = [vp, vp, vp]
    L.fftfilt_crcf_create.restype = vp; L.fftfilt_crcf_create.argtypes = [vp, u, u]
    L.fftfilt_crcf_destroy.argtypes = [vp]
    L.fftfilt_crcf_execute.argtypes = [vp, vp, vp]
    L.resamp2_crcf_create.restype = vp; L.resamp2_crcf_create.argtypes = [u, f, f]
    L.resamp2_crcf_destroy.argtypes = [vp]
    L.resamp2_crcf_decim_execute.argtypes = [vp, vp, vp]
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


# ---- round 4: the operators the first set did not reach ---------------------------------------------

def test_dc_blocker_with_a_dc_offset_states_the_float_state_bar(liquid, oracle):
    """SPEC B.5, the documented divergence: liquid runs v0 = x - a1 v1 entirely in float, and with a DC offset present the
    state sits at |v| ~ DC / alpha (4e2 for 0.01 at 2.4 MS/s), whose float rounding leaves ~3e-5 of noise in y.  The
    canonical oracle (double state) and the GPU (affine scan, double carries) are within 1e-6 of the EXACT filter, so against
    a real liquid they are expected to differ by up to that noise: the bar stated here is 1e-4, NOT north_star's 1e-5, and
    the oracle's `literal` mode (the all-float recurrence, statement for statement) is the one that must match liquid
    to 2e-6."""
    x = synth.complex_signal(400000, 2.4e6, 75) + np.complex64(0.01 + 0.004j)
    alpha = np.float32(2 * np.pi * 10.0 / 2.4e6)
    q = liquid.iirfilt_crcf_create_dc_blocker(alpha)
    y = np.empty_like(x)
    liquid.iirfilt_crcf_execute_block(q, _p(x), x.size, _p(y))
    liquid.iirfilt_crcf_destroy(q)
    lit = oracle.DcBlock(alpha, literal=True).apply(x)
    assert np.abs(y - lit).max() <= 2e-6, "the all-float recurrence of SPEC B.5 is not what this liquid executes"
    canon = oracle.DcBlock(alpha).apply(x)
    d = float(np.abs(y - canon).max())
    assert d <= 1e-4, d            # liquid's own state noise (measured ~3e-5 in tests/test_oracle.py against lfilter)


@pytest.mark.parametrize("profile,bw", [("dx", 1e-4), ("local", 1e-2)])
def test_agc_crcf_rms_profiles(liquid, oracle, profile, bw):
    """agc_create (src/agc.c:39-62): agc_crcf_create, set_bandwidth(AGC_DX_BANDWIDTH / AGC_LOCAL_BANDWIDTH), set_signal_level(target),
    set_gain(1); agc_apply (agc.c:92-100): agc_crcf_execute_block in place; agc_reset (agc.c:227-229): reset + set_gain(1)"""
    x = (synth.complex_signal(150000, 744187.5, 76) * np.float32(0.3)).astype(np.complex64)
    x[60000:90000] *= np.float32(0.05)                      # a fade: the loop has to move
    q = liquid.agc_crcf_create()
    liquid.agc_crcf_set_bandwidth(q, np.float32(bw))
    liquid.agc_crcf_set_signal_level(q, np.float32(0.5 if profile == "local" else 0.9))
    liquid.agc_crcf_set_gain(q, np.float32(1.0))
    y = x.copy()
    liquid.agc_crcf_execute_block(q, _p(y), y.size, _p(y))
    g_liquid = float(liquid.agc_crcf_get_gain(q))
    a = oracle.Agc(744187.5, profile=profile)
    want = a.apply(x)
    assert np.abs(y - want).max() <= 2e-5 * max(1.0, float(np.abs(want).max()))
    assert abs(g_liquid - a.gain) <= 1e-5 * max(1.0, abs(a.gain))
    liquid.agc_crcf_reset(q); liquid.agc_crcf_set_gain(q, np.float32(1.0)); a.reset()
    y2 = x[:20000].copy()
    liquid.agc_crcf_execute_block(q, _p(y2), y2.size, _p(y2))
    liquid.agc_crcf_destroy(q)
    assert np.abs(y2 - a.apply(x[:20000])).max() <= 2e-5 * max(1.0, float(np.abs(y2).max()))


def _complex_taps(n, seed):
    rng = np.random.default_rng(seed)
    h = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.hanning(n) / n
    return h.astype(np.complex64)


def test_firfilt_cccf_is_the_plain_convolution(liquid):
    """filter.c:348 firfilt_cccf_create(master_taps, len) + filter_apply's execute_block (filter.c:452-460): SPEC B.2 --
    y[n] = sum h[k] x[n - k], taps NOT conjugated or reversed, zero history, unit scale"""
    h = _complex_taps(129, 3)
    x = synth.complex_signal(40000, 2.4e6, 77)
    q = liquid.firfilt_cccf_create(_p(h), h.size)
    y = np.empty_like(x)
    liquid.firfilt_cccf_execute_block(q, _p(x), x.size, _p(y))
    liquid.firfilt_cccf_destroy(q)
    want = np.convolve(x.astype(np.complex128), h.astype(np.complex128))[:x.size]
    assert np.abs(y - want).max() <= 1e-5


@pytest.mark.parametrize("kind", ["cccf", "crcf"])
def test_fftfilt_is_the_same_convolution_block_by_block(liquid, kind):
    """filter.c:339-342 fftfilt_cccf_create / fftfilt_crcf_create(taps, len, block) + execute per block (filter.c:513-515): SPEC B.3
    -- overlap-add linear convolution, no latency, unit scale: what k_fftconv16 computes as overlap-save, so only the
    block-quantised COUNT is observable"""
    n_taps, block = 257, 512
    x = synth.complex_signal(block * 40, 2.4e6, 78)
    if kind == "cccf":
        h = _complex_taps(n_taps, 4)
        q = liquid.fftfilt_cccf_create(_p(h), n_taps, block)
    else:
        h = (np.hanning(n_taps) / n_taps).astype(np.float32)
        q = liquid.fftfilt_crcf_create(_p(h), n_taps, block)
    y = np.empty_like(x)
    for b in range(0, x.size, block):
        xb, yb = np.ascontiguousarray(x[b:b + block]), np.empty(block, np.complex64)
        (liquid.fftfilt_cccf_execute if kind == "cccf" else liquid.fftfilt_crcf_execute)(q, _p(xb), _p(yb))
        y[b:b + block] = yb
    (liquid.fftfilt_cccf_destroy if kind == "cccf" else liquid.fftfilt_crcf_destroy)(q)
    want = np.convolve(x.astype(np.complex128), h.astype(np.complex128))[:x.size]
    assert np.abs(y - want).max() <= 2e-5


@pytest.mark.parametrize("m", [3, 5, 10])
def test_resamp2_stage_gain_and_delay(liquid, oracle, m):
    """one half-band decimator by itself, resamp2_crcf_create(m, 0, As) + decim_execute (what msresamp2 chains, SPEC B.6):
    the oracle's stage = 0.5 (delay branch + filter branch) on the same prototype -- DC gain 1 (not 2), centre tap on the
    ODD input sample of the pair delayed by m - 1 pairs.  VERIFY item: liquid <= 1.3 scaled this stage by 2."""
    x = synth.complex_signal(20000, 2.4e6, 79)
    q = liquid.resamp2_crcf_create(m, np.float32(0.0), np.float32(65.0))
    y = np.empty(x.size // 2, np.complex64)
    one = np.empty(1, np.complex64)
    for i in range(y.size):
        pair = np.ascontiguousarray(x[2 * i:2 * i + 2])
        liquid.resamp2_crcf_decim_execute(q, _p(pair), _p(one))
        y[i] = one[0]
    liquid.resamp2_crcf_destroy(q)
    # the oracle's own stage: a ratio just below 0.5 whose LAST stage has this m is not constructible for every m, so the
    # stage is restated here from the oracle's prototype rule (tests/np_design-style, SPEC B.6)
    t = np.arange(4 * m + 1) - 2 * m
    beta = 0.1102 * (65.0 - 8.7)
    proto = np.sinc(t / 2.0) * np.kaiser(4 * m + 1, beta)
    branch = proto[1::2]                                    # odd-indexed taps h[2k+1], k = 0 .. 2m-1 (symmetric)
    ev, od = x[0::2].astype(np.complex128), x[1::2].astype(np.complex128)
    filt = np.convolve(ev, branch)[:y.size]
    delay = np.concatenate([np.zeros(m - 1, np.complex128), od])[:y.size] if m > 1 else od[:y.size]
    want_a = 0.5 * (filt + delay)
    want_b = 0.5 * (np.convolve(od, branch)[:y.size] + np.concatenate([np.zeros(m, np.complex128), ev])[:y.size])
    err = min(float(np.abs(y - want_a).max()), float(np.abs(y - want_b).max()))
    assert err <= 1e-4, "half-band stage: neither branch assignment of SPEC B.6 matches this liquid (gain 1/2 per stage?)"
