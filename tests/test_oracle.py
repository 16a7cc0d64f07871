"""CPU tests of the parity oracle (oracle/iq_oracle.c).

Pinned part: sample_convert against golden vectors produced by the reference's own
src/sample_convert.c (tests/golden/sample_convert.npz, made by tests/golden/gen_golden.py) and, when
oracle/_ref is present, against that build live.

Unpinned part (liquid-dsp is absent from the reference tree and this image): every liquid-derived
operator is cross-checked against an independent numpy / scipy formulation of the same published
algorithm, so that an indexing, gain or phase mistake in the C restatement cannot hide.
"""
import math
import os

import numpy as np
import pytest
import scipy.signal as sps

from iq_tool_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FORMATS = ["cs8", "cu8", "cs16", "cu16", "sc16q11", "cs24", "cs32", "cu32", "cf32"]
NRSC5 = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)


# --------------------------------------------------------------------------------------------
# a3 / a15 / a16 -- pinned to the reference C
# --------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gold_convert():
    return np.load(os.path.join(GOLD, "sample_convert.npz"))


@pytest.mark.parametrize("fmt", FORMATS)
def test_unpack_matches_reference_golden(oracle, gold_convert, fmt):
    raw = gold_convert["unpack_in_" + fmt]
    for g, tag in ((1.0, "g1"), (0.37, "g037"), (-2.5, "gm25")):
        want = gold_convert["unpack_out_%s_%s" % (fmt, tag)]
        got = oracle.to_cf32(raw, fmt, g).view(np.float32)
        assert np.array_equal(got, want), (fmt, g)


@pytest.mark.parametrize("fmt", FORMATS)
def test_pack_matches_reference_golden(oracle, gold_convert, fmt):
    x = gold_convert["pack_in_" + fmt].view(np.complex64)
    assert np.array_equal(oracle.from_cf32(x, fmt), gold_convert["pack_out_" + fmt])


def test_bytes_per_sample_matches_reference_golden(oracle, gold_convert):
    for fid, nbytes in gold_convert["bytes_per_sample"]:
        assert oracle.lib().orc_bytes_per_sample(int(fid)) == int(nbytes)


@pytest.mark.parametrize("fmt", FORMATS)
def test_convert_matches_reference_build_live(oracle, fmt):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(5)
    n = 200003
    raw = rng.integers(0, 256, n * oracle.BYTES[oracle.FMT[fmt]], dtype=np.uint8)
    if fmt == "cf32":
        raw = rng.standard_normal(2 * n).astype(np.float32).view(np.uint8)
    for g in (1.0, 0.123, -7.0):
        assert np.array_equal(oracle.to_cf32(raw, fmt, g).view(np.float32), oracle.ref_to_cf32(raw, fmt, g).view(np.float32))
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7).astype(np.complex64)
    assert np.array_equal(oracle.from_cf32(x, fmt), oracle.ref_from_cf32(x, fmt))


def test_synth_quantiser_is_the_reference_rule(oracle):
    x = synth.complex_signal(100000, 2.4e6, 3)
    for fmt in ("cs16", "cu8", "cs8", "cu16", "sc16q11"):
        assert np.array_equal(synth.quantise(x, fmt), oracle.from_cf32(x, fmt))


# --------------------------------------------------------------------------------------------
# NCO (SPEC B.4)
# --------------------------------------------------------------------------------------------
def test_nco_constants_and_table(oracle):
    w = np.float32(2 * np.pi * 200e3 / 2.4e6)
    nco = oracle.Nco(w)
    # (float)(w * 0.159154943091895) * 2^32, truncated
    p = np.float32(float(w) * 0.159154943091895)
    assert nco.dtheta_u32 == int(float(p) * 4294967296.0)
    assert abs(nco.dtheta_u32 - 2 ** 32 / 12) < 64
    tab = nco.table()
    ref = np.sin((2.0 * np.pi * np.arange(1024, dtype=np.float64) / 1024.0).astype(np.float32).astype(np.float64))
    assert np.abs(tab - ref).max() < 1e-7
    assert oracle.lib().orc_nco_constrain(np.float32(-0.5)) > 2 ** 31        # negative phase wraps


def test_nco_mix_is_table_lookup_exactly(oracle):
    n = 50000
    x = synth.complex_signal(n, 2.4e6, 1)
    for shift, up in ((200e3, True), (-333.3e3, False), (1234.5, True)):
        w = np.float32(2 * np.pi * abs(shift) / 2.4e6)
        nco = oracle.Nco(w)
        y = nco.mix(x, up=up)
        theta = (np.arange(n, dtype=np.uint64) * np.uint64(nco.dtheta_u32)) & np.uint64(0xffffffff)
        idx = ((theta + np.uint64(1 << 21)) >> np.uint64(22)) & np.uint64(1023)
        tab = nco.table().astype(np.float64)
        s = tab[idx.astype(np.int64)]
        c = tab[((idx + np.uint64(256)) & np.uint64(1023)).astype(np.int64)]
        v = c + 1j * (s if up else -s)
        want = (x.astype(np.complex128) * v).astype(np.complex64)
        assert np.abs(y - want).max() <= 1.2e-7
        # and within the table's phase quantisation of the exact oscillator
        exact = x.astype(np.complex128) * np.exp((1j if up else -1j) * 2 * np.pi * theta.astype(np.float64) / 2 ** 32)
        assert np.abs(y - exact).max() <= np.abs(x).max() * (np.pi / 1024 + 1e-6)
        assert nco.theta_u32 == int((n * nco.dtheta_u32) & 0xffffffff)


# --------------------------------------------------------------------------------------------
# DC blocker (SPEC B.5) and iq correction
# --------------------------------------------------------------------------------------------
def test_dc_block_matches_lfilter(oracle):
    n = 300000
    x = synth.complex_signal(n, 2.4e6, 2)
    alpha = np.float32(2 * np.pi * 10.0 / 2.4e6)
    c = -float(np.float32(-1.0) + alpha)                  # a1 = -1 + alpha in float
    want = sps.lfilter([1.0, -1.0], [1.0, -c], x.astype(np.complex128))
    got = oracle.DcBlock(alpha).apply(x)
    assert np.abs(got - want).max() <= 2e-7
    # chunking does not matter
    d = oracle.DcBlock(alpha)
    parts = np.concatenate([d.apply(x[:7]), d.apply(x[7:100000]), d.apply(x[100000:])])
    assert np.array_equal(parts, got)
    # the all-float recurrence (what liquid executes) wanders around the exact filter by the
    # rounding of its state (|v| ~ DC / alpha): this is the parity budget liquid itself needs
    lit = oracle.DcBlock(alpha, literal=True).apply(x)
    dev = np.abs(lit - got).max()
    assert 1e-7 < dev < 2e-3
    d.reset()
    assert np.array_equal(d.apply(x[:1000]), got[:1000])


def test_iq_correct(oracle):
    x = synth.complex_signal(10000, 2.4e6, 4)
    y = oracle.iq_correct(x, 0.01, -0.005)
    magp1 = np.float32(1.0) + np.float32(0.01)
    assert np.array_equal(y.real, x.real * magp1)
    assert np.array_equal(y.imag, x.imag + np.float32(-0.005) * x.real)


# --------------------------------------------------------------------------------------------
# Kaiser design primitives (SPEC B.1)
# --------------------------------------------------------------------------------------------
def test_kaiser_primitives(oracle):
    L = oracle.lib()
    assert abs(L.orc_kaiser_beta_As(60.0) - 0.1102 * (60 - 8.7)) < 1e-6
    assert abs(L.orc_kaiser_beta_As(40.0) - (0.5842 * 19 ** 0.4 + 0.07886 * 19)) < 1e-6
    assert L.orc_kaiser_beta_As(10.0) == 0.0
    for z in (0.0, 0.5, 3.0, 7.3):
        assert abs(L.orc_besseli0(z) - np.i0(z)) < 1e-12 * np.i0(z)
    assert L.orc_estimate_req_filter_len(0.1, 65.0) == 40
    for n, fc, As in ((41, 0.2, 60.0), (3585, 0.319 / 256, 60.0), (1025, 0.0235, 60.0)):
        h = oracle.firdes_kaiser(n, fc, As)
        t = np.arange(n) - (n - 1) / 2
        beta = float(L.orc_kaiser_beta_As(As))
        x = 2 * np.float32(fc).astype(np.float64) * t
        sinc = np.where(np.abs(x) < 0.01, np.cos(np.pi * x / 2) * np.cos(np.pi * x / 4) * np.cos(np.pi * x / 8), np.sinc(x))
        want = sinc * np.kaiser(n, beta)
        assert np.abs(h - want).max() < 2e-7


# --------------------------------------------------------------------------------------------
# msresamp (SPEC B.6)
# --------------------------------------------------------------------------------------------
def _numpy_msresamp_decim(oracle, m, x):
    """independent formulation: FIR + stride for the half-bands, explicit gather for the polyphase"""
    s = x.astype(np.complex128)
    for k in range(m.S):
        h = m.stage_taps(k).astype(np.float64).copy()
        mm = m.stage_m(k)
        h[0::2] = 0.0
        h[2 * mm] = 1.0                                   # centre tap is exactly the delay branch
        y = sps.lfilter(h, [1.0], s)
        s = 0.5 * y[1::2]
    Q = s.size
    step = m.step
    K = -(-(Q << 24) // step)
    k = np.arange(K, dtype=np.uint64)
    P = k * np.uint64(step)
    q = (P >> np.uint64(24)).astype(np.int64)
    arm = ((P >> np.uint64(16)) & np.uint64(255)).astype(np.int64)
    proto = m.arb_proto().astype(np.float64)
    out = np.zeros(K, np.complex128)
    sp = np.concatenate([np.zeros(13, np.complex128), s])
    for n in range(14):
        out += proto[arm + 256 * n] * sp[q + 13 - n]
    return out


@pytest.mark.parametrize("rates", [(2.4e6, 744187.5), (10e6, 2.4e6), (61.44e6, 1488375.0), (2.4e6, 2.0e6), (2.4e6, 1.2e6)])
def test_msresamp_matches_independent_numpy(oracle, rates):
    r = np.float32(rates[1] / rates[0])
    m = oracle.MsResamp(r)
    n = (1 << 17) + 37
    x = synth.complex_signal(n, rates[0], 6)
    y = m.execute(x)
    want = _numpy_msresamp_decim(oracle, m, x[:(n >> m.S) << m.S])
    assert y.size == want.size == math.ceil(((n >> m.S) << 24) / m.step)
    assert np.abs(y - want).max() <= 3e-7


def test_msresamp_structure_and_counts(oracle):
    r = np.float32(744187.5 / 2.4e6)
    m = oracle.MsResamp(r)
    assert (m.S, m.stage_m(0), m.interp) == (1, 10, False)
    assert m.rate_arb == np.float32(2) * r
    assert m.step == int(round(float(np.float32(16777216.0) / m.rate_arb)))
    assert [oracle.MsResamp(np.float32(x)).S for x in (0.9, 0.5, 0.49, 0.24, 0.024, 0.001)] == [0, 0, 1, 2, 5, 9]
    m5 = oracle.MsResamp(np.float32(1488375.0 / 61.44e6))
    assert [m5.stage_m(k) for k in range(5)] == [3, 3, 3, 5, 10]
    # chunk invariance and reset
    x = synth.complex_signal(50001, 2.4e6, 7)
    whole = m.execute(x)
    m.reset()
    parts = np.concatenate([m.execute(x[:1]), m.execute(x[1:16385]), m.execute(x[16385:])])
    assert np.array_equal(whole, parts)
    # per-call counts follow the closed form
    m.reset()
    tot_in = tot_out = 0
    for n in (1, 2, 3, 1000, 16384, 7):
        got = m.execute(x[tot_in:tot_in + n]).size
        tot_in += n
        want_total = math.ceil(((tot_in >> m.S) << 24) / m.step)
        assert tot_out + got == want_total
        tot_out += got


def test_msresamp_tone_gain_and_image_rejection(oracle):
    r = np.float32(744187.5 / 2.4e6)
    m = oracle.MsResamp(r)
    n = 400000
    t = np.arange(n)
    r_eff = 2.0 ** 24 / (m.step * 2)
    for f_in, expect_pass in ((0.0, True), (0.03, True), (0.08, True), (0.30, False), (0.45, False)):
        m.reset()
        y = m.execute(np.exp(2j * np.pi * f_in * t).astype(np.complex64))[4000:]
        amp = np.sqrt(np.mean(np.abs(y) ** 2))
        if expect_pass:
            assert abs(amp - 1.0) < 5e-3
            f_out = np.angle(np.mean(y[1:] * np.conj(y[:-1]))) / (2 * np.pi)
            assert abs(f_out - f_in / r_eff) < 1e-6
        else:
            assert 20 * np.log10(amp + 1e-30) < -55.0


def test_msresamp_interpolation_oracle_only(oracle):
    """r > 1 exists in the oracle (arbitrary stage first, then half-band interpolators)"""
    m = oracle.MsResamp(np.float32(3.3))
    assert m.interp and m.S == 1
    n = 20000
    y = m.execute(np.exp(2j * np.pi * 0.01 * np.arange(n)).astype(np.complex64))
    assert abs(y.size / n - 3.3) < 0.01
    assert abs(np.sqrt(np.mean(np.abs(y[2000:]) ** 2)) - 1.0) < 5e-3


# --------------------------------------------------------------------------------------------
# user filter (src/filter.c)
# --------------------------------------------------------------------------------------------
def test_filter_design_placement_and_kind(oracle):
    f = oracle.Filter(oracle.make_filter_cfg((("lowpass", 300e3, 0.0),), filter_taps=4097, impl="fir"), 61.44e6, 1488375.0)
    assert (f.post, f.impl, f.ntaps, f.block) == (True, 1, 4097, 0)
    taps = f.taps()
    assert np.all(taps.imag == 0) and abs(taps.real.sum() - 1.0) < 1e-6
    f = oracle.Filter(oracle.make_filter_cfg((("passband", 158.5e3, 113e3),), filter_taps=1025), 10e6, 2.4e6)
    assert (f.post, f.impl, f.ntaps, f.block) == (True, 4, 1025, 2048)
    H = np.fft.fft(f.taps(), 1 << 16)
    assert abs(np.abs(H).max() - 1.0) < 1e-3
    fpk = np.fft.fftfreq(1 << 16)[np.argmax(np.abs(H))] * 2.4e6
    assert 102e3 < fpk < 215e3
    # auto length: estimate_req_filter_len, bumped to odd, >= 21
    f = oracle.Filter(oracle.make_filter_cfg((("lowpass", 100e3, 0.0),)), 2.4e6, 2.4e6, no_resample=True)
    want = int(np.float32(60 - 7.95) / (np.float32(14.26) * (np.float32(25e3) / np.float32(2.4e6))))
    want += 1 - want % 2
    assert f.ntaps == want and not f.post and f.impl == 1
    with pytest.raises(ValueError):
        oracle.Filter(oracle.make_filter_cfg((("lowpass", 600e3, 0.0),)), 2.4e6, 744187.5)       # beyond output Nyquist
    with pytest.raises(ValueError):
        oracle.Filter(oracle.make_filter_cfg((("passband", 50e3, 20e3),), filter_taps=1025, fft_size=1024), 2.4e6, 2.4e6, no_resample=True)


def test_filter_apply_is_linear_convolution(oracle):
    x = synth.complex_signal(30000, 2.4e6, 8)
    for reqs, kw in (((("lowpass", 200e3, 0.0),), {}), ((("passband", -300e3, 100e3),), dict(filter_taps=257, impl="fir")),
                     ((("highpass", 100e3, 0.0), ("stopband", 400e3, 50e3)), {})):
        f = oracle.Filter(oracle.make_filter_cfg(reqs, **kw), 2.4e6, 2.4e6, no_resample=True)
        want = sps.lfilter(f.taps().astype(np.complex128), [1.0], x.astype(np.complex128))
        got = np.concatenate([f.apply(x[:12345]), f.apply(x[12345:])])
        assert np.abs(got - want).max() <= 3e-7


def test_fft_filter_counts_and_reset_quirk(oracle):
    x = synth.complex_signal(20000, 2.4e6, 9)
    cfg = oracle.make_filter_cfg((("passband", 300e3, 100e3),), filter_taps=129)
    f = oracle.Filter(cfg, 2.4e6, 2.4e6, no_resample=True)
    assert f.impl == 4 and f.block == 256            # 128 >= L-1, doubled because < 2L (src/filter.c:327-333)
    B = f.block
    full = sps.lfilter(f.taps().astype(np.complex128), [1.0], x.astype(np.complex128))
    pos = emitted = 0
    for n in (100, 500, 1, 1000, 7000, 11399):
        y = f.apply(x[pos:pos + n])
        pos += n
        assert y.size == (pos // B) * B - emitted
        if y.size:
            assert np.abs(y - full[emitted:emitted + y.size]).max() <= 3e-7
        emitted += y.size
    # reset clears the filter history but NOT the remainder (src/filter.c:417-436)
    rem = pos - emitted
    f.reset()
    y = f.apply(x[:B - rem])
    assert y.size == B
    stream = np.concatenate([x[emitted:pos], x[:B - rem]])
    want = sps.lfilter(f.taps().astype(np.complex128), [1.0], stream.astype(np.complex128))
    assert np.abs(y - want).max() <= 3e-7


# --------------------------------------------------------------------------------------------
# chain
# --------------------------------------------------------------------------------------------
def test_chain_is_the_composition_of_its_operators(oracle):
    n = 100000
    raw = synth.raw_stream(n, 10e6, 3, "cs16")
    kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=10e6, target_rate_hz=2.4e6, gain=0.8, shift_hz=-250e3,
              dc_block=True, iq_correct=True, iq_mag=0.01, iq_phase=-0.005, filters=(("lowpass", 500e3, 0.0),))
    out_i, out_c = oracle.Chain(**kw).process(raw, want_cf32=True)
    x = oracle.to_cf32(raw, "cs16", 0.8)
    x = oracle.DcBlock(np.float32(2 * np.pi * 10.0 / 10e6)).apply(x)
    x = oracle.iq_correct(x, 0.01, -0.005)
    x = oracle.Nco(np.float32(2 * np.pi * 250e3 / 10e6)).mix(x, up=False)
    y = oracle.MsResamp(np.float32(2.4e6 / 10e6)).execute(x)
    f = oracle.Filter(oracle.make_filter_cfg((("lowpass", 500e3, 0.0),)), 10e6, 2.4e6)
    assert f.post
    z = f.apply(y)
    assert np.array_equal(z, out_c)
    assert np.array_equal(oracle.from_cf32(z, "cs16"), out_i)


def test_chain_chunk_invariance_and_reset(oracle):
    n = 70000
    raw = synth.raw_stream(n, 2.4e6, 1, "cs16")
    c = oracle.Chain(**NRSC5)
    whole = c.process(raw)
    c.reset()
    parts = np.concatenate([c.process(raw[:2 * 5]), c.process(raw[2 * 5:2 * 20001]), c.process(raw[2 * 20001:])])
    assert np.array_equal(whole, parts)
    assert abs(c.ratio - np.float32(744187.5 / 2.4e6)) == 0
    assert whole.size // 2 <= c.max_out_frames(n)


def test_chain_golden_regression(oracle):
    g = np.load(os.path.join(GOLD, "nrsc5_65536.npz"))
    out_i, out_c = oracle.Chain(**NRSC5).process(g["raw"], want_cf32=True)
    assert np.array_equal(out_i, g["out_cs16"])
    assert np.array_equal(out_c.view(np.float32), g["out_cf32"])
    assert np.array_equal(g["raw"], synth.raw_stream(65536, 2.4e6, 1, "cs16"))


def test_design_golden_regression(oracle):
    g = np.load(os.path.join(GOLD, "design.npz"))
    m = oracle.MsResamp(g["ratio"])
    assert m.step == int(g["step"]) and m.S == int(g["S"])
    assert np.array_equal(m.stage_taps(0), g["hb_taps0"]) and np.array_equal(m.arb_proto(), g["arb_proto"])
    nco = oracle.Nco(np.float32(2 * np.pi * 200e3 / 2.4e6))
    assert nco.dtheta_u32 == int(g["nco_dtheta"]) and np.array_equal(nco.table(), g["nco_table"])


def test_chain_create_errors(oracle):
    for kw in (dict(NRSC5, target_rate_hz=100.0), dict(NRSC5, in_format=3), dict(NRSC5, shift_hz=0.0, shift_after_resample=True),
               dict(NRSC5, shift_hz=2.4e6 * 6), dict(NRSC5, filters=(("lowpass", 600e3, 0.0),))):
        with pytest.raises(ValueError):
            oracle.Chain(**kw)


# --------------------------------------------------------------------------------------------
# output AGC, "digital" profile (src/agc.c) against an independent numpy restatement
# --------------------------------------------------------------------------------------------
def _agc_numpy(x, rate, target=0.9, chunk=16384):
    """agc.c:105-222 with the sample-count clock, written chunk by chunk in float32 numpy"""
    f = np.float32
    locked, peak_mem, g, seen, last_strong = False, f(0.05), f(1.0), 0, 0.0
    y = np.empty_like(x)
    gains = []
    for b in range(0, len(x), chunk):
        blk = x[b:b + chunk]
        peak = f(np.hypot(blk.real.astype(np.float64), blk.imag.astype(np.float64)).max())
        now = seen / rate
        if not locked:
            if peak > peak_mem:
                peak_mem = peak
            safe = f(1e-4) if peak_mem < f(1e-4) else peak_mem
            cur = f(target) / safe
            if seen / rate > 2.0:
                locked, g, last_strong = True, cur, now
        else:
            outp = f(peak * g)
            if outp > f(1.0):
                g = f(0.99) / peak
                last_strong = now
            elif outp > f(f(target) * f(0.75)):
                last_strong = now
            elif now - last_strong > 4.0:
                g = f(g * f(1.0005))
            cur = g
        y[b:b + chunk] = (blk.real * cur + 1j * (blk.imag * cur)).astype(np.complex64)
        gains.append(float(cur))
        seen += len(blk)
    return y, np.array(gains), locked, float(g), float(peak_mem)


def test_agc_digital_matches_numpy_restatement(oracle):
    rate, chunk = 8000.0, 1000
    x = synth.agc_envelope_signal(120000, rate, 31)
    want, gains, locked, g, pm = _agc_numpy(x, rate, chunk=chunk)
    a = oracle.Agc(rate)
    got = a.apply_chunked(x, chunk)
    assert np.array_equal(got.view(np.float32), want.view(np.float32))
    assert a.locked and locked and a.gain == np.float32(g) and a.peak_memory == np.float32(pm)
    assert a.samples_seen == x.size
    # the envelope really drove every branch
    assert np.any(np.diff(gains[:20]) < 0) and np.any(np.diff(gains[20:]) < 0) and np.any(np.diff(gains[40:]) > 0)
    a.reset()
    assert not a.locked and a.gain == 1.0 and a.peak_memory == np.float32(0.05) and a.samples_seen == 0


def _agc_crcf_python(x, alpha):
    """liquid agc_crcf_execute [liquid-mem], one numpy float32 operation per C operation (math.exp / math.log in double
    rounded to float stand in for glibc's correctly rounded expf / logf)"""
    import math
    f = np.float32
    g, p, a = f(1.0), f(1.0), f(alpha)
    out = np.empty(x.size, np.complex64)
    for i, v in enumerate(x):
        yr, yi = f(v.real) * g, f(v.imag) * g
        y2 = f(yr * yr) + f(yi * yi)
        p = f((1.0 - float(a)) * float(p) + float(a) * float(y2))
        if p > f(1e-6):
            g = g * f(math.exp(float(f(f(-0.5) * a) * f(math.log(float(p))))))
        if g > f(1e6):
            g = f(1e6)
        out[i] = complex(yr, yi)
    return out, g, p


@pytest.mark.parametrize("profile,alpha", [("local", 1e-2), ("dx", 1e-4)])
def test_agc_rms_profiles_match_python_restatement(oracle, profile, alpha):
    rng = np.random.default_rng(33)
    x = np.concatenate([0.05 * (rng.standard_normal(6000) + 1j * rng.standard_normal(6000)), np.zeros(3000),
                        0.6 * (rng.standard_normal(4000) + 1j * rng.standard_normal(4000))]).astype(np.complex64)
    want, g, p = _agc_crcf_python(x, alpha)
    a = oracle.Agc(1e6, profile=profile)
    got = np.concatenate([a.apply(x[:777]), a.apply(x[777:])])        # a per-sample loop: the call partition is invisible
    # glibc's expf / logf are within 0.51 ulp, not always the nearest float: a step that rounds the other way leaves the
    # two runs one ulp apart for ~1 / alpha samples
    same = got.view(np.float32) == want.view(np.float32)
    assert same.mean() > 0.9 and np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    assert abs(a.gain - g) <= 1e-6 * g and abs(a.y2_prime - p) <= 1e-6 * p and a.samples_seen == x.size
    a.reset()
    assert a.gain == 1.0 and a.y2_prime == 1.0
    if profile == "local":      # a stationary input settles at unit output power whatever the target says (agc.c:56-59)
        y = oracle.Agc(1e6, profile=profile, target=0.3).apply(x[:6000])
        assert abs(np.mean(np.abs(y[-2000:]) ** 2) - 1.0) < 0.1


def test_agc_in_chain_runs_between_post_nco_and_pack(oracle):
    """chain with the AGC == chain without it, cf32 out, then agc_apply per 16384-frame input chunk"""
    n = 200000
    raw = synth.raw_stream(n, 48e3, 32, "cs16")
    kw = dict(in_format="cs16", input_rate_hz=48e3, target_rate_hz=12e3, shift_hz=1e3)
    plain = oracle.Chain(out_format="cf32", **kw)
    a = oracle.Agc(12e3)
    outs = []
    rb = np.ascontiguousarray(raw).view(np.uint8)
    for b in range(0, n, 16384):
        y = plain.process(rb[b * 4:(b + 16384) * 4]).view(np.complex64)
        if y.size:
            outs.append(a.apply(y))
    want = np.concatenate(outs)
    got = oracle.Chain(out_format="cf32", agc=True, **kw).process(raw).view(np.complex64)
    assert np.array_equal(got.view(np.float32), want.view(np.float32))
    packed = oracle.Chain(out_format="cs16", agc=True, **kw).process(raw)
    assert np.array_equal(packed, oracle.from_cf32(want, "cs16"))


def test_pipelined_three_thread_chain_equals_single_thread(oracle):
    """bench.py's cpu_baseline arrangement (pre / resampler / post threads) changes no output byte"""
    raw = synth.raw_stream(700001, 10e6, 46, "cs16")
    kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=10e6, target_rate_hz=2.4e6, dc_block=True,
              iq_correct=True, iq_mag=0.01, iq_phase=-0.005, filters=(("passband", 158.5e3, 113e3),), filter_taps=1025, agc=True)
    assert np.array_equal(oracle.Chain(**kw).process(raw), oracle.Chain(**kw).process_pipelined(raw))
    kw = dict(in_format="cu8", out_format="cf32", input_rate_hz=1e6, target_rate_hz=3.3e6, shift_hz=1e5, shift_after_resample=True)
    raw = synth.raw_stream(200000, 1e6, 47, "cu8")
    assert np.array_equal(oracle.Chain(**kw).process(raw), oracle.Chain(**kw).process_pipelined(raw))
