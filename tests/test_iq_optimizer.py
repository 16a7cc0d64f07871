"""I/Q imbalance optimiser: the product's host code (libiqgpu, iq_optimizer.cpp) against the oracle's
restatement of src/iq_correct.c:154-219, 315-393 and against an independent numpy formulation.  Host-only code in
the reference and here, so these run without a GPU; the probe / service path is in the gpu section below."""
import numpy as np
import pytest

N = 1024


def imbalanced_block(seed, gain_err=0.04, phase_err=0.03, tones=((0.11, 0.5), (-0.23, 0.2), (0.31, 0.1)), noise=1e-3):
    rng = np.random.default_rng(seed)
    n = np.arange(N)
    x = sum(a * np.exp(2j * np.pi * f * n + 1j * rng.uniform(0, 6.28)) for f, a in tones)
    x = x + noise * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
    # receiver imbalance: Q gain and phase skew (what the correction undoes approximately)
    y = x.real + 1j * ((1 + gain_err) * (x.imag * np.cos(phase_err) + x.real * np.sin(phase_err)))
    return y.astype(np.complex64)


def np_metric(block, mag, phase):
    """numpy restatement of _calculate_power_spectrum + _calculate_imbalance_metric, float64 transform"""
    b = block.astype(np.complex64)
    re = b.real * np.float32(1.0 + np.float32(mag))
    im = b.imag + np.float32(phase) * b.real
    i = np.arange(N, dtype=np.float32)
    w = (np.float32(0.54) - np.float32(0.46) * np.cos(np.float32(2.0) * np.float32(np.pi) * i / np.float32(N - 1))).astype(np.float32)
    X = np.fft.fftshift(np.fft.fft((re * w).astype(np.float64) + 1j * (im * w).astype(np.float64)))
    S = (20.0 * np.log10(np.abs(X).astype(np.float32) / np.float32(N) + np.float32(1e-12))).astype(np.float32)
    lo, hi = int(np.float32(0.05) * 512), int(np.float32(0.95) * 512)
    idx = np.arange(lo, hi)
    p_neg, p_pos = S[idx], S[N - 1 - idx]
    m = (p_pos > -80.0) | (p_neg > -80.0)
    d = (p_pos - p_neg)[m].astype(np.float64)
    return float((d * d).sum()), S


@pytest.fixture(scope="module")
def prod():
    import iq_tool_amd
    iq_tool_amd.load()
    return iq_tool_amd


def test_metric_matches_oracle_and_numpy(prod, oracle):
    for seed in range(6):
        blk = imbalanced_block(seed)
        po, oo = prod.IqOptimizer(seed=1), oracle.IqOptimizer(seed=1)
        for mag, ph in ((0.0, 0.0), (0.01, -0.005), (-0.04, -0.03), (0.0001, 0.0001)):
            a, b = po.metric(blk, mag, ph), oo.metric(blk, mag, ph)
            c, _ = np_metric(blk, mag, ph)
            assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (seed, mag, ph, a, b)
            assert abs(b - c) <= 2e-4 * max(1.0, abs(c)), (seed, mag, ph, b, c)


def test_bounds_and_window_constants(prod):
    # bins [25, 486) of 512 and the Hamming window of iq_correct.c:122-124: a pure tone in bin +100 and nothing else
    n = np.arange(N)
    blk = np.exp(2j * np.pi * 100 * n / N).astype(np.complex64)
    o = prod.IqOptimizer(seed=3)
    m = o.metric(blk, 0.0, 0.0)
    ref, S = np_metric(blk, 0.0, 0.0)
    assert abs(m - ref) <= 1e-3 * ref
    assert np.argmax(S) == 512 + 100                      # shifted spectrum: DC at 512
    # a tone outside the inner 90 % of bins (bin 500 > 486) is ignored on both sides -> its own peak adds nothing
    out = np.exp(2j * np.pi * 500 * n / N).astype(np.complex64) * 1e-3
    assert abs(o.metric((blk + out).astype(np.complex64), 0.0, 0.0) - m) <= 0.05 * m


def test_hill_climb_on_frozen_seeds_matches_oracle(prod, oracle):
    """25 random +-1e-4 steps, keep improvements, 5 % smoothing (iq_correct.c:191-216): same seed, same block ->
    same factors.  A candidate whose metric ties the best to float rounding may be taken by one side only; each
    such flip moves a factor by at most 0.05 * 1e-4."""
    worst = 0.0
    exact = 0
    for seed in range(1, 13):
        blk = imbalanced_block(100 + seed)
        po, oo = prod.IqOptimizer(seed=seed), oracle.IqOptimizer(seed=seed)
        t = 1.0
        for it in range(8):
            up, uo = po.run_optimization(blk, t), oo.run(blk, t)
            assert up and uo
            t += 0.6
        (pm, pp), (om, oph) = po.factors(), oo.factors()
        d = max(abs(pm - om), abs(pp - oph))
        worst = max(worst, d)
        exact += d <= 1e-7
        st = po.stats()
        assert st["runs"] == 8 and st["final_metric"] >= st["initial_metric"]
    assert worst <= 4 * 0.05 * 1e-4 + 1e-7, worst
    assert exact >= 9, exact


def test_gates_interval_and_power(prod, oracle):
    blk = imbalanced_block(5)
    po, oo = prod.IqOptimizer(seed=2), oracle.IqOptimizer(seed=2)
    # 500 ms interval (IQ_CORRECTION_INTERVAL_MS) on the caller's clock, last run initially at 0
    assert not po.run_optimization(blk, 0.3) and not oo.run(blk, 0.3)
    assert po.run_optimization(blk, 0.6) and oo.run(blk, 0.6)
    assert not po.run_optimization(blk, 1.0) and not oo.run(blk, 1.0)       # 400 ms later
    assert po.run_optimization(blk, 1.2) and oo.run(blk, 1.2)
    # flat noise: peak-to-average below 20 dB -> skipped, and the interval does NOT restart (iq_correct.c:170-178)
    rng = np.random.default_rng(0)
    noise = (0.1 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(np.complex64)
    f0 = po.factors()
    assert not po.run_optimization(noise, 5.0) and not oo.run(noise, 5.0)
    assert po.factors() == f0
    assert po.stats()["skipped_power"] == 1 and po.stats()["power_range_db"] < 20.0
    assert abs(po.stats()["power_range_db"] - oo.power_range()) < 0.05
    assert po.run_optimization(blk, 5.1)                                     # 5.1 - 1.2 >= 0.5: not restarted by the skip


def test_climb_reduces_image_and_injected_rng(prod):
    # the utility is the squared dB asymmetry: raising it suppresses the image of a one-sided spectrum
    blk = imbalanced_block(77, gain_err=0.02, phase_err=0.01, tones=((0.2, 0.7),), noise=1e-4)
    calls = []

    def rng():
        calls.append(1)
        return 1.0 if (len(calls) * 2654435761) & 0x10000 else -1.0
    o = prod.IqOptimizer(rng=rng)
    m0 = o.metric(blk, 0.0, 0.0)
    t = 1.0
    for _ in range(400):
        assert o.run_optimization(blk, t)
        t += 1.0
    assert len(calls) == 400 * 50                       # gain direction then phase direction, 25 passes
    mag, ph = o.factors()
    assert o.metric(blk, mag, ph) > m0
    assert abs(mag) > 1e-4 or abs(ph) > 1e-4


@pytest.mark.gpu
def test_probe_block_is_the_pre_processed_chunk_head(gpu, oracle):
    from iq_tool_amd import synth
    raw = synth.raw_stream(1 << 17, 2.4e6, 4, "cs16")
    kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3,
              dc_block=True, iq_correct=True, iq_mag=0.01, iq_phase=-0.005)
    ch = gpu.Chain(**kw)
    ch.enable_iq_probe()
    assert ch.read_iq_probe() is None
    rb = raw.view(np.uint8)
    half = rb.size // 2
    ch.process(rb[:half])
    b1 = ch.read_iq_probe()
    ch.process(rb[half:])
    b2 = ch.read_iq_probe()
    ch.process(rb[:4 * 100])                           # a call shorter than 1024 frames leaves the block alone
    assert np.array_equal(ch.read_iq_probe(), b2)
    # the oracle's pre-processing of the same stream, cf32, no resampler (pre_processor_apply_chain)
    pre = oracle.Chain(in_format="cs16", out_format="cf32", input_rate_hz=2.4e6, target_rate_hz=2.4e6, no_resample=True,
                       shift_hz=200e3, dc_block=True, iq_correct=True, iq_mag=0.01, iq_phase=-0.005)
    want = pre.process(raw).view(np.complex64)
    n_half = half // 4
    assert np.abs(b1 - want[:1024]).max() <= 1e-5
    assert np.abs(b2 - want[n_half:n_half + 1024]).max() <= 1e-5


@pytest.mark.gpu
def test_service_loop_feeds_the_chain_like_the_reference_threads(gpu, oracle):
    """pre-processor -> 1024-sample hand-off -> optimiser -> factors -> next chunks (src/pipeline.c:468-476,
    src/utility_threads.c:35-47, src/iq_correct.c:141-152), on the stream clock; the oracle does the same."""
    from iq_tool_amd import synth
    rate = 2.4e6
    n_call = 1 << 18
    calls = 12
    raw = synth.raw_stream(n_call * calls, rate, 8, "cs16").view(np.uint8)
    kw = dict(in_format="cs16", out_format="cf32", input_rate_hz=rate, target_rate_hz=744187.5, shift_hz=0.0, iq_correct=True)
    g, o = gpu.Chain(**kw), oracle.Chain(**kw)
    pre = oracle.Chain(in_format="cs16", out_format="cf32", input_rate_hz=rate, target_rate_hz=rate, no_resample=True, iq_correct=True)
    g.enable_iq_probe()
    go, oo = gpu.IqOptimizer(seed=11), oracle.IqOptimizer(seed=11)
    t = 10.0
    updates = 0
    for i in range(calls):
        seg = raw[i * n_call * 4:(i + 1) * n_call * 4]
        a = g.process(seg)
        b = o.process(seg)
        assert a.shape == b.shape and np.abs(a.view(np.complex64) - b.view(np.complex64)).max() <= 1e-5, i
        # reference side: the block comes from the pre-processor output with the factors of that call
        blk = pre.process(seg[:1024 * 4]).view(np.complex64)
        pre.reset()
        upd = go.service(g, t)
        if oo.run(blk, t):
            m, p = oo.factors()
            o.set_iq_factors(m, p); pre.set_iq_factors(m, p)
            assert upd
            updates += 1
        else:
            assert not upd
        (gm, gp), (om, op) = go.factors(), oo.factors()
        assert max(abs(gm - om), abs(gp - op)) <= 2 * 0.05 * 1e-4 + 1e-7
        t += n_call / rate * 6          # 0.65 s of wall time per call: every call may run
    assert updates >= calls - 1
