"""numpy-only restatement of the reference's filter design path (src/filter.c:43-92 placement, 138-393 design) with the
liquid-dsp primitives written out from their published definitions (DESIGN.md SPEC B.1 Kaiser, B.4 NCO table).  A THIRD
formulation beside iq_tool_amd/csrc/design.cpp and oracle/iq_oracle.c -- written against the reference source, in
float64 with numpy idioms (np.i0, np.convolve, np.exp), so that a misreading shared by the two C restatements shows.
Test infrastructure only."""
import numpy as np

LOWPASS, HIGHPASS, PASSBAND, STOPBAND = 1, 2, 3, 4
KIND = dict(lowpass=1, highpass=2, passband=3, stopband=4)


def kaiser_beta(As):
    As = abs(As)
    if As > 50.0:
        return 0.1102 * (As - 8.7)
    if As > 21.0:
        return 0.5842 * (As - 21.0) ** 0.4 + 0.07886 * (As - 21.0)
    return 0.0


def firdes_kaiser(n, fc, As):
    """liquid_firdes_kaiser(n, fc, As, mu = 0): sinc(2 fc t) * kaiser(beta) at t = i - (n-1)/2, window argument 2 t / (n-1)"""
    t = np.arange(n, dtype=np.float64) - (n - 1) / 2.0
    beta = kaiser_beta(As)
    r = 2.0 * t / (n - 1)
    w = np.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - r * r))) / np.i0(beta)
    return (np.sinc(2.0 * fc * t) * w)


def estimate_req_filter_len(df, As):
    """(unsigned)((As - 7.95) / (14.26 df)) in float"""
    return int(np.float32((np.float32(As) - np.float32(7.95)) / (np.float32(14.26) * np.float32(df))))


def nco_table_phasors(dtheta_rad, n):
    """nco_crcf LIQUID_NCO: uint32 phase, 1024-entry sine table, index = rounded top 10 bits (SPEC B.4)"""
    f = np.float32(np.float32(dtheta_rad) * np.float32(0.159154943091895))
    frac = np.float32(f - np.floor(f))
    d = np.uint64(np.float32(frac) * np.float32(4294967296.0)) & np.uint64(0xFFFFFFFF)
    theta = (np.arange(n, dtype=np.uint64) * d) & np.uint64(0xFFFFFFFF)
    idx = ((theta + np.uint64(1 << 21)) >> np.uint64(22)) & np.uint64(1023)
    tab = np.sin(np.float32(2.0 * np.pi) * np.arange(1024, dtype=np.float32) / np.float32(1024.0)).astype(np.float32)
    s = tab[idx.astype(np.int64)]
    c = tab[(idx.astype(np.int64) + 256) & 1023]
    return c.astype(np.float64) + 1j * s.astype(np.float64)


def design(filters, input_rate, target_rate, no_resample=False, filter_taps=0, transition_width_hz=0.0,
           attenuation_db=0.0, impl="auto", fft_size=0):
    """returns dict(post, taps (complex128), is_complex, impl, block) or raises ValueError like the fatal paths"""
    reqs = [(KIND[t] if isinstance(t, str) else t, np.float32(f1), np.float32(f2)) for t, f1, f2 in filters]
    out_rate = input_rate if no_resample else target_rate
    # ---- _configure_filter_stage (filter.c:43-92)
    post = False
    if reqs and not no_resample and out_rate < input_rate:
        mx = np.float32(0.0)
        for t, f1, f2 in reqs:
            cur = abs(f1) if t in (LOWPASS, HIGHPASS) else np.float32(abs(f1) + np.float32(f2 / np.float32(2.0)))
            mx = max(mx, cur)
        if mx > out_rate / 2.0:
            raise ValueError("filter beyond output Nyquist")
        post = True
    fs = np.float32(out_rate if post else input_rate)
    As = np.float32(attenuation_db) if attenuation_db > 0 else np.float32(60.0)
    master = np.array([1.0 + 0.0j])
    by_peak = False
    is_complex = False
    if filter_taps and filter_taps % 2 == 0:
        filter_taps += 1                                            # src/config.c:233-236
    for t, f1, f2 in reqs:
        if t != LOWPASS:
            by_peak = True
        if filter_taps > 0:
            n = int(filter_taps)
        else:
            if transition_width_hz > 0:
                tw = np.float32(transition_width_hz)
            else:
                ref = f1 if t in (LOWPASS, HIGHPASS) else f2
                tw = np.float32(abs(ref) * np.float32(0.25))
            if tw < 1.0:
                tw = np.float32(1.0)
            n = estimate_req_filter_len(np.float32(tw / fs), As)
            if n % 2 == 0:
                n += 1
            if n < 21:
                n = 21
        stage_complex = t == PASSBAND and abs(f1) > 1e-9
        if stage_complex:
            is_complex = True
            proto = firdes_kaiser(n, float(np.float32(np.float32(f2 / np.float32(2.0)) / fs)), float(As))
            fc = np.float32(f1 / fs)
            cur = nco_table_phasors(np.float32(np.float32(2.0 * np.pi) * fc), n) * proto.astype(np.float32).astype(np.float64)
        else:
            if t in (LOWPASS, HIGHPASS):
                proto = firdes_kaiser(n, float(np.float32(f1 / fs)), float(As))
            else:
                proto = firdes_kaiser(n, float(np.float32(np.float32(f2 / fs) / np.float32(2.0))), float(As))
            proto = proto.astype(np.float32).astype(np.float64)
            if t in (HIGHPASS, STOPBAND):                            # _invert_filter_spectrum
                proto = -proto
                proto[(n - 1) // 2] += 1.0
            cur = proto + 0.0j
        master = np.convolve(master, cur)
    if by_peak or is_complex:
        f = np.arange(2048) / 2048.0 - 0.5
        k = np.arange(master.size)
        H = np.exp(-2j * np.pi * np.outer(f, k)) @ master
        mx = np.abs(H).max()
        if mx > 1e-9:
            master = master / mx
    else:
        g = master.real.sum()
        if abs(g) > 1e-9:
            master = master / g
    choice = impl if impl != "auto" else ("fft" if is_complex else "fir")
    block = 0
    if choice == "fft":
        if fft_size > 0:
            block = fft_size // 2
            if block < master.size - 1:
                raise ValueError("fft size too small")
        else:
            block = 1
            while block < master.size - 1:
                block *= 2
            if block < master.size * 2:
                block *= 2
    fi = {("fir", False): 1, ("fir", True): 2, ("fft", False): 3, ("fft", True): 4}[(choice, is_complex)]
    return dict(post=post, taps=master, is_complex=is_complex, impl=fi, block=block)
