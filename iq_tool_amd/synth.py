"""Synthetic I/Q test streams: three complex tones at -150 kHz, +30 kHz and
+200 kHz scaled (0.20, 0.25, 0.15), complex Gaussian noise sigma 0.05, DC offset (0.01, -0.02),
clipped to +-0.999 and quantised to the raw-file sample format with the reference's packing rule
(src/sample_convert.c:40-73).  numpy only; used by bench.py, the harness and the tests."""
import numpy as np

TONES_HZ = (-150e3, 30e3, 200e3)
TONE_AMP = (0.20, 0.25, 0.15)


def complex_signal(n, rate_hz, seed, start=0):
    """cf32 samples [start, start+n) of the stream with this seed (noise is re-seeded per call
    with (seed, start) so arbitrary windows are reproducible)."""
    t = (np.arange(start, start + n, dtype=np.float64)) / float(rate_hz)
    x = np.zeros(n, np.complex128)
    for f, a in zip(TONES_HZ, TONE_AMP):
        x += a * np.exp(2j * np.pi * f * t)
    rng = np.random.default_rng([int(seed), int(start)])
    x += 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x += 0.01 - 0.02j
    x = np.clip(x.real, -0.999, 0.999) + 1j * np.clip(x.imag, -0.999, 0.999)
    return x.astype(np.complex64)


def quantise(x, fmt):
    """cf32 -> interleaved integer frames, float32 arithmetic as convert_cf32_to_block does it."""
    v = np.ascontiguousarray(x, np.complex64).view(np.float32)
    half = np.float32(0.5)
    if fmt in ("cs16", "sc16q11", "cs8"):
        scale, lo, hi, dt = {"cs16": (32767.0, -32768.0, 32767.0, np.int16),
                             "sc16q11": (2048.0, -32768.0, 32767.0, np.int16),
                             "cs8": (127.0, -128.0, 127.0, np.int8)}[fmt]
        w = v * np.float32(scale)
        w = np.where(w > 0, w + half, w - half).astype(np.float32)
        w = np.clip(w, np.float32(lo), np.float32(hi))
        return np.trunc(w).astype(dt)
    if fmt in ("cu8", "cu16"):
        scale, off, hi, dt = {"cu8": (127.0, 127.5, 255.0, np.uint8),
                              "cu16": (32767.0, 32767.5, 65535.0, np.uint16)}[fmt]
        w = (v * np.float32(scale)).astype(np.float32) + np.float32(off)
        w = np.clip(w, np.float32(0.0), np.float32(hi))
        return np.trunc(w + half).astype(dt)
    if fmt == "cf32":
        return v.copy()
    if fmt == "cs32":
        w = np.clip(np.round(v.astype(np.float64) * 2147483647.0), -2147483648.0, 2147483647.0)
        return w.astype(np.int32)
    if fmt == "cu32":
        w = np.clip(np.round(v.astype(np.float64) * 2147483647.0 + 2147483647.5), 0.0, 4294967295.0)
        return w.astype(np.uint32)
    if fmt == "cs24":
        w = np.clip(np.round(v.astype(np.float64) * 8388607.0), -8388608.0, 8388607.0).astype(np.int32)
        b = np.empty((w.size, 3), np.uint8)            # 3 little-endian bytes per component
        b[:, 0] = w & 0xff; b[:, 1] = (w >> 8) & 0xff; b[:, 2] = (w >> 16) & 0xff
        return b.reshape(-1)
    raise ValueError("synth.quantise: format %r not supported" % fmt)


def raw_stream(n, rate_hz, seed, fmt, chunk=1 << 22):
    """n frames of the synthetic stream in the given raw format (generated in chunks)."""
    parts = []
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        parts.append(quantise(complex_signal(m, rate_hz, seed, s), fmt))
    return np.concatenate(parts) if parts else np.zeros(0, np.int16)


def agc_envelope_signal(n, rate_hz, seed):
    """Complex noise whose envelope walks the digital AGC (src/agc.c:105-222) through every branch:
    an early burst while scanning, a burst after the 2 s lock that clips (ratchet), a fade longer
    than the 4 s hang time (creep), then a recovery."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * np.float32(0.05)
    t = np.arange(n) / float(rate_hz)
    env = np.ones(n, np.float32)
    env[(t > 0.5) & (t < 0.8)] = 3.0
    env[(t > 3.0) & (t < 3.2)] = 6.0
    env[t > 4.0] = 0.2
    env[t > 11.0] = 1.5
    return x * env


def hash_stream(n, seed, fmt="cs16", first=0):
    """Frames [first, first + n) of the counter-hash stream the harness generates with `--synthetic FRAMES --synthetic-hash SEED`
    (iq_tool_amd/csrc/harness/iqgpu_run.c hash_fill; shard s of a run uses seed SEED + s and counts from its own frame 0):
    frame k -> splitmix64(seed * 0xD1342543DE82EF95 + k); cs16 takes the two low 16-bit words as signed values >> 2 (quarter
    scale), every other integer format the low bytes of the hash as they are.  Any range of a 2.5 G-frame shard is reproducible
    without the shard."""
    with np.errstate(over="ignore"):
        z = np.arange(first, first + n, dtype=np.uint64) + np.uint64((int(seed) * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    if fmt == "cs16":
        out = np.empty(2 * n, np.int16)
        out[0::2] = (z & np.uint64(0xFFFF)).astype(np.uint16).view(np.int16) >> 2
        out[1::2] = ((z >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.uint16).view(np.int16) >> 2
        return out
    bps = {"cu8": 2, "cs8": 2, "cu16": 4, "sc16q11": 4, "cs24": 6, "cs32": 8, "cu32": 8}[fmt]
    b = z.view(np.uint8).reshape(n, 8)[:, :bps].reshape(-1).copy()
    return b.view({"cu8": np.uint8, "cs8": np.int8, "cu16": np.uint16, "sc16q11": np.int16, "cs24": np.uint8,
                   "cs32": np.int32, "cu32": np.uint32}[fmt])
