"""ctypes front-end to the parity oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package (iq_tool_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

FMT = dict(cu8=8, cs8=9, cu16=10, cs16=11, cs24=12, cu32=13, cs32=14, cf32=15, sc16q11=16)
FILT = dict(none=0, lowpass=1, highpass=2, passband=3, stopband=4)
IMPL = dict(auto=0, fir=1, fft=2)
BYTES = {8: 2, 9: 2, 10: 4, 11: 4, 16: 4, 12: 6, 13: 8, 14: 8, 15: 8}
# numpy view dtype for the interleaved scalar components of each format (cs24 stays bytes)
NP_DTYPE = {8: np.uint8, 9: np.int8, 10: np.uint16, 11: np.int16, 16: np.int16, 12: np.uint8,
            13: np.uint32, 14: np.int32, 15: np.float32}


class FilterReq(C.Structure):
    _fields_ = [("type", C.c_int), ("f1_hz", C.c_float), ("f2_hz", C.c_float)]


class FilterCfg(C.Structure):
    _fields_ = [("n_req", C.c_int), ("req", FilterReq * 5),
                ("transition_width_hz", C.c_float), ("attenuation_db", C.c_float),
                ("filter_taps", C.c_int), ("impl_request", C.c_int), ("fft_size", C.c_int)]


class ChainDesc(C.Structure):
    _fields_ = [("in_format", C.c_int), ("out_format", C.c_int),
                ("input_rate_hz", C.c_double), ("target_rate_hz", C.c_double),
                ("gain", C.c_float),
                ("shift_hz", C.c_double), ("shift_after_resample", C.c_int),
                ("dc_block_enable", C.c_int),
                ("iq_correct_enable", C.c_int), ("iq_mag", C.c_float), ("iq_phase", C.c_float),
                ("no_resample", C.c_int),
                ("filter", FilterCfg),
                ("dc_f32_literal", C.c_int),
                ("agc_enable", C.c_int), ("agc_target", C.c_float), ("agc_clock", C.c_int), ("agc_profile", C.c_int)]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    need = force or not all(os.path.exists(os.path.join(HERE, f))
                            for f in ("liboracle.so", "liboracle_fast.so"))
    src_m = max(os.path.getmtime(os.path.join(HERE, f)) for f in ("iq_oracle.c", "iq_oracle.h"))
    for f in ("liboracle.so", "liboracle_fast.so"):
        p = os.path.join(HERE, f)
        if os.path.exists(p) and os.path.getmtime(p) < src_m:
            need = True
    if need or (os.path.isdir("/root/reference/src")
                and not os.path.exists(os.path.join(HERE, "_ref", "libsampleconvert_ref.so"))):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)


def _proto(lib):
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    P = lambda name, res, args: (setattr(getattr(lib, name), "restype", res),
                                 setattr(getattr(lib, name), "argtypes", args))
    P("orc_bytes_per_sample", sz, [C.c_int])
    P("orc_convert_block_to_cf32", C.c_int, [vp, vp, sz, C.c_int, C.c_float])
    P("orc_convert_cf32_to_block", C.c_int, [vp, vp, sz, C.c_int])
    P("orc_nco_constrain", u32, [C.c_float])
    P("orc_nco_create", vp, [])
    P("orc_nco_destroy", None, [vp])
    P("orc_nco_set_frequency", None, [vp, C.c_float])
    P("orc_nco_set_phase", None, [vp, C.c_float])
    P("orc_nco_get_dtheta_u32", u32, [vp])
    P("orc_nco_get_theta_u32", u32, [vp])
    P("orc_nco_table", C.POINTER(C.c_float), [vp])
    P("orc_nco_mix_block", None, [vp, C.c_int, vp, vp, sz])
    P("orc_dcblock_create", vp, [C.c_float, C.c_int])
    P("orc_dcblock_destroy", None, [vp])
    P("orc_dcblock_reset", None, [vp])
    P("orc_dcblock_apply", None, [vp, vp, sz])
    P("orc_iq_correct_apply", None, [vp, sz, C.c_float, C.c_float])
    P("orc_kaiser_beta_As", C.c_float, [C.c_float])
    P("orc_besseli0", C.c_double, [C.c_double])
    P("orc_kaiser_window", C.c_double, [C.c_uint, C.c_uint, C.c_double])
    P("orc_firdes_kaiser", None, [C.c_uint, C.c_float, C.c_float, C.c_float, vp])
    P("orc_estimate_req_filter_len", C.c_uint, [C.c_float, C.c_float])
    P("orc_msresamp_create", vp, [C.c_float, C.c_float])
    P("orc_msresamp_destroy", None, [vp])
    P("orc_msresamp_reset", None, [vp])
    P("orc_msresamp_execute", None, [vp, vp, C.c_uint, vp, C.POINTER(C.c_uint)])
    P("orc_msresamp_is_interp", C.c_int, [vp])
    P("orc_msresamp_num_stages", C.c_uint, [vp])
    P("orc_msresamp_stage_m", C.c_uint, [vp, C.c_uint])
    P("orc_msresamp_stage_taps", C.POINTER(C.c_float), [vp, C.c_uint])
    P("orc_msresamp_rate_arb", C.c_float, [vp])
    P("orc_msresamp_step", u32, [vp])
    P("orc_msresamp_arb_proto", C.POINTER(C.c_float), [vp])
    P("orc_filter_create", vp, [C.POINTER(FilterCfg), C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int)])
    P("orc_filter_destroy", None, [vp])
    P("orc_filter_reset", None, [vp])
    P("orc_filter_is_post", C.c_int, [vp])
    P("orc_filter_impl", C.c_int, [vp])
    P("orc_filter_block_size", C.c_uint, [vp])
    P("orc_filter_ntaps", C.c_uint, [vp])
    P("orc_filter_taps", C.POINTER(C.c_float), [vp])
    P("orc_filter_apply", C.c_uint, [vp, vp, C.c_uint, vp])
    P("orc_agc_create", vp, [C.c_float, C.c_double, C.c_int])
    P("orc_agc_create_profile", vp, [C.c_int, C.c_float, C.c_double, C.c_int])
    P("orc_agc_y2_prime", C.c_float, [vp])
    P("orc_agc_destroy", None, [vp])
    P("orc_agc_reset", None, [vp])
    P("orc_agc_set_wall_time", None, [vp, C.c_double])
    P("orc_agc_apply", None, [vp, vp, C.c_uint])
    P("orc_agc_is_locked", C.c_int, [vp])
    P("orc_agc_gain", C.c_float, [vp])
    P("orc_agc_peak_memory", C.c_float, [vp])
    P("orc_agc_samples_seen", C.c_uint64, [vp])
    P("orc_chain_agc", vp, [vp])
    P("orc_chain_create", vp, [C.POINTER(ChainDesc), C.POINTER(C.c_int)])
    P("orc_chain_destroy", None, [vp])
    P("orc_chain_reset", None, [vp])
    P("orc_chain_set_iq_factors", None, [vp, C.c_float, C.c_float])
    P("orc_chain_ratio", C.c_float, [vp])
    P("orc_chain_max_out_frames", sz, [vp, sz])
    P("orc_chain_process", sz, [vp, vp, sz, vp, vp])
    P("orc_chain_process_pipelined", sz, [vp, vp, sz, vp])
    P("orc_iqopt_create", vp, [])
    P("orc_iqopt_destroy", None, [vp])
    P("orc_iqopt_seed", None, [vp, C.c_uint32])
    P("orc_iqopt_set_factors", None, [vp, C.c_float, C.c_float])
    P("orc_iqopt_get_factors", None, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)])
    P("orc_iqopt_power_range", C.c_float, [vp])
    P("orc_iqopt_metric", C.c_float, [vp, vp, C.c_float, C.c_float])
    P("orc_iqopt_run", C.c_int, [vp, vp, C.c_double])
    return lib


_libs = {}


def lib(fast=False):
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name not in _libs:
        path = os.path.join(HERE, name)
        # (tools/sanitize_host.py: the same source under ASan + UBSan for the sanitized run of the CPU suite)
        if not fast and os.environ.get("IQGPU_ORACLE_LIB"):
            path = os.environ["IQGPU_ORACLE_LIB"]
        if not os.path.exists(path):
            build()
        _libs[name] = _proto(C.CDLL(path))
    return _libs[name]


def ref_lib():
    """The reference's own sample_convert.c (oracle/_ref), or None when not built."""
    path = os.path.join(HERE, "_ref", "libsampleconvert_ref.so")
    if not os.path.exists(path):
        return None
    if "ref" not in _libs:
        r = C.CDLL(path)
        r.get_bytes_per_sample.restype = C.c_size_t
        r.get_bytes_per_sample.argtypes = [C.c_int]
        r.convert_block_to_cf32.restype = C.c_bool
        r.convert_block_to_cf32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_float]
        r.convert_cf32_to_block.restype = C.c_bool
        r.convert_cf32_to_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _libs["ref"] = r
    return _libs["ref"]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fmt_id(f):
    return FMT[f] if isinstance(f, str) else int(f)


# ---------------------------------------------------------------- operators (numpy in/out)
def to_cf32(raw, fmt, gain=1.0, L=None):
    L = L or lib()
    fmt = fmt_id(fmt)
    raw = np.ascontiguousarray(raw)
    n = raw.nbytes // BYTES[fmt]
    out = np.empty(n, np.complex64)
    ok = L.orc_convert_block_to_cf32(_ptr(raw), _ptr(out), n, fmt, gain)
    assert ok
    return out


def from_cf32(x, fmt, L=None):
    L = L or lib()
    fmt = fmt_id(fmt)
    x = np.ascontiguousarray(x, np.complex64)
    out = np.empty(x.size * BYTES[fmt], np.uint8)
    ok = L.orc_convert_cf32_to_block(_ptr(x), _ptr(out), x.size, fmt)
    assert ok
    return out.view(NP_DTYPE[fmt])


def ref_to_cf32(raw, fmt, gain=1.0):
    R = ref_lib()
    fmt = fmt_id(fmt)
    raw = np.ascontiguousarray(raw)
    n = raw.nbytes // BYTES[fmt]
    out = np.empty(n, np.complex64)
    assert R.convert_block_to_cf32(_ptr(raw), _ptr(out), n, fmt, gain)
    return out


def ref_from_cf32(x, fmt):
    R = ref_lib()
    fmt = fmt_id(fmt)
    x = np.ascontiguousarray(x, np.complex64)
    out = np.empty(x.size * BYTES[fmt], np.uint8)
    assert R.convert_cf32_to_block(_ptr(x), _ptr(out), x.size, fmt)
    return out.view(NP_DTYPE[fmt])


class Nco:
    def __init__(self, dtheta, L=None):
        self.L = L or lib()
        self.q = self.L.orc_nco_create()
        self.L.orc_nco_set_frequency(self.q, dtheta)

    def __del__(self):
        self.L.orc_nco_destroy(self.q)

    @property
    def dtheta_u32(self):
        return self.L.orc_nco_get_dtheta_u32(self.q)

    @property
    def theta_u32(self):
        return self.L.orc_nco_get_theta_u32(self.q)

    def table(self):
        return np.ctypeslib.as_array(self.L.orc_nco_table(self.q), (1024,)).copy()

    def reset(self):
        self.L.orc_nco_set_phase(self.q, 0.0)

    def mix(self, x, up=True):
        x = np.ascontiguousarray(x, np.complex64)
        y = np.empty_like(x)
        self.L.orc_nco_mix_block(self.q, int(up), _ptr(x), _ptr(y), x.size)
        return y


class DcBlock:
    def __init__(self, alpha, literal=False, L=None):
        self.L = L or lib()
        self.q = self.L.orc_dcblock_create(alpha, int(literal))

    def __del__(self):
        self.L.orc_dcblock_destroy(self.q)

    def reset(self):
        self.L.orc_dcblock_reset(self.q)

    def apply(self, x):
        y = np.array(x, np.complex64, copy=True)
        self.L.orc_dcblock_apply(self.q, _ptr(y), y.size)
        return y


def iq_correct(x, mag, phase, L=None):
    L = L or lib()
    y = np.array(x, np.complex64, copy=True)
    L.orc_iq_correct_apply(_ptr(y), y.size, mag, phase)
    return y


def firdes_kaiser(n, fc, As, mu=0.0, L=None):
    L = L or lib()
    h = np.empty(n, np.float32)
    L.orc_firdes_kaiser(n, fc, As, mu, _ptr(h))
    return h


class MsResamp:
    def __init__(self, r, As=60.0, L=None):
        self.L = L or lib()
        self.q = self.L.orc_msresamp_create(np.float32(r), As)
        assert self.q

    def __del__(self):
        self.L.orc_msresamp_destroy(self.q)

    def reset(self):
        self.L.orc_msresamp_reset(self.q)

    @property
    def interp(self):
        return bool(self.L.orc_msresamp_is_interp(self.q))

    @property
    def S(self):
        return self.L.orc_msresamp_num_stages(self.q)

    @property
    def step(self):
        return self.L.orc_msresamp_step(self.q)

    @property
    def rate_arb(self):
        return self.L.orc_msresamp_rate_arb(self.q)

    def stage_m(self, k):
        return self.L.orc_msresamp_stage_m(self.q, k)

    def stage_taps(self, k):
        m = self.stage_m(k)
        return np.ctypeslib.as_array(self.L.orc_msresamp_stage_taps(self.q, k), (4 * m + 1,)).copy()

    def arb_proto(self):
        return np.ctypeslib.as_array(self.L.orc_msresamp_arb_proto(self.q), (2 * 7 * 256,)).copy()

    def execute(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        r = max(1.0, float(2 ** self.S * 2.0) if self.interp else 1.0)
        y = np.empty(int(np.ceil(x.size * r)) + 256, np.complex64)
        ny = C.c_uint(0)
        self.L.orc_msresamp_execute(self.q, _ptr(x), x.size, _ptr(y), C.byref(ny))
        return y[:ny.value].copy()


def make_filter_cfg(reqs=(), transition_width_hz=0.0, attenuation_db=0.0, filter_taps=0,
                    impl="auto", fft_size=0, cls=FilterCfg, req_cls=FilterReq):
    cfg = cls()
    cfg.n_req = len(reqs)
    for i, (t, f1, f2) in enumerate(reqs):
        cfg.req[i] = req_cls(FILT[t] if isinstance(t, str) else t, f1, f2)
    cfg.transition_width_hz = transition_width_hz
    cfg.attenuation_db = attenuation_db
    cfg.filter_taps = filter_taps
    cfg.impl_request = IMPL[impl] if isinstance(impl, str) else impl
    cfg.fft_size = fft_size
    return cfg


class Filter:
    def __init__(self, cfg, input_rate, target_rate, no_resample=False, L=None):
        self.L = L or lib()
        err = C.c_int(0)
        self.q = self.L.orc_filter_create(C.byref(cfg), input_rate, target_rate, int(no_resample), C.byref(err))
        self.err = err.value
        if not self.q:
            raise ValueError("orc_filter_create failed: %d" % self.err)

    def __del__(self):
        if getattr(self, "q", None):
            self.L.orc_filter_destroy(self.q)

    post = property(lambda s: bool(s.L.orc_filter_is_post(s.q)))
    impl = property(lambda s: s.L.orc_filter_impl(s.q))
    block = property(lambda s: s.L.orc_filter_block_size(s.q))
    ntaps = property(lambda s: s.L.orc_filter_ntaps(s.q))

    def taps(self):
        n = self.ntaps
        return np.ctypeslib.as_array(self.L.orc_filter_taps(self.q), (2 * n,)).copy().view(np.complex64)

    def reset(self):
        self.L.orc_filter_reset(self.q)

    def apply(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        y = np.empty(x.size + self.block + 1, np.complex64)
        n = self.L.orc_filter_apply(self.q, _ptr(x), x.size, _ptr(y))
        return y[:n].copy()


class Agc:
    """agc_create / agc_apply / agc_reset (src/agc.c); for the "digital" profile apply() takes ONE chunk,
    the RMS profiles "dx" / "local" (liquid agc_crcf) are per-sample loops and do not care"""
    PROFILES = {"dx": 1, "local": 2, "digital": 3}

    def __init__(self, sample_rate, target=0.0, clock="samples", L=None, handle=None, profile="digital"):
        self.L = L or lib()
        self.own = handle is None
        self.q = handle or self.L.orc_agc_create_profile(self.PROFILES[profile], target, sample_rate, {"samples": 0, "wall": 1}[clock])

    def __del__(self):
        if getattr(self, "q", None) and self.own:
            self.L.orc_agc_destroy(self.q)

    def reset(self):
        self.L.orc_agc_reset(self.q)

    def set_wall_time(self, t):
        self.L.orc_agc_set_wall_time(self.q, t)

    def apply(self, x):
        y = np.array(x, np.complex64, copy=True)
        self.L.orc_agc_apply(self.q, _ptr(y), y.size)
        return y

    def apply_chunked(self, x, chunk=16384):
        return np.concatenate([self.apply(x[i:i + chunk]) for i in range(0, len(x), chunk)]) if len(x) else np.zeros(0, np.complex64)

    locked = property(lambda s: bool(s.L.orc_agc_is_locked(s.q)))
    gain = property(lambda s: s.L.orc_agc_gain(s.q))
    peak_memory = property(lambda s: s.L.orc_agc_peak_memory(s.q))
    y2_prime = property(lambda s: s.L.orc_agc_y2_prime(s.q))
    samples_seen = property(lambda s: s.L.orc_agc_samples_seen(s.q))


def make_desc(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5,
              gain=1.0, shift_hz=0.0, shift_after_resample=False, dc_block=False,
              iq_correct=False, iq_mag=0.0, iq_phase=0.0, no_resample=False,
              filters=(), transition_width_hz=0.0, attenuation_db=0.0, filter_taps=0,
              filter_impl="auto", fft_size=0, dc_f32_literal=False,
              agc=False, agc_target=0.0, agc_clock="samples", agc_chunk_frames=0, agc_profile="digital"):
    assert agc_chunk_frames in (0, 16384), "the oracle chunks like the reference: 16384 frames"
    d = ChainDesc()
    d.agc_enable = int(bool(agc))
    d.agc_target = agc_target
    d.agc_profile = Agc.PROFILES[agc_profile] if isinstance(agc_profile, str) else agc_profile
    d.agc_clock = {"samples": 0, "wall": 1}[agc_clock] if isinstance(agc_clock, str) else agc_clock
    d.in_format = fmt_id(in_format)
    d.out_format = fmt_id(out_format)
    d.input_rate_hz = input_rate_hz
    d.target_rate_hz = target_rate_hz
    d.gain = gain
    d.shift_hz = shift_hz
    d.shift_after_resample = int(shift_after_resample)
    d.dc_block_enable = int(dc_block)
    d.iq_correct_enable = int(iq_correct)
    d.iq_mag = iq_mag
    d.iq_phase = iq_phase
    d.no_resample = int(no_resample)
    d.filter = make_filter_cfg(filters, transition_width_hz, attenuation_db, filter_taps,
                               filter_impl, fft_size)
    d.dc_f32_literal = int(dc_f32_literal)
    return d


class Chain:
    """Oracle for pre_processor -> resampler -> post_processor on one stream."""

    def __init__(self, L=None, **kw):
        self.L = L or lib()
        self.desc = make_desc(**kw)
        err = C.c_int(0)
        self.c = self.L.orc_chain_create(C.byref(self.desc), C.byref(err))
        if not self.c:
            raise ValueError("orc_chain_create failed: %d" % err.value)
        self.ibps = BYTES[self.desc.in_format]
        self.obps = BYTES[self.desc.out_format]

    def __del__(self):
        if getattr(self, "c", None):
            self.L.orc_chain_destroy(self.c)

    def reset(self):
        self.L.orc_chain_reset(self.c)

    def set_iq_factors(self, mag, phase):
        self.L.orc_chain_set_iq_factors(self.c, mag, phase)

    @property
    def agc(self):
        h = self.L.orc_chain_agc(self.c)
        return Agc(0, L=self.L, handle=h) if h else None

    @property
    def ratio(self):
        return self.L.orc_chain_ratio(self.c)

    def max_out_frames(self, n):
        return self.L.orc_chain_max_out_frames(self.c, n)

    def process_pipelined(self, raw):
        """three concurrent stage threads, as the reference runs the chain (src/pipeline.c:96-116)"""
        raw = np.ascontiguousarray(raw)
        n = raw.nbytes // self.ibps
        cap = self.max_out_frames(n) + 16384
        out = np.empty(cap * self.obps, np.uint8)
        k = self.L.orc_chain_process_pipelined(self.c, _ptr(raw), n, _ptr(out))
        return out[:k * self.obps].view(NP_DTYPE[self.desc.out_format]).copy()

    def process(self, raw, want_cf32=False):
        raw = np.ascontiguousarray(raw)
        n = raw.nbytes // self.ibps
        cap = self.max_out_frames(n) + 16384
        out = np.empty(cap * self.obps, np.uint8)
        tap = np.empty(cap, np.complex64) if want_cf32 else None
        k = self.L.orc_chain_process(self.c, _ptr(raw), n, _ptr(out), _ptr(tap) if want_cf32 else None)
        res = out[:k * self.obps].view(NP_DTYPE[self.desc.out_format]).copy()
        if want_cf32:
            return res, tap[:k].copy()
        return res


class IqOptimizer:
    """orc_iqopt_*: iq_correct_run_optimization (src/iq_correct.c:154-219) on 1024-sample cf32 blocks"""

    def __init__(self, seed=1, L=None):
        self.L = L or lib()
        self.q = self.L.orc_iqopt_create()
        self.L.orc_iqopt_seed(self.q, seed)

    def __del__(self):
        if getattr(self, "q", None):
            self.L.orc_iqopt_destroy(self.q)
            self.q = None

    def set_factors(self, mag, phase):
        self.L.orc_iqopt_set_factors(self.q, mag, phase)

    def factors(self):
        m, p = C.c_float(0), C.c_float(0)
        self.L.orc_iqopt_get_factors(self.q, C.byref(m), C.byref(p))
        return m.value, p.value

    def metric(self, block, mag, phase):
        b = np.ascontiguousarray(block, np.complex64)
        return self.L.orc_iqopt_metric(self.q, b.ctypes.data_as(C.c_void_p), mag, phase)

    def run(self, block, now_sec):
        b = np.ascontiguousarray(block, np.complex64)
        return bool(self.L.orc_iqopt_run(self.q, b.ctypes.data_as(C.c_void_p), now_sec))

    def power_range(self):
        return self.L.orc_iqopt_power_range(self.q)
