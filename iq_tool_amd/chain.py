"""Host-side handle for one I/Q stream on one MI355X: a thin object over the C ABI of
include/iqgpu.h.  It stands where the reference's pre-processor, resampler and post-processor
stage threads stand (src/pipeline.c:436-595); argument names follow AppConfig
(include/app_context.h:66-138)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BYTES_PER_FRAME, FILTER, FILTER_IMPL, FMT, AgcState, ChainDesc, ChainInfo, FilterReq,
                   IqgpuError, Profile, check)

AGC_PROFILE = {"off": 0, "dx": 1, "local": 2, "digital": 3}

_NP_VIEW = {8: np.uint8, 9: np.int8, 10: np.uint16, 11: np.int16, 16: np.int16, 12: np.uint8,
            13: np.uint32, 14: np.int32, 15: np.float32}


def _fmt(f):
    return FMT[f] if isinstance(f, str) else int(f)


def make_desc(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5,
              resample_ratio=0.0, gain=1.0, shift_hz=0.0, shift_after_resample=False,
              dc_block=False, iq_correct=False, iq_mag=0.0, iq_phase=0.0, no_resample=False,
              filters=(), transition_width_hz=0.0, attenuation_db=0.0, filter_taps=0,
              filter_impl="auto", fft_size=0, device=0, block_samples=0,
              agc=False, agc_profile="digital", agc_target=0.0, agc_clock="samples", agc_chunk_frames=0):
    lib = _lib.load()
    _lib.apply_debug_env()          # (the test / bench mirror forwards IQGPU_<NAME> variables to iqgpu_debug_set; the library reads none)
    d = ChainDesc()
    lib.iqgpu_chain_desc_init(C.byref(d))
    d.in_format = _fmt(in_format)
    d.out_format = _fmt(out_format)
    d.input_rate_hz = float(input_rate_hz)
    d.target_rate_hz = float(target_rate_hz)
    d.resample_ratio = float(resample_ratio)
    d.gain = float(gain)
    d.shift_hz = float(shift_hz)
    d.shift_after_resample = int(bool(shift_after_resample))
    d.dc_block_enable = int(bool(dc_block))
    d.iq_correct_enable = int(bool(iq_correct))
    d.iq_mag = float(iq_mag)
    d.iq_phase = float(iq_phase)
    d.no_resample = int(bool(no_resample))
    d.n_filters = len(filters)
    for i, (t, f1, f2) in enumerate(filters[:5]):
        d.filters[i] = FilterReq(FILTER[t] if isinstance(t, str) else int(t), float(f1), float(f2))
    d.transition_width_hz = float(transition_width_hz)
    d.attenuation_db = float(attenuation_db)
    # the reference bumps an even --filter-taps to the next odd number (src/config.c:233-236)
    ft = int(filter_taps)
    if ft != 0 and ft % 2 == 0:
        ft += 1
    d.filter_taps = ft
    d.filter_impl = FILTER_IMPL[filter_impl] if isinstance(filter_impl, str) else int(filter_impl)
    d.fft_size = int(fft_size)
    d.device_ordinal = int(device)
    d.block_samples = int(block_samples)
    d.agc_enable = int(bool(agc))
    d.agc_profile = AGC_PROFILE[agc_profile] if isinstance(agc_profile, str) else int(agc_profile)
    d.agc_target = float(agc_target)
    d.agc_clock = {"samples": 0, "wall": 1}[agc_clock] if isinstance(agc_clock, str) else int(agc_clock)
    d.agc_chunk_frames = int(agc_chunk_frames)
    return d


def design_out_frames(frames_in, **kw):
    """frames a FRESH chain of this description emits for a stream of frames_in frames (iqgpu_design_out_frames: the closed
    form a stitching writer places shard outputs with; no device needed)"""
    d = make_desc(**kw)
    n = C.c_size_t(0)
    check(_lib.load().iqgpu_design_out_frames(C.byref(d), int(frames_in), C.byref(n)))
    return int(n.value)


def bind_thread_to_device(ordinal):
    """iqgpu_bind_thread_to_device: the calling thread onto the NUMA node of HIP device `ordinal` (sysfs only, no HIP call --
    meant to run before the first GPU call and before pinned buffers are allocated).  Returns (node, pci_bus_id, error): node -1
    when the host does not say or nothing could be bound, error None or the library's message."""
    lib = _lib.load()
    _lib.apply_debug_env()
    node, bus = C.c_int(-1), C.create_string_buffer(64)
    rc = lib.iqgpu_device_numa_node(int(ordinal), C.byref(node), bus, 64)
    if rc != 0:
        return -1, "", lib.iqgpu_last_error().decode("utf-8", "replace")
    rc = lib.iqgpu_bind_thread_to_device(int(ordinal), C.byref(node))
    return node.value, bus.value.decode(), (None if rc == 0 else lib.iqgpu_last_error().decode("utf-8", "replace"))


class Chain:
    """pre_processor -> resampler -> post_processor for one stream, on one GPU."""

    def __init__(self, desc=None, **kw):
        self._lib = _lib.load()
        self.desc = desc if desc is not None else make_desc(**kw)
        _lib.apply_debug_env()
        h = C.c_void_p()
        check(self._lib.iqgpu_chain_create(C.byref(self.desc), C.byref(h)))
        self._h = h
        self.device = self.desc.device_ordinal
        self.in_bytes = BYTES_PER_FRAME[self.desc.in_format]
        self.out_bytes = BYTES_PER_FRAME[self.desc.out_format]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.iqgpu_chain_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- description ----
    def info(self):
        i = ChainInfo()
        check(self._lib.iqgpu_chain_get_info(self._h, C.byref(i)))
        return i

    def filter_taps(self):
        n = self._lib.iqgpu_chain_get_filter_taps(self._h, None, 0)
        buf = np.zeros(2 * n, np.float32)
        self._lib.iqgpu_chain_get_filter_taps(self._h, buf.ctypes.data_as(C.c_void_p), n)
        return buf.view(np.complex64)

    def max_out_frames(self, frames_in):
        return self._lib.iqgpu_chain_max_out_frames(self._h, frames_in)

    def next_out_frames(self, frames_in):
        return self._lib.iqgpu_chain_next_out_frames(self._h, frames_in)

    # ---- stream ----
    def process(self, raw):
        """raw: numpy array holding whole frames in in_format (any dtype).  Returns the output
        frames as a numpy array of the output format's component type."""
        raw = np.ascontiguousarray(raw)
        n = raw.nbytes // self.in_bytes
        cap = (self.next_out_frames(n) + 1) * self.out_bytes
        out = np.empty(cap, np.uint8)
        got = C.c_size_t(0)
        check(self._lib.iqgpu_chain_process(self._h, raw.ctypes.data_as(C.c_void_p), n,
                                            out.ctypes.data_as(C.c_void_p), cap, C.byref(got)))
        return out[:got.value * self.out_bytes].view(_NP_VIEW[self.desc.out_format]).copy()

    def process_device(self, d_in, frames_in, d_out, out_capacity_bytes):
        """Both pointers are device addresses (ints) on this chain's GPU; asynchronous."""
        got = C.c_size_t(0)
        check(self._lib.iqgpu_chain_process_device(self._h, C.c_void_p(d_in), frames_in,
                                                   C.c_void_p(d_out), out_capacity_bytes, C.byref(got)))
        return got.value

    def submit(self, in_ptr, frames_in, out_ptr, out_capacity_bytes):
        """Queues H2D copy -> kernels -> D2H copy of one batch (host addresses, preferably pinned) and
        returns (frames_out, ticket) at once; collect(ticket) waits for the output bytes."""
        got, ticket = C.c_size_t(0), C.c_uint64(0)
        check(self._lib.iqgpu_chain_submit(self._h, C.c_void_p(in_ptr), frames_in, C.c_void_p(out_ptr),
                                           out_capacity_bytes, C.byref(got), C.byref(ticket)))
        return got.value, ticket.value

    def collect(self, ticket):
        check(self._lib.iqgpu_chain_collect(self._h, C.c_uint64(ticket)))

    def process_pipelined(self, raw, batch_frames):
        """process() through submit / collect: the stream in batches of batch_frames with up to
        iqgpu_chain_pipeline_depth() of them in flight, pinned staging buffers on both sides."""
        raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        n = raw.nbytes // self.in_bytes
        depth = self._lib.iqgpu_chain_pipeline_depth()
        cap = self.max_out_frames(batch_frames) * self.out_bytes
        slots = [(PinnedBuffer(batch_frames * self.in_bytes), PinnedBuffer(cap)) for _ in range(depth)]
        outs, flight = [], []

        def drain_one():
            t, got, ob = flight.pop(0)
            self.collect(t)
            outs.append(ob.array[:got * self.out_bytes].copy())

        pos = i = 0
        while pos < n:
            f = min(batch_frames, n - pos)
            if len(flight) == depth:
                drain_one()
            ib, ob = slots[i % depth]
            ib.array[:f * self.in_bytes] = raw[pos * self.in_bytes:(pos + f) * self.in_bytes]
            got, t = self.submit(ib.ptr, f, ob.ptr, cap)
            flight.append((t, got, ob))
            pos += f
            i += 1
        while flight:
            drain_one()
        out = np.concatenate(outs) if outs else np.empty(0, np.uint8)
        return out.view(_NP_VIEW[self.desc.out_format]).copy()

    def reset(self):
        check(self._lib.iqgpu_chain_reset(self._h))

    def agc_state(self):
        """dict of the AGC fields the reference keeps in AppResources (synchronises)"""
        st = AgcState()
        check(self._lib.iqgpu_chain_get_agc_state(self._h, C.byref(st)))
        return dict(locked=bool(st.locked), peak_memory=st.peak_memory, gain=st.current_gain,
                    last_strong_peak_time=st.last_strong_peak_time, samples_seen=st.samples_seen)

    def set_iq_factors(self, mag, phase):
        check(self._lib.iqgpu_chain_set_iq_factors(self._h, mag, phase))

    def enable_iq_probe(self, on=True):
        check(self._lib.iqgpu_chain_enable_iq_probe(self._h, int(on)))

    def read_iq_probe(self):
        """the block the reference hands its optimiser (src/pipeline.c:468-476): first 1024 pre-processed
        samples of the most recent call of >= 1024 frames, or None"""
        blk = np.empty(1024, np.complex64)
        valid = C.c_int(0)
        check(self._lib.iqgpu_chain_read_iq_probe(self._h, blk.ctypes.data_as(C.c_void_p), C.byref(valid)))
        return blk if valid.value else None

    # ---- plumbing ----
    def set_stream(self, hip_stream):
        check(self._lib.iqgpu_chain_set_stream(self._h, C.c_void_p(hip_stream)))

    def get_stream(self):
        return self._lib.iqgpu_chain_get_stream(self._h)

    def synchronize(self):
        check(self._lib.iqgpu_chain_synchronize(self._h))

    def set_profiling(self, on=True):
        check(self._lib.iqgpu_chain_set_profiling(self._h, int(on)))

    def front_kernel(self):
        """name of the front kernel the last process call launched (iqgpu_chain_front_kernel)"""
        return self._lib.iqgpu_chain_front_kernel(self._h).decode()

    def profile(self):
        p = Profile()
        check(self._lib.iqgpu_chain_get_profile(self._h, C.byref(p)))
        return {name: dict(launches=int(p.launches[i]), ms=float(p.ms[i]))
                for i, name in enumerate(_lib.K_NAMES)}


class PinnedBuffer:
    """hipHostMalloc'd (page-locked) host buffer with a numpy view, for submit() / collect()."""

    def __init__(self, nbytes):
        self._lib = _lib.load()
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(self._lib.iqgpu_host_malloc_pinned(self.nbytes, C.byref(p)))
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr))[:self.nbytes]

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._lib.iqgpu_host_free_pinned(C.c_void_p(self.ptr))
            self.ptr = None

    __del__ = free


class DeviceBuffer:
    """hipMalloc'd buffer through the C ABI (for hosts without their own HIP binding)."""

    def __init__(self, nbytes, device=0):
        self._lib = _lib.load()
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(self._lib.iqgpu_device_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(self._lib.iqgpu_memcpy_h2d(self.device, C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, nbytes=None, dtype=np.uint8):
        nbytes = self.nbytes if nbytes is None else int(nbytes)
        out = np.empty(nbytes, np.uint8)
        check(self._lib.iqgpu_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), nbytes))
        return out.view(dtype)

    def free(self):
        if getattr(self, "ptr", None):
            self._lib.iqgpu_device_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    __del__ = free
