"""ctypes loader for libiqgpu.so.  Fails loudly when the HIP library is missing: there is no
CPU fallback anywhere in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IQGPU_LIB") or os.path.join(HERE, "lib", "libiqgpu.so")

FMT = dict(cu8=8, cs8=9, cu16=10, cs16=11, cs24=12, cu32=13, cs32=14, cf32=15, sc16q11=16)
FMT_NAME = {v: k for k, v in FMT.items()}
BYTES_PER_FRAME = {8: 2, 9: 2, 10: 4, 11: 4, 16: 4, 12: 6, 13: 8, 14: 8, 15: 8}
FILTER = dict(none=0, lowpass=1, highpass=2, passband=3, stopband=4)
FILTER_IMPL = dict(auto=0, fir=1, fft=2)
K_NAMES = ("dc_prefix", "dc_scan", "front", "filter", "move", "agc", "cascade")

ERRORS = {0: "OK", -1: "EINVAL", -2: "ENODEV", -3: "ENOMEM", -4: "ERATIO", -5: "EFORMAT", -6: "ESHIFT",
          -7: "EFILTER", -8: "ECAPACITY", -9: "EHIP", -10: "EUNSUPPORTED"}


class FilterReq(C.Structure):
    _fields_ = [("type", C.c_int), ("f1_hz", C.c_float), ("f2_hz", C.c_float)]


class ChainDesc(C.Structure):
    _fields_ = [("in_format", C.c_int), ("out_format", C.c_int),
                ("input_rate_hz", C.c_double), ("target_rate_hz", C.c_double),
                ("resample_ratio", C.c_float), ("gain", C.c_float),
                ("shift_hz", C.c_double), ("shift_after_resample", C.c_int),
                ("dc_block_enable", C.c_int),
                ("iq_correct_enable", C.c_int), ("iq_mag", C.c_float), ("iq_phase", C.c_float),
                ("no_resample", C.c_int),
                ("n_filters", C.c_int), ("filters", FilterReq * 5),
                ("transition_width_hz", C.c_float), ("attenuation_db", C.c_float),
                ("filter_taps", C.c_int), ("filter_impl", C.c_int), ("fft_size", C.c_int),
                ("device_ordinal", C.c_int), ("block_samples", C.c_size_t),
                ("agc_enable", C.c_int), ("agc_profile", C.c_int), ("agc_target", C.c_float),
                ("agc_clock", C.c_int), ("agc_chunk_frames", C.c_uint32)]


class AgcState(C.Structure):
    _fields_ = [("locked", C.c_int), ("peak_memory", C.c_float), ("current_gain", C.c_float),
                ("reserved", C.c_int), ("last_strong_peak_time", C.c_double), ("samples_seen", C.c_uint64)]


class ChainInfo(C.Structure):
    _fields_ = [("ratio", C.c_float), ("interp", C.c_int), ("num_halfband_stages", C.c_int),
                ("stage_m", C.c_int * 16), ("rate_arb", C.c_float), ("arb_step", C.c_uint32),
                ("nco_dtheta", C.c_uint32), ("dc_alpha", C.c_float),
                ("filter_post_resample", C.c_int), ("filter_impl", C.c_int),
                ("filter_ntaps", C.c_uint32), ("filter_block", C.c_uint32),
                ("history_samples", C.c_uint32)]


class Profile(C.Structure):
    _fields_ = [("launches", C.c_uint64 * 8), ("ms", C.c_double * 8)]


class IqOptimizerStats(C.Structure):
    _fields_ = [("calls", C.c_uint64), ("runs", C.c_uint64), ("skipped_interval", C.c_uint64),
                ("skipped_power", C.c_uint64), ("accepted", C.c_uint64),
                ("initial_metric", C.c_float), ("final_metric", C.c_float),
                ("average_power_db", C.c_float), ("power_range_db", C.c_float)]


RAND_DIR_FN = C.CFUNCTYPE(C.c_float, C.c_void_p)


class WavInfo(C.Structure):
    _fields_ = [("source_software", C.c_int),
                ("software_name", C.c_char * 64), ("software_version", C.c_char * 64), ("radio_model", C.c_char * 128),
                ("software_name_present", C.c_int), ("software_version_present", C.c_int), ("radio_model_present", C.c_int),
                ("center_freq_hz", C.c_double), ("center_freq_hz_present", C.c_int),
                ("timestamp_unix", C.c_int64), ("timestamp_unix_present", C.c_int),
                ("timestamp_str", C.c_char * 64), ("timestamp_str_present", C.c_int),
                ("sdr_info_present", C.c_int),
                ("sample_rate", C.c_int32), ("channels", C.c_int32), ("bits_per_sample", C.c_int32), ("format_tag", C.c_int32),
                ("in_format", C.c_int),
                ("data_offset", C.c_uint64), ("data_bytes", C.c_uint64), ("frames", C.c_uint64)]


class IqgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libiqgpu %s (%d): %s" % (ERRORS.get(code, "?"), code, msg))
        self.code = code


# every symbol include/iqgpu.h declares: (name, restype, argtypes)
_vp, _sz = C.c_void_p, C.c_size_t
SYMBOLS = [
    ("iqgpu_abi_version", C.c_int, []),
    ("iqgpu_last_error", C.c_char_p, []),
    ("iqgpu_device_count", C.c_int, []),
    ("iqgpu_device_pci_bus_id", C.c_int, [C.c_int, C.c_char_p, _sz]),
    ("iqgpu_device_numa_node", C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_char_p, _sz]),
    ("iqgpu_bind_thread_to_device", C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    ("iqgpu_chain_desc_init", None, [C.POINTER(ChainDesc)]),
    ("iqgpu_chain_create", C.c_int, [C.POINTER(ChainDesc), C.POINTER(_vp)]),
    ("iqgpu_chain_destroy", None, [_vp]),
    ("iqgpu_chain_get_info", C.c_int, [_vp, C.POINTER(ChainInfo)]),
    ("iqgpu_chain_get_filter_taps", C.c_int, [_vp, _vp, _sz]),
    ("iqgpu_design_probe", C.c_int, [C.POINTER(ChainDesc), C.POINTER(ChainInfo), _vp, _sz, _vp, _sz, _vp, _sz]),
    ("iqgpu_design_out_frames", C.c_int, [C.POINTER(ChainDesc), _sz, C.POINTER(_sz)]),
    ("iqgpu_chain_process", C.c_int, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    ("iqgpu_chain_process_device", C.c_int, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    ("iqgpu_chain_submit", C.c_int, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz), C.POINTER(C.c_uint64)]),
    ("iqgpu_chain_collect", C.c_int, [_vp, C.c_uint64]),
    ("iqgpu_chain_pipeline_depth", C.c_int, []),
    ("iqgpu_chain_reset", C.c_int, [_vp]),
    ("iqgpu_chain_get_agc_state", C.c_int, [_vp, C.POINTER(AgcState)]),
    ("iqgpu_chain_set_iq_factors", C.c_int, [_vp, C.c_float, C.c_float]),
    ("iqgpu_chain_max_out_frames", _sz, [_vp, _sz]),
    ("iqgpu_chain_next_out_frames", _sz, [_vp, _sz]),
    ("iqgpu_iq_optimizer_create", C.c_int, [C.POINTER(_vp)]),
    ("iqgpu_iq_optimizer_destroy", None, [_vp]),
    ("iqgpu_iq_optimizer_seed", C.c_int, [_vp, C.c_uint32]),
    ("iqgpu_iq_optimizer_set_rng", C.c_int, [_vp, RAND_DIR_FN, _vp]),
    ("iqgpu_iq_optimizer_set_factors", C.c_int, [_vp, C.c_float, C.c_float]),
    ("iqgpu_iq_optimizer_get_factors", C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("iqgpu_iq_optimizer_get_stats", C.c_int, [_vp, C.POINTER(IqOptimizerStats)]),
    ("iqgpu_iq_optimizer_metric", C.c_float, [_vp, _vp, C.c_float, C.c_float]),
    ("iqgpu_iq_optimizer_run", C.c_int, [_vp, _vp, C.c_double, C.POINTER(C.c_int)]),
    ("iqgpu_iq_optimizer_touch", C.c_int, [_vp, C.c_double]),
    ("iqgpu_chain_enable_iq_probe", C.c_int, [_vp, C.c_int]),
    ("iqgpu_chain_read_iq_probe", C.c_int, [_vp, _vp, C.POINTER(C.c_int)]),
    ("iqgpu_iq_optimizer_service", C.c_int, [_vp, _vp, C.c_double, C.POINTER(C.c_int)]),
    ("iqgpu_wav_info_init", None, [C.POINTER(WavInfo)]),
    ("iqgpu_wav_parse_auxi", C.c_int, [_vp, _sz, C.POINTER(WavInfo)]),
    ("iqgpu_wav_parse_filename", C.c_int, [C.c_char_p, C.POINTER(WavInfo)]),
    ("iqgpu_wav_probe", C.c_int, [C.c_char_p, C.POINTER(WavInfo)]),
    ("iqgpu_wav_shift_hz", C.c_int, [C.POINTER(WavInfo), C.c_float, C.c_float, C.POINTER(C.c_double)]),
    ("iqgpu_chain_set_stream", C.c_int, [_vp, _vp]),
    ("iqgpu_chain_get_stream", _vp, [_vp]),
    ("iqgpu_chain_synchronize", C.c_int, [_vp]),
    ("iqgpu_chain_set_profiling", C.c_int, [_vp, C.c_int]),
    ("iqgpu_chain_get_profile", C.c_int, [_vp, C.POINTER(Profile)]),
    ("iqgpu_chain_front_kernel", C.c_char_p, [_vp]),
    ("iqgpu_debug_set", C.c_int, [C.c_char_p, C.c_char_p]),
    ("iqgpu_debug_list", C.c_int, [C.c_char_p, _sz]),
    ("iqgpu_chain_debug_read_scratch", C.c_int, [_vp, _vp]),
    ("iqgpu_get_bytes_per_sample", _sz, [C.c_int]),
    ("iqgpu_convert_block_to_cf32", C.c_int, [_vp, _vp, _sz, C.c_int, C.c_float, C.c_int]),
    ("iqgpu_convert_cf32_to_block", C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int]),
    ("iqgpu_device_malloc", C.c_int, [C.c_int, _sz, C.POINTER(_vp)]),
    ("iqgpu_device_free", C.c_int, [C.c_int, _vp]),
    ("iqgpu_host_malloc_pinned", C.c_int, [_sz, C.POINTER(_vp)]),
    ("iqgpu_host_free_pinned", C.c_int, [_vp]),
    ("iqgpu_memcpy_h2d", C.c_int, [C.c_int, _vp, _vp, _sz]),
    ("iqgpu_memcpy_d2h", C.c_int, [C.c_int, _vp, _vp, _sz]),
    ("iqgpu_memcpy_h2d_async", C.c_int, [_vp, _vp, _sz, _vp]),
    ("iqgpu_memcpy_d2h_async", C.c_int, [_vp, _vp, _sz, _vp]),
    ("iqgpu_stream_create", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("iqgpu_stream_destroy", C.c_int, [_vp]),
    ("iqgpu_stream_synchronize", C.c_int, [_vp]),
    ("iqgpu_event_create", C.c_int, [C.POINTER(_vp)]),
    ("iqgpu_event_destroy", C.c_int, [_vp]),
    ("iqgpu_event_record", C.c_int, [_vp, _vp]),
    ("iqgpu_stream_wait_event", C.c_int, [_vp, _vp]),
    ("iqgpu_event_elapsed_ms", C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
]

_lib = None


def load():
    """Returns the loaded library; raises if libiqgpu.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libiqgpu.so is missing at %s -- run `python -m iq_tool_amd.build` (needs hipcc). "
                "iq_tool_amd has no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


# The library reads no switch from the environment (ABI v6: iqgpu_debug_set).  The tests, bench.py and the A/B scripts under tools/
# select kernels with IQGPU_<NAME>=... variables of THEIR OWN process; this mirror forwards them -- explicitly, right before a
# chain is designed or created -- so that `IQGPU_NO_FAST=1 python bench.py` keeps working while a host that links libiqgpu sees
# no environment dependence at all.
DEBUG_NAMES = ("force_generic", "no_fast", "agc_nofuse", "no_raw0", "no_kt", "fft_no_r16", "no_fat", "force_fat", "fat", "mid8",
               "no_s2", "no_fused_move", "no_p0", "no_casc2", "no_mid_8bit", "fuse_filter", "tap_fold", "steal", "steal_min",
               "steal_rounds", "steal_stride", "steal_lanes", "run_weights", "cus", "fft_log2n", "fft_threads", "fft_geometry", "casc2_min_run",
               "sysfs_root")


def apply_debug_env():
    """sets the library's diagnostic switches to what this process's IQGPU_<NAME> variables say (all others cleared)"""
    lib = load()
    check(lib.iqgpu_debug_set(None, None))
    for name in DEBUG_NAMES:
        v = os.environ.get("IQGPU_" + name.upper())
        if v:
            check(lib.iqgpu_debug_set(name.encode(), v.encode()))


def debug_switches():
    """what iqgpu_debug_list reports: {name: value} of the switches that are set"""
    buf = C.create_string_buffer(4096)
    check(load().iqgpu_debug_list(buf, len(buf)))
    return dict(kv.split("=", 1) for kv in buf.value.decode().split(";") if kv)


def check(rc):
    if rc != 0:
        raise IqgpuError(rc, load().iqgpu_last_error().decode("utf-8", "replace"))
