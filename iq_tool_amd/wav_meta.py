"""WAV capture metadata -> frequency shift: host-side mirror of the reference's WAV input module
(src/input_wav.c:146-438 parsers, 592-629 wav_initialize) over the C ABI of include/iqgpu.h."""
import ctypes as C
import os

from . import _lib
from ._lib import WavInfo, IqgpuError, check

SOFTWARE = {0: "Unknown", 1: "SDR Console", 2: "SDR#", 3: "SDRuno", 4: "SDRconnect"}


def _as_dict(md):
    d = {}
    for name, _t in WavInfo._fields_:
        v = getattr(md, name)
        d[name] = v.decode("utf-8", "replace") if isinstance(v, bytes) else v
    return d


def parse_auxi(chunk, md=None):
    """_parse_auxi_xml_expat, else _parse_binary_auxi_data (src/input_wav.c:175-181)"""
    lib = _lib.load()
    if md is None:
        md = WavInfo()
        lib.iqgpu_wav_info_init(C.byref(md))
    buf = bytes(chunk)
    ok = lib.iqgpu_wav_parse_auxi(buf, len(buf), C.byref(md))
    return bool(ok), md


def parse_filename(base, md=None):
    """parse_sdr_metadata_from_filename (src/input_wav.c:192-260)"""
    lib = _lib.load()
    if md is None:
        md = WavInfo()
        lib.iqgpu_wav_info_init(C.byref(md))
    ok = lib.iqgpu_wav_parse_filename(os.fsencode(base), C.byref(md))
    return bool(ok), md


def probe(path):
    """wav_initialize up to the metadata (src/input_wav.c:544-612)"""
    lib = _lib.load()
    md = WavInfo()
    rc = lib.iqgpu_wav_probe(os.fsencode(path), C.byref(md))
    if rc != 0:
        raise IqgpuError(rc, "not a 2-channel cs16 / cu8 WAV capture: %s" % path)
    return md


def shift_hz(md, center_target_hz=0.0, freq_shift_hz=0.0):
    """resources->nco_shift_hz (src/input_wav.c:614-629): 0.0 when --wav-center-target-freq is not used"""
    lib = _lib.load()
    out = C.c_double(0.0)
    rc = lib.iqgpu_wav_shift_hz(C.byref(md), center_target_hz, freq_shift_hz, C.byref(out))
    if rc != 0:
        raise IqgpuError(rc, "conflicting shift options, or no centre frequency in the file")
    return out.value


def as_dict(md):
    return _as_dict(md)
