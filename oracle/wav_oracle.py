"""wav_oracle.py -- TEST INFRASTRUCTURE ONLY: restatement of the reference's WAV metadata rules
(/root/reference/src/input_wav.c) used to check iq_tool_amd/csrc/wav_meta.cpp.

Parity status: restated from the reference's own source (its input module needs libsndfile, absent here, so it
cannot be compiled: "parity unpinned" by reference vectors).  The XML side goes through the SAME parser library
the reference uses -- expat, via Python's pyexpat -- with the reference's start-element rules
(expat_start_element_handler, input_wav.c:345-412), including what survives a parse error (XML_Parse's result
is ignored, input_wav.c:421-428)."""
import math
import re
import struct
import xml.parsers.expat


def new_md():
    return dict(source_software=0, software_name=None, software_version=None, radio_model=None,
                center_freq_hz=None, timestamp_unix=None, timestamp_str=None)


def _timegm(year, month, day, hour, minute, sec):
    """timegm_portable (input_wav.c:262-280): mktime under TZ='' normalises out-of-range fields and has no
    four-digit year limit; the Gregorian calendar repeats every 400 years (146097 days)"""
    import datetime
    y, m = year + (month - 1) // 12, (month - 1) % 12 + 1
    cyc, yy = divmod(y - 2000, 400)
    days = (datetime.date(2000 + yy, m, 1).toordinal() - datetime.date(1970, 1, 1).toordinal()) + cyc * 146097 + (day - 1)
    return days * 86400 + hour * 3600 + minute * 60 + sec


def _strtod_full(s):
    """strtod that consumed the whole string (endptr at NUL) and gave a finite value: leading white space,
    decimal or hexadecimal floats; inf / nan parse but are not finite"""
    t = s.lstrip(" \t\n\v\f\r")
    if re.fullmatch(r"[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)", t):
        v = float(t)
    elif re.fullmatch(r"[+-]?0[xX]([0-9a-fA-F]+\.?[0-9a-fA-F]*|\.[0-9a-fA-F]+)([pP][+-]?\d+)?", t):
        v = float.fromhex(t)
    else:
        return None
    return v if math.isfinite(v) else None


def parse_auxi_xml(data, md):
    """_parse_auxi_xml_expat, input_wav.c:414-438"""
    def start(name, atts):
        if name != "Definition":
            return
        for k, v in atts.items():
            if k == "SoftwareName":
                md["software_name"] = v[:63]
            elif k == "SoftwareVersion":
                md["software_version"] = v[:63]
            elif k == "RadioModel":
                md["radio_model"] = v[:127]
            elif k == "RadioCenterFreq":
                d = _strtod_full(v)
                if d is not None:
                    md["center_freq_hz"] = d
            elif k == "UTCSeconds":
                if md["timestamp_unix"] is None and re.fullmatch(r"\s*[+-]?\d+", v):
                    md["timestamp_unix"] = int(v)
            elif k == "CurrentTimeUTC":
                md["timestamp_str"] = v[:63]
                m = re.match(r"\s*([+-]?\d+)-([+-]?\d+)-([+-]?\d+) ([+-]?\d+):([+-]?\d+):([+-]?\d+)", v)
                if m:
                    day, month, year, hh, mm, ss = (int(g) for g in m.groups())
                    md["timestamp_unix"] = _timegm(year, month, day, hh, mm, ss)
    p = xml.parsers.expat.ParserCreate()
    p.ordered_attributes = False
    p.StartElementHandler = start
    try:
        p.Parse(bytes(data), True)
    except xml.parsers.expat.ExpatError:
        pass                                            # the reference ignores XML_Parse's status
    any_data = any(md[k] is not None for k in ("software_name", "radio_model", "center_freq_hz", "timestamp_unix"))
    if any_data and md["software_name"] and "SDR Console" in md["software_name"]:
        md["source_software"] = 1
    return any_data


def parse_auxi_binary(data, md):
    """_parse_binary_auxi_data, input_wav.c:282-333"""
    if len(data) < 16 + 16 + 4:
        return False
    y, mo, _dow, d, h, mi, s, _ms = struct.unpack_from("<8H", data, 0)
    parsed = False
    if md["timestamp_unix"] is None:
        md["timestamp_unix"] = _timegm(y, mo, d, h, mi, s)
        parsed = True
        if md["timestamp_str"] is None:
            md["timestamp_str"] = "%04d-%02d-%02d %02d:%02d:%02d UTC" % (y, mo, d, h, mi, s)
    f = struct.unpack_from("<I", data, 32)[0]
    if f > 0 and md["center_freq_hz"] is None:
        md["center_freq_hz"] = float(f)
        parsed = True
    return parsed


def parse_auxi(data, md):
    """process_specific_chunk's order, input_wav.c:175-181"""
    if parse_auxi_xml(data, md):
        return True
    return parse_auxi_binary(data, md)


def parse_filename(base, md):
    """parse_sdr_metadata_from_filename, input_wav.c:192-260"""
    parsed = sharp = False
    if md["center_freq_hz"] is None:
        i = base.lower().find("hz")
        if i >= 0:
            us = base.rfind("_", 0, i)
            if us >= 0 and us + 1 < i and i - us - 1 < 32:
                v = _strtod_full(base[us + 1:i])
                if v is not None and v > 0:
                    md["center_freq_hz"] = v
                    parsed = sharp = True
    if md["timestamp_unix"] is None:
        for m in re.finditer(r"_", base):
            t = base[m.start():]
            # strlen >= 17, t[9] == '_', t[16] == 'Z', sscanf("_%4d%2d%2d_%2d%2d%2dZ") == 6
            mm = re.match(r"_(\d{4})(\d{2})(\d{2})_(\d{2})(\d{2})(\d{2})Z", t)
            if mm:
                y, mo, d, h, mi, s = (int(g) for g in mm.groups())
                md["timestamp_unix"] = _timegm(y, mo, d, h, mi, s)
                if md["timestamp_str"] is None:
                    md["timestamp_str"] = "%04d-%02d-%02d %02d:%02d:%02d UTC" % (y, mo, d, h, mi, s)
                parsed = sharp = True
                break
    if md["source_software"] == 0:
        if sharp:
            md["source_software"] = 2
        elif base.startswith("SDRuno_"):
            md["source_software"] = 3
        elif base.startswith("SDRconnect_"):
            md["source_software"] = 4
        if md["source_software"] != 0 and md["software_name"] is None:
            md["software_name"] = {1: "SDR Console", 2: "SDR#", 3: "SDRuno", 4: "SDRconnect"}[md["source_software"]]
            parsed = True
    return parsed


def shift_hz(md, center_target_hz, freq_shift_hz):
    """wav_initialize, input_wav.c:614-629; returns (error_or_None, nco_shift_hz)"""
    import numpy as np
    tgt = float(np.float32(center_target_hz))
    if tgt == 0.0:
        return None, 0.0
    if float(np.float32(freq_shift_hz)) != 0.0:
        return "conflict", 0.0
    if md["center_freq_hz"] is None:
        return "no-center", 0.0
    return None, md["center_freq_hz"] - tgt
