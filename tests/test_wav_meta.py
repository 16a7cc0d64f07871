"""WAV capture metadata -> frequency shift: iq_tool_amd/csrc/wav_meta.cpp against the restatement of
src/input_wav.c:146-438, 592-629 in oracle/wav_oracle.py (XML through expat itself, as in the reference) on the
committed fixtures of tests/golden/wav (written by tests/golden/gen_wav_fixtures.py).  Host-only code."""
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WAV = os.path.join(HERE, "golden", "wav")


@pytest.fixture(scope="module")
def wm():
    import iq_tool_amd
    iq_tool_amd.load()
    from iq_tool_amd import wav_meta
    return wav_meta


@pytest.fixture(scope="module")
def wo():
    from oracle import wav_oracle
    return wav_oracle


def chunks_of(path):
    b = open(path, "rb").read()
    pos, out = 12, {}
    ds64_data = None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack_from("<I", b, pos + 4)[0]
        if cid == b"ds64":
            ds64_data = struct.unpack_from("<Q", b, pos + 16)[0]
        if cid == b"data" and size == 0xFFFFFFFF:
            size = ds64_data
        out.setdefault(cid, b[pos + 8:pos + 8 + size])
        pos += 8 + size + (size & 1)
    return out


def same(md, ref):
    d = md if isinstance(md, dict) else None
    from iq_tool_amd import wav_meta
    d = wav_meta.as_dict(md)
    assert d["source_software"] == ref["source_software"]
    for k in ("software_name", "software_version", "radio_model", "timestamp_str"):
        assert bool(d[k + "_present"]) == (ref[k] is not None), k
        if ref[k] is not None:
            assert d[k] == ref[k], k
    assert bool(d["center_freq_hz_present"]) == (ref["center_freq_hz"] is not None)
    if ref["center_freq_hz"] is not None:
        assert d["center_freq_hz"] == ref["center_freq_hz"]
    assert bool(d["timestamp_unix_present"]) == (ref["timestamp_unix"] is not None)
    if ref["timestamp_unix"] is not None:
        assert d["timestamp_unix"] == ref["timestamp_unix"]


@pytest.mark.parametrize("name,fmt,rate,frames,center,software,unix", [
    ("SDRSharp_20240131_123456Z_97900000Hz_IQ.wav", 11, 2400000, 64, 97900000.0, 2, 1706704496),
    ("console_capture.wav", 11, 2400000, 64, 97900000.0, 1, 1706704496),
    ("SDRuno_20240131_123456Z_97900kHz.wav", 8, 2000000, 64, 97900000.0, 3, 1706704496),
    ("rf64_capture.wav", 11, 744187, 64, 97900000.0, 1, 1706704496),
    ("truncated_xml.wav", 11, 2400000, 64, 97900000.0, 1, 1706704496),
])
def test_probe_fixtures(wm, wo, name, fmt, rate, frames, center, software, unix):
    path = os.path.join(WAV, name)
    md = wm.probe(path)
    assert (md.in_format, md.sample_rate, md.channels, md.frames) == (fmt, rate, 2, frames)
    # the reference's order: auxi chunk first, then the base name fills what is still unset (input_wav.c:598-606)
    ref = wo.new_md()
    ch = chunks_of(path)
    if b"auxi" in ch:
        wo.parse_auxi(ch[b"auxi"], ref)
    wo.parse_filename(name, ref)
    same(md, ref)
    assert md.center_freq_hz == center and md.source_software == software and md.timestamp_unix == unix
    assert md.data_bytes == frames * (4 if fmt == 11 else 2)
    raw = open(path, "rb").read()[md.data_offset:md.data_offset + md.data_bytes]
    assert raw == ch[b"data"]


@pytest.mark.parametrize("name", ["killed_size0_97900000Hz.wav", "killed_sizeff_97900000Hz.wav", "killed_toolong_97900000Hz.wav"])
def test_unclosed_capture_runs_to_the_end_of_the_file(wm, name):
    """the data header of a capture that was never closed says 0, 0xFFFFFFFF or more than the file holds: libsndfile
    (sf_open, src/input_wav.c:556) reports the frames that are really there, and so must the probe"""
    path = os.path.join(WAV, name)
    md = wm.probe(path)
    assert md.frames == 64 and md.data_bytes == 256
    assert md.data_offset + md.data_bytes == os.path.getsize(path)
    assert md.center_freq_hz == 97900000.0


def test_empty_data_chunk_stays_empty(wm):
    """a data size of 0 is only "to the end of the file" under the 8-byte RIFF size of a file that was never finalised
    (libsndfile wav.c: chunk_size == 0 && RIFFsize == 8 && filelength > 44); under a finalised header it is an empty chunk,
    and the LIST chunk behind it is not sample data"""
    md = wm.probe(os.path.join(WAV, "empty_data_97900000Hz.wav"))
    assert md.frames == 0 and md.data_bytes == 0
    assert md.center_freq_hz == 97900000.0


def test_rejected_files(wm):
    import iq_tool_amd
    for name in ("mono.wav", "pcm24.wav", "float32_extensible.wav"):
        with pytest.raises(iq_tool_amd.IqgpuError) as e:
            wm.probe(os.path.join(WAV, name))
        assert "EFORMAT" in str(e.value), name
    with pytest.raises(iq_tool_amd.IqgpuError):
        wm.probe(os.path.join(WAV, "does_not_exist.wav"))
    with pytest.raises(iq_tool_amd.IqgpuError):
        wm.probe(__file__)                                   # not a RIFF file


def test_shift_rule(wm, wo):
    md = wm.probe(os.path.join(WAV, "console_capture.wav"))
    ref = wo.new_md()
    wo.parse_auxi(chunks_of(os.path.join(WAV, "console_capture.wav"))[b"auxi"], ref)
    # --wav-center-target-freq is a float option: 97.7 MHz is not representable, the difference is taken in double
    for tgt in (97.7e6, 97900000.0, 98.1e6, 1.0):
        err, want = wo.shift_hz(ref, tgt, 0.0)
        assert err is None
        assert wm.shift_hz(md, tgt, 0.0) == want
        assert want == 97900000.0 - float(np.float32(tgt))
    assert wm.shift_hz(md, 0.0, 12345.0) == 0.0             # option unused: --freq-shift applies (frequency_shift.c:27-31)
    import iq_tool_amd
    with pytest.raises(iq_tool_amd.IqgpuError):            # both options
        wm.shift_hz(md, 97.7e6, 1000.0)
    ok, bare = wm.parse_filename("plain_capture.wav")
    assert not ok
    with pytest.raises(iq_tool_amd.IqgpuError):            # no centre frequency in the file
        wm.shift_hz(bare, 97.7e6, 0.0)


NAMES = [
    "SDRSharp_20240131_123456Z_97900000Hz_IQ.wav", "SDRSharp_20150804_204253Z_101100kHz_IQ.wav",
    "SDRuno_20200907_184033Z_7140kHz.wav", "SDRconnect_IQ_20231224_060000_1000000HZ.wav",
    "gqrx_20240131_123456_97900000_2400000_fc.raw", "capture_1.5e6Hz.wav", "x_-5Hz.wav", "x_0Hz.wav", "x_12abHz.wav",
    "_Hz.wav", "Hz_100Hz.wav", "a_b_c_433920000hz_d.wav", "baseband_14070000Hz_13-45-12_24-02-2024.wav",
    "noise.wav", "rec_20241301_250000Z_1Hz.wav", "SDRuno_plain.wav", "A_20240229_235959Z.wav", "A_2024022_235959Z_5Hz.wav",
    "tone_1234567890123456789012345678901234Hz.wav", "x_ 7Hz.wav", "x_7 Hz.wav", "x_infHz.wav", "x_nanHz.wav", "x_0x10Hz.wav",
]


@pytest.mark.parametrize("name", NAMES)
def test_filename_rules(wm, wo, name):
    ok, md = wm.parse_filename(name)
    ref = wo.new_md()
    want = wo.parse_filename(name, ref)
    d = wm.as_dict(md)
    assert ok == want, (name, d, ref)
    same(md, ref)


def test_auxi_variants(wm, wo):
    base = chunks_of(os.path.join(WAV, "console_capture.wav"))[b"auxi"]
    cases = [
        base,
        base.replace(b'RadioCenterFreq="97900000"', b'RadioCenterFreq="97.9e6"'),
        base.replace(b'RadioCenterFreq="97900000"', b'RadioCenterFreq="97900000 Hz"'),      # strtod leaves a tail: ignored
        base.replace(b'UTCSeconds="1706704496"', b'UTCSeconds="12x"'),
        base.replace(b'CurrentTimeUTC="31-01-2024 12:34:56"', b'CurrentTimeUTC="bad"'),
        base.replace(b"SDR Console", b"SDR Console V3"),
        base.replace(b"SDR Console", b"HDSDR"),
        base.replace(b"<Definition", b"<definition"),                                          # element names are case sensitive
        b"<Definition RadioCenterFreq='1e6'/>",
        b"<a><Definition SoftwareName=\"x\" RadioModel=\"" + b"m" * 300 + b"\"/></a>",       # over-long strings are truncated
        b"<?xml version='1.0'?><r><!-- <Definition RadioCenterFreq='5'/> --><Definition UTCSeconds='77'/></r>",
        b"<r><Definition RadioCenterFreq='1e6'></r>",                                         # mismatched tag AFTER the element
        b"<r><Definition RadioCenterFreq='1e6' RadioCenterFreq='2e6'/></r>",                  # duplicate attribute: not well-formed
        b"not xml at all, shorter than 36",
        b"",
    ]
    st = struct.pack("<8H", 2023, 12, 0, 24, 6, 0, 1, 0)
    cases.append(st + st + struct.pack("<I", 7140000) + bytes(64))            # binary (SDRuno)
    cases.append(st + st + struct.pack("<I", 0) + bytes(64))                  # zero frequency: time only
    cases.append((st + st)[:30])                                              # too short
    for i, c in enumerate(cases):
        ref = wo.new_md()
        want = wo.parse_auxi(c, ref) if c else False
        ok, md = wm.parse_auxi(c)
        assert ok == want, (i, c[:60])
        same(md, ref)
