"""Property / fuzz tests of the host side of libiqgpu (no device): the WAV metadata parsers -- code that reads sizes, offsets and
text out of files somebody else wrote (src/input_wav.c:146-438) -- and the create-time design path over random descriptors
(src/setup.c:91-122, src/filter.c:43-393).  VERDICT r5 item 4.

Two kinds of property: ROBUSTNESS (any input: the call returns, reports an error code it documents, and leaves the struct in a
state its own invariants describe -- run under ASan + UBSan by tools/sanitize_host.py this is where an over-read, an overflow or a
shift by 64 would surface) and DIFFERENTIAL (inputs both sides define: the same answer as oracle/wav_oracle.py -- expat itself,
as in the reference -- or as oracle/iq_oracle.c's design).

hypothesis runs derandomised (the same examples on every run: a CI failure reproduces) at IQGPU_FUZZ_EXAMPLES examples per
property (default 150, a few seconds each; `IQGPU_FUZZ_EXAMPLES=20000 python tools/sanitize_host.py test -k fuzz` for a soak)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, Phase, assume, example, given, settings     # noqa: E402
from hypothesis import strategies as st                                   # noqa: E402

N_EX = int(os.environ.get("IQGPU_FUZZ_EXAMPLES", "150"))
FUZZ = settings(max_examples=N_EX, deadline=None, derandomize=True, database=None,
                suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture, HealthCheck.data_too_large])

ERR_CODES = set(range(-10, 1))


@pytest.fixture(scope="module")
def lib():
    import iq_tool_amd
    return iq_tool_amd.load()


@pytest.fixture(scope="module")
def wm():
    from iq_tool_amd import wav_meta
    return wav_meta


@pytest.fixture(scope="module")
def wo():
    from oracle import wav_oracle
    return wav_oracle


def md_invariants(d):
    """what an iqgpu_wav_info must look like whatever was parsed: flags are 0 / 1, strings are terminated inside their arrays,
    a present centre frequency is finite"""
    for k in ("software_name", "software_version", "radio_model", "timestamp_str", "center_freq_hz", "timestamp_unix"):
        assert d[k + "_present"] in (0, 1), k
    assert len(d["software_name"]) <= 63 and len(d["software_version"]) <= 63 and len(d["radio_model"]) <= 127 and len(d["timestamp_str"]) <= 63
    assert 0 <= d["source_software"] <= 4
    if d["center_freq_hz_present"]:
        assert np.isfinite(d["center_freq_hz"])


def same_as_oracle(wm, md, ref):
    d = wm.as_dict(md)
    md_invariants(d)
    assert d["source_software"] == ref["source_software"]
    for k in ("software_name", "software_version", "radio_model", "timestamp_str"):
        assert bool(d[k + "_present"]) == (ref[k] is not None), (k, d[k], ref[k])
        if ref[k] is not None:
            assert d[k] == ref[k], k
    assert bool(d["center_freq_hz_present"]) == (ref["center_freq_hz"] is not None), (d["center_freq_hz"], ref["center_freq_hz"])
    if ref["center_freq_hz"] is not None:
        assert d["center_freq_hz"] == ref["center_freq_hz"]
    assert bool(d["timestamp_unix_present"]) == (ref["timestamp_unix"] is not None)
    if ref["timestamp_unix"] is not None:
        assert d["timestamp_unix"] == ref["timestamp_unix"]


# ------------------------------------------------------------------------------------------------------------
# auxi chunk
# ------------------------------------------------------------------------------------------------------------
@FUZZ
@given(st.binary(min_size=0, max_size=600))
def test_fuzz_auxi_any_bytes(wm, blob):
    """any bytes: the parser returns and leaves a well-formed struct (robustness; under the sanitizers: no over-read)"""
    ok, md = wm.parse_auxi(blob)
    md_invariants(wm.as_dict(md))
    assert ok in (True, False)


_ATTR_TEXT = st.text(alphabet=st.sampled_from("abcXYZ 0123456789.-+:_/#()eExX"), min_size=0, max_size=140)
_NUM_TEXT = st.one_of(
    st.integers(-10**12, 10**12).map(str),
    st.floats(allow_nan=False, allow_infinity=False, width=64).map(repr),
    st.sampled_from(["", " 97900000", "97900000 ", "9.79e7", "0x1p20", "inf", "nan", "-0", "1e400", "12abc", "+5", " \t7"]),
    _ATTR_TEXT)
_TIME_TEXT = st.one_of(
    st.tuples(st.integers(-3, 40), st.integers(-3, 15), st.integers(1600, 2500), st.integers(-2, 30), st.integers(-2, 70), st.integers(-2, 70))
    .map(lambda t: "%d-%d-%d %d:%d:%d" % t),
    _ATTR_TEXT)


@st.composite
def definition_xml(draw):
    """a well-formed SDR Console style document: <SDR-XML-Root><Definition attr=... /></SDR-XML-Root> with attributes drawn from the
    ones the reference looks at (src/input_wav.c:345-412) and a few it does not, values anything printable"""
    names = draw(st.lists(st.sampled_from(["SoftwareName", "SoftwareVersion", "RadioModel", "RadioCenterFreq", "UTCSeconds", "CurrentTimeUTC",
                                           "Other", "SampleRate"]), unique=True, max_size=8))
    parts = []
    for n in names:
        if n == "RadioCenterFreq" or n == "UTCSeconds":
            v = draw(_NUM_TEXT)
        elif n == "CurrentTimeUTC":
            v = draw(_TIME_TEXT)
        elif n == "SoftwareName":
            v = draw(st.one_of(st.sampled_from(["SDR Console", "SDR Console V3.2", "other"]), _ATTR_TEXT))
        else:
            v = draw(_ATTR_TEXT)
        v = v.replace("&", "&amp;").replace("<", "&lt;").replace('"', "&quot;")
        parts.append('%s="%s"' % (n, v))
    elem = draw(st.sampled_from(["Definition", "Definition", "Definition", "Other"]))
    pad = draw(st.sampled_from(["", "\n", "  ", "\r\n\t"]))
    head = draw(st.sampled_from(["", '<?xml version="1.0"?>', '<?xml version="1.0" encoding="UTF-8"?>\n']))
    doc = "%s<SDR-XML-Root>%s<%s %s/>%s</SDR-XML-Root>" % (head, pad, elem, " ".join(parts), pad)
    return doc.encode("utf-8")


@FUZZ
@given(definition_xml(), st.integers(0, 3))
def test_fuzz_auxi_wellformed_xml_equals_expat(wm, wo, doc, trailing_nuls):
    """well-formed documents (what a recorder writes; the chunk is often NUL-padded): field for field what the reference's expat
    handler takes from them -- strtod's full-string rule for the frequency, %d-%d-%d %d:%d:%d through timegm for the time, 63 / 127
    byte truncation of the strings"""
    blob = doc + b"\0" * trailing_nuls
    ref = wo.new_md()
    wo.parse_auxi(blob.rstrip(b"\0") if trailing_nuls else blob, ref)
    # (a timestamp far outside what time_t arithmetic of the oracle's _timegm and the library agree on is not a capture: skip)
    assume(ref["timestamp_unix"] is None or abs(ref["timestamp_unix"]) < 2**55)
    ok, md = wm.parse_auxi(blob)
    same_as_oracle(wm, md, ref)


@FUZZ
@given(definition_xml(), st.data())
def test_fuzz_auxi_mutated_xml_is_survived(wm, doc, data):
    """the same documents cut short, with bytes flipped, with a tag opened and never closed: whatever the hand-written walker makes of
    them, it returns with a well-formed struct (what survives a parse error is pinned by the truncated_xml fixture, not here)"""
    b = bytearray(doc)
    for _ in range(data.draw(st.integers(0, 4))):
        if not b:
            break
        i = data.draw(st.integers(0, len(b) - 1))
        op = data.draw(st.integers(0, 3))
        if op == 0:
            b = b[:i]
        elif op == 1:
            b[i] = data.draw(st.integers(0, 255))
        elif op == 2:
            b[i:i] = data.draw(st.sampled_from([b"<", b"<a ", b'"', b"&", b"<!--", b"<![CDATA[", b"\0", b"='"]))
        else:
            del b[i:i + data.draw(st.integers(1, 8))]
    ok, md = wm.parse_auxi(bytes(b))
    md_invariants(wm.as_dict(md))


@FUZZ
@given(st.binary(min_size=0, max_size=80))
def test_fuzz_auxi_binary_equals_oracle(wm, wo, blob):
    """SDRuno / SDRconnect binary chunks (SYSTEMTIME + centre frequency at byte 32, src/input_wav.c:282-333): any bytes that are not
    XML go this way on both sides"""
    assume(b"<" not in blob)
    ref = wo.new_md()
    want_ok = wo.parse_auxi(blob, ref)
    ok, md = wm.parse_auxi(blob)
    assert ok == want_ok
    same_as_oracle(wm, md, ref)


# ------------------------------------------------------------------------------------------------------------
# file names
# ------------------------------------------------------------------------------------------------------------
_NAME_PIECE = st.one_of(
    st.sampled_from(["SDRSharp", "SDRuno", "SDRconnect", "IQ", "_", "__", "Hz", "kHz", "hz", "HZ", ".wav", "Z", "-", " ", "0x1p3", "1e3", "nan", "inf"]),
    st.integers(0, 10**11).map(str),
    st.tuples(st.integers(0, 9999), st.integers(0, 99), st.integers(0, 99), st.integers(0, 99), st.integers(0, 99), st.integers(0, 99))
    .map(lambda t: "_%04d%02d%02d_%02d%02d%02dZ" % t),
    st.text(alphabet=st.sampled_from("abcdefXYZ0123456789_.-+"), min_size=0, max_size=12))


@FUZZ
@given(st.lists(_NAME_PIECE, min_size=0, max_size=10).map("".join))
def test_fuzz_filename_equals_oracle(wm, wo, name):
    """SDR#-style base names (src/input_wav.c:192-260): the first `hz` with an underscore in front of it and fewer than 32 characters
    between, strtod's full-string rule, the first _YYYYMMDD_HHMMSSZ, the recorder prefix"""
    assume("\0" not in name and "/" not in name)
    ref = wo.new_md()
    want_ok = wo.parse_filename(name, ref)
    assume(ref["timestamp_unix"] is None or abs(ref["timestamp_unix"]) < 2**55)
    ok, md = wm.parse_filename(name)
    assert ok == want_ok, name
    same_as_oracle(wm, md, ref)


@FUZZ
@given(st.binary(min_size=0, max_size=300))
def test_fuzz_filename_any_bytes(wm, raw):
    assume(b"\0" not in raw)
    lib = wm._lib.load()
    md = wm.WavInfo()
    lib.iqgpu_wav_info_init(C.byref(md))
    ok = lib.iqgpu_wav_parse_filename(raw, C.byref(md))
    assert ok in (0, 1)
    md_invariants(wm.as_dict(md))


# ------------------------------------------------------------------------------------------------------------
# whole files: RIFF / RF64 walks
# ------------------------------------------------------------------------------------------------------------
def _fmt_chunk(channels, rate, bits, tag=1, extensible=False, size_lie=None):
    body = struct.pack("<HHIIHH", 0xFFFE if extensible else tag, channels, rate & 0xFFFFFFFF, (rate * channels * bits // 8) & 0xFFFFFFFF,
                       (channels * bits // 8) & 0xFFFF, bits)
    if extensible:
        body += struct.pack("<HHI", 22, bits, 3) + struct.pack("<H", tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    return b"fmt " + struct.pack("<I", len(body) if size_lie is None else size_lie) + body


@st.composite
def wav_file(draw):
    """header + a chunk list in any order: fmt (sound or not), auxi (XML, binary or junk), data, ds64, LIST / unknown chunks, with the
    ways files go wrong -- odd sizes with and without their pad byte, sizes beyond the file, 0 and 0xFFFFFFFF data sizes, a ds64
    that lies, a cut anywhere"""
    rf64 = draw(st.booleans())
    chunks = []
    n_frames = draw(st.integers(0, 40))
    bits = draw(st.sampled_from([8, 16, 16, 16, 24, 32]))
    channels = draw(st.sampled_from([1, 2, 2, 2, 2, 3]))
    rate = draw(st.sampled_from([0, 1, 2400000, 2000000, 744187, 2**31, 2**32 - 1]))
    data_body = bytes(draw(st.binary(min_size=n_frames * channels * bits // 8, max_size=n_frames * channels * bits // 8)))
    kinds = draw(st.lists(st.sampled_from(["fmt", "auxi", "data", "LIST", "junk", "ds64", "fmt", "data"]), min_size=0, max_size=7))
    if draw(st.booleans()):
        kinds = ["fmt", "auxi", "data"] + kinds
    for k in kinds:
        if k == "fmt":
            chunks.append(_fmt_chunk(channels, rate, bits, tag=draw(st.sampled_from([1, 1, 1, 3, 0xFFFE])), extensible=draw(st.booleans()),
                                     size_lie=draw(st.sampled_from([None, None, None, 0, 15, 16, 17, 40, 2**31, 2**32 - 1]))))
        elif k == "auxi":
            body = draw(st.one_of(definition_xml(), st.binary(min_size=0, max_size=90)))
            lie = draw(st.sampled_from([None, None, None, 0, len(body) + 1, len(body) + 1000, 2**24, 2**32 - 1]))
            chunks.append(b"auxi" + struct.pack("<I", len(body) if lie is None else lie) + body + (b"\0" if len(body) & 1 and draw(st.booleans()) else b""))
        elif k == "data":
            lie = draw(st.sampled_from([None, None, None, 0, 0xFFFFFFFF, len(data_body) + 7, 2**31]))
            chunks.append(b"data" + struct.pack("<I", len(data_body) if lie is None else lie) + data_body + (b"\0" if len(data_body) & 1 and draw(st.booleans()) else b""))
        elif k == "ds64":
            chunks.append(b"ds64" + struct.pack("<I", draw(st.sampled_from([28, 24, 8, 0, 2**32 - 1]))) +
                          struct.pack("<QQQI", draw(st.integers(0, 2**64 - 1)), draw(st.sampled_from([0, len(data_body), 2**40, 2**64 - 1])), 0, 0))
        elif k == "LIST":
            inner = b"INFO" + b"ISFT" + struct.pack("<I", 4) + b"abc\0"
            chunks.append(b"LIST" + struct.pack("<I", draw(st.sampled_from([len(inner), len(inner) + 3, 2**32 - 2]))) + inner)
        else:
            body = draw(st.binary(min_size=0, max_size=20))
            chunks.append(draw(st.binary(min_size=4, max_size=4)) + struct.pack("<I", draw(st.sampled_from([len(body), len(body), 2**32 - 1, 1]))) + body)
    payload = b"".join(chunks)
    riff_size = draw(st.sampled_from([4 + len(payload), 8, 0, 2**32 - 1]))
    head = draw(st.sampled_from([b"RF64" if rf64 else b"RIFF"] * 6 + [b"RIFX", b"FORM"]))
    form = draw(st.sampled_from([b"WAVE"] * 6 + [b"AVI ", b"wave"]))
    blob = head + struct.pack("<I", riff_size) + form + payload
    cut = draw(st.one_of(st.none(), st.integers(0, len(blob))))
    return blob if cut is None else blob[:cut]


@FUZZ
@given(wav_file(), st.sampled_from(["capture.wav", "SDRSharp_20240131_123456Z_97900000Hz_IQ.wav", "SDRuno_x.wav", "a_Hz.wav", "_0Hz"]))
def test_fuzz_wav_probe_survives_any_file(wm, tmp_path_factory, blob, name):
    """whatever the file holds, iqgpu_wav_probe returns one of its documented codes; on IQGPU_OK the description is self-consistent:
    two channels, one of the two sample formats, a positive rate, the data range inside the file, whole frames"""
    lib = wm._lib.load()
    d = tmp_path_factory.mktemp("wavfuzz")
    path = os.path.join(str(d), name)
    with open(path, "wb") as fh:
        fh.write(blob)
    md = wm.WavInfo()
    rc = lib.iqgpu_wav_probe(os.fsencode(path), C.byref(md))
    assert rc in (0, -1, -5), rc                                       # OK, EINVAL (not a WAV / unreadable / no rate), EFORMAT
    if rc == 0:
        assert md.channels == 2 and md.in_format in (8, 11) and md.sample_rate > 0
        assert md.data_offset <= len(blob) and md.data_offset + md.data_bytes <= len(blob), (md.data_offset, md.data_bytes, len(blob))
        bpf = 4 if md.in_format == 11 else 2
        assert md.frames == md.data_bytes // bpf
        md_invariants(wm.as_dict(md))
    os.unlink(path)
    # and a path that does not exist, a directory: EINVAL, nothing else
    assert lib.iqgpu_wav_probe(os.fsencode(path), C.byref(md)) == -1
    assert lib.iqgpu_wav_probe(os.fsencode(str(d)), C.byref(md)) == -1


@FUZZ
@given(st.floats(allow_nan=True, allow_infinity=True, width=32), st.floats(allow_nan=True, allow_infinity=True, width=32),
       st.one_of(st.none(), st.floats(allow_nan=False, allow_infinity=False)))
def test_fuzz_wav_shift_rule(wm, wo, target, shift_arg, center):
    """the shift rule of wav_initialize (src/input_wav.c:614-629) over every float the two options can hold"""
    lib = wm._lib.load()
    md = wm.WavInfo()
    lib.iqgpu_wav_info_init(C.byref(md))
    ref = wo.new_md()
    if center is not None:
        md.center_freq_hz, md.center_freq_hz_present = center, 1
        ref["center_freq_hz"] = center
    out = C.c_double(123.0)
    rc = lib.iqgpu_wav_shift_hz(C.byref(md), target, shift_arg, C.byref(out))
    err, want = wo.shift_hz(ref, target, shift_arg)
    if np.isnan(target) or np.isnan(shift_arg):
        assert rc in (0, -6)                                           # (a NaN option never reaches this rule in the reference: config.c refuses it)
        return
    assert (rc != 0) == (err is not None), (rc, err)
    if rc == 0:
        assert out.value == want or (np.isnan(out.value) and np.isnan(want))


# ------------------------------------------------------------------------------------------------------------
# the create-time design path over random descriptors
# ------------------------------------------------------------------------------------------------------------
def _probe(lib, **kw):
    from iq_tool_amd import _lib
    from iq_tool_amd.chain import make_desc
    d = make_desc(**kw)
    info = _lib.ChainInfo()
    ft = np.zeros(2 * 70000, np.float32)
    hb = np.zeros(8192, np.float32)
    arb = np.zeros(3584, np.float32)
    rc = lib.iqgpu_design_probe(C.byref(d), C.byref(info), ft.ctypes.data_as(C.c_void_p), 70000,
                                hb.ctypes.data_as(C.c_void_p), 8192, arb.ctypes.data_as(C.c_void_p), 3584)
    return rc, d, info, ft.view(np.complex64), hb, arb


_RATE = st.one_of(st.sampled_from([2.4e6, 10e6, 61.44e6, 2.048e6, 744187.5, 1488375.0, 48e3, 8e3, 1.0, 250e3, 20e6]),
                  st.floats(min_value=1.0, max_value=1e9, allow_nan=False))


@FUZZ
@given(_RATE, _RATE, st.floats(min_value=-5e6, max_value=5e6, allow_nan=False), st.booleans(), st.booleans())
def test_fuzz_resampler_design_equals_oracle(lib, fin, fout, shift, dc, after):
    """random rates through iqgpu_design_probe: ERATIO exactly when the float ratio leaves [0.001, 1000] (src/setup.c:109-112); else the
    doubling rule, every stage length and tap, the arbitrary stage's rate, step and 3585-tap prototype, the NCO step and the DC
    blocker's alpha equal the oracle's, bit for bit"""
    from oracle import pyoracle
    pyoracle.build()
    kw = dict(input_rate_hz=fin, target_rate_hz=fout, shift_hz=shift, dc_block=dc, shift_after_resample=after)
    rc, d, info, _, hb, arb = _probe(lib, **kw)
    r = np.float32(fout / fin)
    assert rc in ERR_CODES
    if not (np.isfinite(r) and np.float32(0.001) <= r <= np.float32(1000.0)):
        assert rc == -4, (rc, r)
        return
    if rc != 0:
        assert rc == -6, (rc, lib.iqgpu_last_error())                   # ESHIFT: |shift| beyond what the rate carries
        return
    assert info.ratio == r
    m = pyoracle.MsResamp(r)
    assert info.interp == (1 if m.interp else 0)
    assert info.num_halfband_stages == m.S and info.arb_step == m.step and info.rate_arb == m.rate_arb
    o = 0
    for k in range(m.S):
        t = m.stage_taps(k)
        assert info.stage_m[k] == m.stage_m(k)
        assert np.array_equal(hb[o:o + t.size], t)
        o += t.size
    assert np.array_equal(arb, m.arb_proto())
    if dc:
        assert info.dc_alpha == np.float32(2.0 * np.pi * np.float32(10.0) / fin) or info.dc_alpha == np.float32(2.0 * 3.14159265358979323846 * 10.0 / fin)
    # and the shard writer's closed form on a few lengths: it answers, monotonically, within the capacity rule's bound
    prev = 0
    for n in (0, 1, 1000, 65537):
        got = C.c_size_t(0)
        assert lib.iqgpu_design_out_frames(C.byref(d), n, C.byref(got)) == 0
        # (an interpolating chain emits whole bursts of 2^S per sample of its arbitrary stage)
        burst = (1 << int(info.num_halfband_stages)) if info.interp else 1
        assert got.value >= prev and got.value <= int(np.ceil(n * float(r))) + 2 * burst + 2, (n, got.value)
        prev = got.value


# Frequencies are drawn as FRACTIONS of the rate the filter runs at (0.01 .. 0.7: beyond Nyquist included, for the fatal paths): the
# automatic transition width is a quarter of the cut-off (src/filter.c:182-195), so a 2 kHz low-pass at 10 MS/s is a filter of
# 90 000 taps and a chain of them a convolution of minutes -- both sides design it, in agreement, but not 150 times.  The degenerate
# requests (0 Hz, the 1 Hz transition floor) are the explicit examples at a 48 kHz rate.
_FRAC = st.floats(min_value=0.01, max_value=0.7, allow_nan=False)
_FILTER = st.one_of(
    st.tuples(st.sampled_from(["lowpass", "highpass"]), _FRAC, st.just(0.0)),
    st.tuples(st.sampled_from(["passband", "stopband"]), st.one_of(_FRAC, _FRAC.map(lambda x: -x), st.just(0.0)), _FRAC))


# (a failing descriptor is reported as found: shrinking re-designs filters of thousands of taps hundreds of times)
@settings(FUZZ, phases=[Phase.explicit, Phase.generate])
@example((48e3, 8e3), [("lowpass", 1.0, 0.0)], 0, 0.0, 0.0, "auto", 0, False)
@example((48e3, 8e3), [("passband", 0.0, 1.0)], 0, 1.0, 0.0, "auto", 0, False)
@example((48e3, 8e3), [("highpass", 24e3, 0.0)], 21, 0.0, 0.0, "fft", 1, True)
@example((2.4e6, 744187.5), [("stopband", 1e3, 1e9)], 97, 0.0, 0.0, "fir", 0, False)
@given(st.sampled_from([(2.4e6, 744187.5), (2.4e6, 1488375.0), (10e6, 2.4e6), (1.0e6, 2.5e6), (2.4e6, 2.4e6), (48e3, 8e3)]),
       st.lists(_FILTER, min_size=1, max_size=3), st.sampled_from([0, 0, 0, 21, 64, 97, 1024, 4097]),
       st.sampled_from([0.0, 0.0, 0.004, 0.05]), st.sampled_from([0.0, 0.0, 30.0, 60.0, 90.0, 200.0]),
       st.sampled_from(["auto", "auto", "fir", "fft"]), st.sampled_from([0, 0, 512, 2048, 1000, 1 << 20]), st.booleans())
def test_fuzz_filter_design_equals_oracle(lib, rates, reqs, ntaps, tw, att, impl, fft_size, nores):
    """random filter requests through iqgpu_design_probe against oracle/iq_oracle.c's design (src/filter.c:43-393: validation,
    placement, odd bump, automatic length, chain convolution, normalisation, implementation choice, block size): the same verdict
    -- accepted or EFILTER -- and, when accepted, the same placement, implementation, length, block and taps to the bit"""
    from oracle import pyoracle
    pyoracle.build()
    fin, fout = rates
    if max(abs(r[1]) for r in reqs) <= 0.7 and max(abs(r[2]) for r in reqs) <= 0.7:      # (the explicit examples are in Hz already)
        fs = float(np.float32(min(fin, fout) if not nores else fin))
        reqs = [(t, float(np.float32(f1 * fs)), float(np.float32(f2 * fs))) for t, f1, f2 in reqs]
        tw = float(np.float32(tw * fs))
    kw = dict(input_rate_hz=fin, target_rate_hz=fout, no_resample=nores, filters=tuple(reqs), filter_taps=ntaps, transition_width_hz=tw,
              attenuation_db=att, filter_impl=impl, fft_size=fft_size)
    rc, d, info, ft, _, _ = _probe(lib, **kw)
    assert rc in ERR_CODES
    odd = ntaps + 1 if ntaps and ntaps % 2 == 0 else ntaps
    cfg = pyoracle.make_filter_cfg(tuple(reqs), transition_width_hz=tw, attenuation_db=att, filter_taps=odd, impl=impl, fft_size=fft_size)
    try:
        f = pyoracle.Filter(cfg, fin, fin if nores else fout, no_resample=nores)
    except ValueError:
        assert rc == -7, (rc, lib.iqgpu_last_error(), kw)
        return
    assert rc == 0, (rc, lib.iqgpu_last_error(), kw)
    assert (bool(info.filter_post_resample), info.filter_impl, info.filter_ntaps, info.filter_block) == (f.post, f.impl, f.ntaps, f.block), kw
    n_cmp = min(f.ntaps, ft.size)               # (the probe copies what the caller's buffer holds: a 1 Hz transition asks for millions of taps)
    assert np.array_equal(ft[:n_cmp].view(np.float32), f.taps()[:n_cmp].view(np.float32)), kw
