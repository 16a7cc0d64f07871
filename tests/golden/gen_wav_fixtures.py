#!/usr/bin/env python3
"""Writes the tiny WAV captures under tests/golden/wav/ (data files, a few hundred bytes each) that the WAV
metadata tests read: SDR# file-name metadata, SDR Console `auxi` XML, SDRuno binary `auxi`, an RF64 header, and
files the reference's WAV input rejects.  Layouts follow what the reference parses (src/input_wav.c:146-438)."""
import os
import struct

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wav")


def chunk(cid, body):
    return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")


def fmt(channels, rate, bits, tag=1):
    ba = channels * bits // 8
    return chunk(b"fmt ", struct.pack("<HHIIHH", tag, channels, rate, rate * ba, ba, bits))


def wav(name, chunks, rf64=False, data_len=None, riff_size=None):
    body = b"WAVE" + b"".join(chunks)
    if rf64:
        head = b"RF64" + struct.pack("<I", 0xFFFFFFFF)
    else:
        head = b"RIFF" + struct.pack("<I", len(body) if riff_size is None else riff_size)
    with open(os.path.join(HERE, name), "wb") as fh:
        fh.write(head + body)


def main():
    os.makedirs(HERE, exist_ok=True)
    rng = np.random.default_rng(5)
    pcm16 = rng.integers(-2000, 2000, 2 * 64, dtype=np.int16).tobytes()
    pcm8 = rng.integers(100, 156, 2 * 64, dtype=np.uint8).tobytes()
    # 1. SDR#: everything is in the file name
    wav("SDRSharp_20240131_123456Z_97900000Hz_IQ.wav", [fmt(2, 2400000, 16), chunk(b"data", pcm16)])
    # 2. SDR Console: auxi XML in front of the data
    xml = ('<?xml version="1.0" encoding="UTF-8"?>\n<SDR-XML-Root xml:lang="EN" Description="Saved recording data" Created="31-Jan-2024 12:34">\n'
           '<Definition CurrentTimeUTC="31-01-2024 12:34:56" Filename="31-Jan-2024 123456.789 97.900MHz.wav" FirstFile="x" '
           'RadioModel="SDRplay RSP1A &amp; friends" RadioCenterFreq="97900000" SampleRate="2400000" SoftwareName="SDR Console" '
           'SoftwareVersion="Version 3.3 build 2947" UTCSeconds="1706704496" BitsPerSample=\'16\' />\n</SDR-XML-Root>\n').encode()
    wav("console_capture.wav", [fmt(2, 2400000, 16), chunk(b"auxi", xml), chunk(b"data", pcm16)])
    # 3. SDRuno: binary auxi (SYSTEMTIME start, SYSTEMTIME stop, centre frequency as uint32 at byte 32), cu8 samples
    st = struct.pack("<8H", 2024, 1, 3, 31, 12, 34, 56, 0)
    auxi = st + st + struct.pack("<I", 97900000) + bytes(128)
    wav("SDRuno_20240131_123456Z_97900kHz.wav", [fmt(2, 2000000, 8), chunk(b"auxi", auxi), chunk(b"data", pcm8)])
    # 4. RF64 header (what output_wav_rf64 writes): sizes in ds64, auxi behind the data, an odd-sized chunk in between
    ds64 = chunk(b"ds64", struct.pack("<QQQI", 0, len(pcm16), 64, 0))
    wav("rf64_capture.wav", [ds64, fmt(2, 744187, 16), chunk(b"LIST", b"odd"), b"data" + struct.pack("<I", 0xFFFFFFFF) + pcm16, chunk(b"auxi", xml)], rf64=True)
    # 5. XML that breaks off after its Definition element: what expat delivered before the error still counts
    wav("truncated_xml.wav", [fmt(2, 2400000, 16), chunk(b"auxi", xml[:xml.index(b"</SDR-XML-Root>")] + b"<broken"), chunk(b"data", pcm16)])
    # 6. rejected by the reference: one channel; 24-bit PCM; float
    wav("mono.wav", [fmt(1, 48000, 16), chunk(b"data", pcm16)])
    wav("pcm24.wav", [fmt(2, 2400000, 24), chunk(b"data", pcm16 + pcm16[:64])])
    wav("float32_extensible.wav", [chunk(b"fmt ", struct.pack("<HHIIHH", 0xFFFE, 2, 2400000, 2400000 * 8, 8, 32) + struct.pack("<HHIH", 22, 32, 3, 3) + bytes(14)), chunk(b"data", pcm16)])
    # 7. a recorder that was killed: the data header says 0 (under the 8-byte RIFF size of a file never finalised) / 0xFFFFFFFF /
    #    more than the file holds -- libsndfile reads to the end of the file
    for name, claimed, riff in (("killed_size0_97900000Hz.wav", 0, 8), ("killed_sizeff_97900000Hz.wav", 0xFFFFFFFF, None), ("killed_toolong_97900000Hz.wav", 4 * 64 + 4000, None)):
        wav(name, [fmt(2, 2400000, 16), b"data" + struct.pack("<I", claimed) + pcm16], riff_size=riff)
    # 8. a legitimately EMPTY data chunk under a finalised RIFF header, a LIST chunk behind it: zero frames, and the LIST bytes are not samples
    wav("empty_data_97900000Hz.wav", [fmt(2, 2400000, 16), chunk(b"data", b""), chunk(b"LIST", bytes(range(64)))])
    print(sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
