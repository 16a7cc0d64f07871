#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/.

  sample_convert.npz   inputs + outputs of the REFERENCE's own src/sample_convert.c, compiled where
                       it lies under /root/reference into oracle/_ref/ (oracle/Makefile).  These pin
                       orc_convert_* / the GPU pack + unpack bit-for-bit.
  nrsc5_65536.npz      65 536 synthetic cs16 frames (seed 1) through the NRSC-5 chain of the ORACLE
                       (the reference's liquid-dsp half cannot be built here): a regression pin of the
                       restatement, not a reference output.
  design.npz           half-band / polyphase / NCO / filter design constants of the oracle.

Run here (needs /root/reference for the first file):  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from iq_tool_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

FORMATS = ["cs8", "cu8", "cs16", "cu16", "sc16q11", "cs24", "cs32", "cu32", "cf32"]


def convert_vectors():
    assert po.ref_lib() is not None, "oracle/_ref/libsampleconvert_ref.so missing: run `make -C oracle` with /root/reference present"
    rng = np.random.default_rng(20251121)
    out = {}
    n = 1500
    for f in FORMATS:
        fid = po.FMT[f]
        raw = rng.integers(0, 256, n * po.BYTES[fid], dtype=np.uint8)
        if f == "cf32":
            raw = (rng.standard_normal(2 * n).astype(np.float32) * np.float32(0.7)).view(np.uint8)
        # extreme codes first
        if f in ("cs8", "cu8"):
            raw[:8] = [0, 255, 128, 127, 1, 254, 129, 126]
        if f in ("cs16", "cu16", "sc16q11"):
            raw[:16] = np.array([0, 32767, -32768, -1, 1, 16384, -16384, 255], np.int16).view(np.uint8)
        out["unpack_in_" + f] = raw
        for g, tag in ((1.0, "g1"), (0.37, "g037"), (-2.5, "gm25")):
            out["unpack_out_%s_%s" % (f, tag)] = po.ref_to_cf32(raw, f, g).view(np.float32)
    # pack: random, rounding boundaries, clamps (finite, inside the range where the reference is defined)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * np.float32(0.6)
    for f in FORMATS:
        scale = {"cs8": 127, "cu8": 127, "cs16": 32767, "cu16": 32767, "sc16q11": 2048, "cs24": 8388607,
                 "cs32": 2147483647, "cu32": 2147483647, "cf32": 1}[f]
        halves = (np.arange(-40, 40, dtype=np.float32) + np.float32(0.5)) / np.float32(scale)
        edge = np.array([0, 1, -1, 0.5, -0.5, 1.5, -1.5, 2.0, -2.0, 1e-8, -1e-8, 0.999999, -0.999999], np.float32)
        ext = np.concatenate([edge, halves, np.nextafter(halves, np.float32(1)), np.nextafter(halves, np.float32(-1))])
        xi = x.copy()
        xi[:ext.size] = ext + 1j * ext[::-1]
        out["pack_in_" + f] = xi.view(np.float32)
        out["pack_out_" + f] = po.ref_from_cf32(xi, f)
    out["bytes_per_sample"] = np.array([[fid, po.ref_lib().get_bytes_per_sample(fid)] for fid in range(0, 18)], np.int64)
    np.savez_compressed(os.path.join(HERE, "sample_convert.npz"), **out)


def chain_vectors():
    raw = synth.raw_stream(65536, 2.4e6, 1, "cs16")
    kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
    out_i, out_c = po.Chain(**kw).process(raw, want_cf32=True)
    np.savez_compressed(os.path.join(HERE, "nrsc5_65536.npz"), raw=raw, out_cs16=out_i, out_cf32=out_c.view(np.float32))


def design_vectors():
    r = np.float32(744187.5 / 2.4e6)
    m = po.MsResamp(r)
    nco = po.Nco(np.float32(2 * np.pi * 200e3 / 2.4e6))
    f3 = po.Filter(po.make_filter_cfg((("passband", 158.5e3, 113e3),), filter_taps=1025), 10e6, 2.4e6)
    np.savez_compressed(os.path.join(HERE, "design.npz"),
                        ratio=np.float32(r), S=m.S, step=np.uint32(m.step), rate_arb=np.float32(m.rate_arb),
                        hb_m=np.array([m.stage_m(k) for k in range(m.S)]), hb_taps0=m.stage_taps(0), arb_proto=m.arb_proto(),
                        nco_dtheta=np.uint32(nco.dtheta_u32), nco_table=nco.table(),
                        cfg3_taps=f3.taps().view(np.float32), cfg3_block=f3.block, cfg3_impl=f3.impl)


if __name__ == "__main__":
    po.build()
    convert_vectors()
    chain_vectors()
    design_vectors()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
