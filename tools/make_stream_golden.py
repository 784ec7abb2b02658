#!/usr/bin/env python3
"""SURVEY.md 8(f1) fixtures from REAL BITSTREAMS (build container only).  For each stream below: seeded synthetic YUV ->
the reference's own encoder CLI (oracle/_ref/enc265, all-intra: the only structure it can emit) ->
the RECORDING reference decoder (oracle/_ref/f1_dec = libde265 + oracle/f1_recorder.patch + oracle/f1_recorder.cc),
which dumps per picture the de265hip_picture_desc the product consumes and libde265's own decoded picture.  Written:

  tests/golden/stream_<name>.bin     the bitstream (an output of the reference's tools on synthetic input)
  tests/golden/stream_<name>.npz     the recorded descs + MD5 of libde265's picture before / after its post-filters

    make -C oracle f1 f2 && python tools/make_stream_golden.py
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import f1_stream  # noqa: E402

REFDIR = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")

STREAMS = [
    # SURVEY 8d config 1 workload: 1280x720 8-bit all-intra, q 30, seeded sin*cos luma (2 of its 8 frames keep the fixture small)
    dict(name="720p_intra_q30", w=1280, h=720, frames=2, seed=7, noise=10, enc=["-q", "30"]),
    dict(name="wqvga_intra_q20_ctb64", w=416, h=240, frames=4, seed=11, noise=18, enc=["-q", "20", "--max-cb-size", "64"]),
    dict(name="cif_intra_q35_nxn", w=352, h=288, frames=3, seed=13, noise=30,
         enc=["-q", "35", "--CB-IntraPartMode", "fixed", "--CB-IntraPartMode-Fixed-partMode", "NxN", "--max-cb-size", "16", "--max-tb-size", "16"]),
    dict(name="qcif_intra_q12_tb8", w=176, h=144, frames=3, seed=17, noise=40, enc=["-q", "12", "--min-tb-size", "8", "--min-cb-size", "16"]),
]


# streams of the synthetic bitstream WRITER (oracle/f2_writer.cc, `make -C oracle f2`): the inter, weighted,
# PCM, cu_qp_delta, AMP, multi-slice, WPP, tiles, deblocking-override and SAO-merge syntax the reference's own encoder cannot
# emit.  Every picture carries a decoded-picture-hash SEI (MD5), which libde265 checks while it decodes (sei.cc:273).
F2_STREAMS = [
    dict(name="f2_p_ctb32", args="gop=P pics=4 w=192 h=128 seed=1"),
    dict(name="f2_b_10bit_wp_ctb64", args="gop=B pics=5 w=256 h=144 log2ctb=6 bits=10 wp=1 slices=2 seed=2"),
    dict(name="f2_ldb_slices_ctb16", args="gop=LDB pics=4 w=176 h=144 log2ctb=4 log2maxtb=4 slices=4 lists_mod=1 sdh=1 tskip=1 "
                                          "tqbypass=1 cip=1 nref=3 seed=3"),
    dict(name="f2_i_10bit_pcm7", args="gop=I pics=2 w=200 h=136 log2ctb=6 bits=10 pcm_bits=7 pcm_lf_off=1 seed=4"),
    dict(name="f2_b_wpp_slices", args="gop=B pics=5 w=256 h=192 wpp=1 slices=3 seed=5"),
    dict(name="f2_ldb_scaling_lists_10bit", args="gop=LDB pics=3 w=192 h=128 scaling=2 bits=10 tskip=1 seed=7"),
    dict(name="f2_p_wpp_dependent_segments", args="gop=P pics=3 w=256 h=192 dep=60 wpp=1 slices=2 seed=8"),
    dict(name="f2_p_tiles_4x3", args="gop=P pics=3 w=256 h=192 log2ctb=4 log2maxtb=4 tile_cols=4 tile_rows=3 tile_uniform=0 lf_tiles=0 slices=5 seed=6"),
    # range-extension streams (4:2:2 / 4:4:4 transform trees and chroma intra modes, cross-component prediction,
    # implicit + explicit RDPCM, transform-skip rotation and large transform-skip blocks, intra smoothing off, high-precision
    # weighted-prediction offsets)
    dict(name="f4_ldb_444_xcc_rdpcm_rot", args="gop=LDB pics=4 w=192 h=128 chroma=3 xcc=1 irdpcm=1 erdpcm=1 rot=1 tskip=1 tskip_log2=5 "
                                               "tqbypass=1 seed=21"),
    dict(name="f4_b_422_10bit_wp_hpo", args="gop=B pics=5 w=192 h=128 chroma=2 bits=10 wp=1 hpo=1 irdpcm=1 erdpcm=1 rot=1 tskip=1 "
                                            "tskip_log2=4 tqbypass=1 nosmooth=1 slices=2 seed=22"),
    dict(name="f4_i_444_10bit_pcm_ctb64", args="gop=I pics=2 w=200 h=136 chroma=3 log2ctb=6 bits=10 pcm_bits=7 seed=23"),
]


def synth_yuv(w, h, n, seed, noise):
    """luma sin(x/23)*cos(y/31)*A + 128 + noise, seeded; chroma smooth + noise."""
    rng = np.random.default_rng(seed)
    x = np.arange(w)[None, :]
    y = np.arange(h)[:, None]
    tex = rng.uniform(-1, 1, (h, w))
    out = bytearray()
    for f in range(n):
        Y = np.sin((x + 3 * f) / 23.0) * np.cos((y + 2 * f) / 31.0) * 90 + 128 + np.roll(tex, (f, 2 * f), (0, 1)) * noise
        U = np.sin(x[:, ::2] / 17.0 + f) * 40 + 128 + rng.uniform(-noise / 3, noise / 3, (h // 2, w // 2))
        V = np.cos(y[::2] / 13.0 - f) * 40 + 128 + rng.uniform(-noise / 3, noise / 3, (h // 2, w // 2))
        for p in (Y, U, V):
            out += np.clip(np.rint(p), 0, 255).astype(np.uint8).tobytes()
    return bytes(out)


def md5(planes):
    m = hashlib.md5()
    for p in planes:
        m.update(np.ascontiguousarray(p).tobytes())
    return m.hexdigest()


def record(bitstream, outdir):
    env = dict(os.environ, F1_OUT=outdir)
    r = subprocess.run([os.path.join(REFDIR, "f1_dec"), bitstream], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and not r.stderr.strip(), r.stderr
    pics = []
    for fn in sorted(os.listdir(outdir)):
        rp, pre, fin = f1_stream.load_dump(os.path.join(outdir, fn))
        pics.append((rp, {"prefilter": md5(pre), "final": md5(fin)}))
    return pics


def main():
    only = set(sys.argv[1:])                                    # optional: regenerate just the named streams
    for st in STREAMS:
        if only and st["name"] not in only:
            continue
        with tempfile.TemporaryDirectory() as td:
            yuv, bits = os.path.join(td, "in.yuv"), os.path.join(GOLD, "stream_%s.bin" % st["name"])
            open(yuv, "wb").write(synth_yuv(st["w"], st["h"], st["frames"], st["seed"], st["noise"]))
            subprocess.check_call([os.path.join(REFDIR, "enc265"), "-i", yuv, "-w", str(st["w"]), "-h", str(st["h"]),
                                   "-f", str(st["frames"]), "--sop-structure", "intra", "-o", bits] + st["enc"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                  cwd=td)                   # (enc265 drops a recon.yuv into its working directory)
            dumps = os.path.join(td, "dumps")
            os.makedirs(dumps)
            pics = record(bits, dumps)
            assert len(pics) == st["frames"], (st["name"], len(pics))
            f1_stream.save_fixture(os.path.join(GOLD, "stream_%s.npz" % st["name"]), pics)
            print(st["name"], "bitstream %d B," % os.path.getsize(bits), len(pics), "pictures,",
                  sum(rp.meta["n_tus"] for rp, _ in pics), "TUs,", sum(rp.meta["n_coeffs"] for rp, _ in pics), "coefficients; fixture",
                  os.path.getsize(os.path.join(GOLD, "stream_%s.npz" % st["name"])), "B")
    for st in F2_STREAMS:
        if only and st["name"] not in only:
            continue
        with tempfile.TemporaryDirectory() as td:
            bits = os.path.join(GOLD, "stream_%s.bin" % st["name"])
            subprocess.check_call([os.path.join(REFDIR, "f2_writer"), "out=" + bits] + st["args"].split())
            os.remove(bits + ".chk")
            pics = record(bits, td)
            f1_stream.save_fixture(os.path.join(GOLD, "stream_%s.npz" % st["name"]), pics)
            print(st["name"], "bitstream %d B," % os.path.getsize(bits), len(pics), "pictures,", sum(rp.meta["n_pus"] for rp, _ in pics), "PUs,",
                  sum(rp.meta["n_tus"] for rp, _ in pics), "TUs,", sum(rp.meta["n_coeffs"] for rp, _ in pics), "coefficients; fixture",
                  os.path.getsize(os.path.join(GOLD, "stream_%s.npz" % st["name"])), "B")
    shutil.rmtree(os.path.join(GOLD, "__pycache__"), ignore_errors=True)


if __name__ == "__main__":
    main()
