"""SURVEY.md 8(f1) + 8(f2): real bitstreams.  tests/golden/stream_*.bin are HEVC streams: the reference's enc265 on seeded
synthetic YUV (all-intra: the only structure its encoder emits) and, as stream_f2_*, the output of the synthetic
bitstream WRITER oracle/f2_writer.cc (I/P/B pictures with reference picture sets, weighted prediction, PCM, cu_qp_delta,
AMP, several slices per picture, deblocking overrides, SAO merge ...: tools/make_stream_golden.py lists the settings).
tests/golden/stream_*.npz hold, per picture in DECODE order, what the
hooks of the RECORDING reference decoder (oracle/f1_recorder.patch + oracle/f1_recorder.cc, built by `make -C oracle f1`)
collected at slice.cc:3424 / motion.cc:279 / slice.cc:4185 / decctx.cc:757 -- a de265hip_picture_desc -- and the MD5 of
libde265's own decoded picture before and after its post-filters.

CPU:  the restatement replays the recorded descs to libde265's pictures; the product's host helper derives the same
      edge flags as libde265 did; in the build container the recording decoder still reproduces the fixtures.
GPU:  the descs go through the product's INCREMENTAL recorder API (de265hip_record_* -> de265hip_recorder_submit), as
      the hooks of an integrated libde265 would drive it, and the HIP path's picture equals libde265's: config 1
      (720p 8-bit all-intra) end to end from a real bitstream."""
import glob
import hashlib
import os
import subprocess
import tempfile

import numpy as np
import pytest

import f1_stream
import pyoracle
import pyref
from libde265_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "stream_*.npz")))
IDS = [os.path.basename(f)[7:-4] for f in FIXTURES]
F1_DEC = os.path.join(ROOT, "oracle", "_ref", "f1_dec")
F2_WRITER = os.path.join(ROOT, "oracle", "_ref", "f2_writer")


def warnings_of(stderr):
    """what the decoder printed, minus libde265's notice that a stream without WPP / tiles is decoded by one thread"""
    return [l for l in stderr.splitlines() if l.strip() and "Cannot run decoder multi-threaded" not in l]


def md5(planes):
    m = hashlib.md5()
    for p in planes:
        m.update(np.ascontiguousarray(p).tobytes())
    return m.hexdigest()


def test_fixtures_present():
    assert len(FIXTURES) >= 8 and "720p_intra_q30" in IDS and "f2_b_10bit_wp_ctb64" in IDS


def test_f2_fixtures_hold_inter_pictures_with_real_reference_lists():
    """the f2 streams exist to bring P/B pictures through the real parser: PUs, both lists, weights, several slices"""
    pics = f1_stream.load_fixture(os.path.join(ROOT, "tests", "golden", "stream_f2_b_10bit_wp_ctb64.npz"))
    assert [rp.meta["poc"] for rp, _ in pics] == [0, 4, 2, 1, 3]                 # decode order of a hierarchical GOP
    assert all(rp.meta["n_pus"] > 0 for rp, _ in pics[1:]) and all(rp.meta["n_slices"] == 2 for rp, _ in pics)
    assert {rp.to_desc().slices[0].slice_type for rp, _ in pics} >= {0, 2}       # B and I slices


@pytest.mark.parametrize("fx", FIXTURES, ids=IDS)
def test_oracle_replays_recorded_stream_pictures(fx):
    dpb = {}                                                     # libde265's own DPB index -> decoded picture
    for i, (rp, dg) in enumerate(f1_stream.load_fixture(fx)):
        P = rp.params
        d = rp.to_desc()
        for stage, key in ((_abi.STAGE_PREFILTER, "prefilter"), (_abi.STAGE_FINAL, "final")):
            out = pyoracle.alloc_planes(P.width, P.height, P.bit_depth_luma, chroma_format=P.chroma_format_idc)
            pyoracle.reconstruct(d, None, dpb, out, stage)
            assert md5(out) == dg[key], "%s picture %d: %s differs from libde265's decoder" % (os.path.basename(fx), i, key)
        dpb[rp.meta["dst_slot"]] = out


@pytest.mark.parametrize("fx", FIXTURES, ids=IDS)
def test_host_helper_derives_the_edge_flags_libde265_derived(fx):
    from libde265_amd import backend
    for rp, _ in f1_stream.load_fixture(fx):
        cb_log2, cb_part, tu_split, noedge = rp.structure()
        d = rp.to_desc()
        got = noedge.copy()
        backend.derive_edge_flags(d.params, d.slices, d.n_slices, d.ctbs, cb_log2, cb_part, tu_split, got)
        assert np.array_equal(got, noedge | rp.a["edges"])


@pytest.mark.skipif(not os.path.exists(F1_DEC), reason="recording decoder (make -C oracle f1) not built here")
@pytest.mark.parametrize("threads", [0, 4], ids=["sequential", "4_worker_threads"])
@pytest.mark.parametrize("fx", FIXTURES, ids=IDS)
def test_recording_decoder_still_reproduces_the_fixture(fx, threads):
    """threads: libde265's own worker threads (WPP rows / tiles parse concurrently, decctx.cc:976-1178): the hooks then fire
    on several threads and the recorder's per-thread buffers must merge to exactly the sequential decode order"""
    bits = fx[:-4] + ".bin"
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([F1_DEC, bits], env=dict(os.environ, F1_OUT=td, F1_THREADS=str(threads)), capture_output=True, text=True)
        assert r.returncode == 0 and not warnings_of(r.stderr), r.stderr[-2000:]   # no warnings, SEI picture hashes (f2 streams) verified
        dumps = sorted(os.listdir(td))
        fixture = f1_stream.load_fixture(fx)
        assert len(dumps) == len(fixture)
        for fn, (rp, dg) in zip(dumps, fixture):
            rp2, pre, fin = f1_stream.load_dump(os.path.join(td, fn))
            assert md5(pre) == dg["prefilter"] and md5(fin) == dg["final"]
            assert rp2.meta == rp.meta and all(np.array_equal(rp2.a[k], rp.a[k]) for k in f1_stream.SECTIONS)


@pytest.mark.gpu
@pytest.mark.parametrize("fx", FIXTURES, ids=IDS)
def test_gpu_replays_recorded_stream_pictures_through_the_recorder_api(fx):
    from libde265_amd import backend
    assert backend.device_count() > 0
    dec = backend.Decoder()
    try:
        for i, (rp, dg) in enumerate(f1_stream.load_fixture(fx)):
            P = rp.params
            d = rp.to_desc()
            sf = rp.a["scaling"] if rp.a["scaling"].size else None
            rec = backend.Recorder(P, sf)
            rec.record_desc(d)                                   # record_slice / _ctb / _tu / _pu / _pcm / _blk_planes one by one
            slot = rp.meta["dst_slot"]                           # reference lists name libde265's DPB indices: keep them
            assert slot < _abi.MAX_DPB_SLOTS
            dec.dpb_alloc(slot, P.width, P.height, P.bit_depth_luma, chroma_format=P.chroma_format_idc)
            pic = rec.submit(dec, slot)
            try:
                for stage, key in ((_abi.STAGE_PREFILTER, "prefilter"), (_abi.STAGE_FINAL, "final")):
                    dec.upload(slot, pyoracle.alloc_planes(P.width, P.height, P.bit_depth_luma, chroma_format=P.chroma_format_idc))
                    dec.run(pic, stage)
                    dec.sync()
                    got = dec.download(slot, P.width, P.height, P.bit_depth_luma)
                    assert md5(got) == dg[key], "%s picture %d: %s differs from libde265's decoder" % (os.path.basename(fx), i, key)
            finally:
                pic.free()
                rec.free()
    finally:
        dec.close()


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(F1_DEC), reason="patched reference decoder (make -C oracle f1) did not travel")
@pytest.mark.parametrize("mode", ["sync", "pipelined"])
@pytest.mark.parametrize("fx", FIXTURES, ids=IDS)
def test_libde265_decodes_the_bitstream_with_the_hip_back_end(fx, mode):
    """SURVEY 8(f1) end to end: the PATCHED libde265 (oracle/f1_recorder.patch: 9 lines) parses the real bitstream on the
    host - NAL, SPS/PPS, slice headers, CABAC - and every decode_TU / generate_inter_prediction_samples / post-filter call
    is replaced by the MI355X back end through de265hip_record_* -> recorder_submit -> picture_run -> dpb_download
    (F1_MODE=hip, oracle/f1_recorder.cc).  What de265_get_next_picture then hands out must be byte-identical to what
    the unpatched CPU path decodes (the fixtures' MD5s): `dec265 --accel hip` in everything but the option parser.
    mode "pipelined" (SURVEY 8(f3)): four libde265 worker threads parse, pictures are only enqueued on the device by a submit
    pool of three worker threads, their copy-out into pinned picture memory is waited for at output time (F1_PIPELINE=3 F1_THREADS=4)."""
    from libde265_amd import backend
    assert backend.device_count() > 0
    bits = fx[:-4] + ".bin"
    fixture = f1_stream.load_fixture(fx)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "out.yuv")
        env = dict(os.environ, F1_MODE="hip", F1_HIP_LIB=backend.SO_PATH)
        if mode == "pipelined":
            env.update(F1_PIPELINE="3", F1_THREADS="4", F1_CHECK_HASH="0")      # the hash check would wait for every picture at once
        r = subprocess.run([F1_DEC, bits, out], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and not warnings_of(r.stderr), r.stderr[-2000:]   # stderr: libde265's own SEI MD5 check (f2 streams carry the hash)
        assert r.stdout.split()[0] == str(len(fixture)), r.stdout
        data = open(out, "rb").read()
    off = 0
    fixture = sorted(fixture, key=lambda e: e[0].meta["poc"])     # fixtures are in decode order, the decoder outputs by POC
    for i, (rp, dg) in enumerate(fixture):
        P = rp.params
        bpp = 2 if P.bit_depth_luma > 8 else 1
        cw, ch = P.width // (2 if P.chroma_format_idc in (1, 2) else 1), P.height // (2 if P.chroma_format_idc == 1 else 1)
        n = (P.width * P.height + 2 * cw * ch) * bpp
        m = hashlib.md5(data[off:off + n]).hexdigest()
        off += n
        assert m == dg["final"], "%s picture %d: the HIP-backed decoder's output differs from the CPU decoder's" % (os.path.basename(fx), i)
    assert off == len(data)


# SURVEY 8(d) picture formats at FULL size, from real bitstreams written on the spot (too large to commit as fixtures):
# config 2's format (1080p 8-bit, inter) and the north-star format (3840x2160 10-bit, inter)
FULL_SIZE = [
    ("1080p8_B_wp_2slices", "gop=B pics=5 w=1920 h=1080 log2ctb=6 slices=2 wp=1 seed=11"),
    ("1080p8_LDB_ctb16_3refs", "gop=LDB pics=4 w=1920 h=1080 log2ctb=4 log2maxtb=4 nref=3 slices=5 sdh=1 tskip=1 seed=12"),
    ("1080p8_P_tiles_wpp_off", "gop=P pics=3 w=1920 h=1080 log2ctb=5 tile_cols=5 tile_rows=3 tile_uniform=0 slices=4 seed=14"),
    ("1080p8_B_scaling_lists", "gop=B pics=3 w=1920 h=1080 log2ctb=5 scaling=2 seed=16"),
    ("4k10_B", "gop=B pics=4 w=3840 h=2160 bits=10 log2ctb=6 seed=13"),
    ("4k10_B_wpp", "gop=B pics=3 w=3840 h=2160 bits=10 log2ctb=6 wpp=1 slices=2 seed=15"),
    # range-extension streams at full size
    ("1080p8_LDB_444_xcc_rdpcm_rot", "gop=LDB pics=3 w=1920 h=1080 chroma=3 xcc=1 irdpcm=1 erdpcm=1 rot=1 tskip=1 tskip_log2=5 tqbypass=1 seed=17"),
    ("1080p10_B_422_wp_hpo", "gop=B pics=3 w=1920 h=1080 chroma=2 bits=10 log2ctb=6 wp=1 hpo=1 irdpcm=1 erdpcm=1 rot=1 tskip=1 tskip_log2=4 nosmooth=1 slices=2 seed=18"),
]


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(F1_DEC) and os.path.exists(F2_WRITER)), reason="f1_dec / f2_writer (make -C oracle f1 f2) did not travel")
@pytest.mark.parametrize("name,args", FULL_SIZE, ids=[n for n, _ in FULL_SIZE])
def test_full_size_synthetic_streams_decode_identically_with_the_hip_back_end(name, args):
    """SURVEY 8(f1)+8(f2) at BASELINE.json's picture sizes: f2_writer writes an inter stream, the reference decodes it on
    the CPU (its scalar path) and again with every reconstruction call offloaded to the MI355X (F1_MODE=hip); the two
    output files must be byte-identical."""
    from libde265_amd import backend
    assert backend.device_count() > 0
    with tempfile.TemporaryDirectory() as td:
        bits, cpu, hip = (os.path.join(td, f) for f in ("s.bin", "cpu.yuv", "hip.yuv"))
        subprocess.check_call([F2_WRITER, "out=" + bits] + args.split())
        env = {k: v for k, v in os.environ.items() if k not in ("F1_OUT", "F1_MODE")}
        r = subprocess.run([F1_DEC, bits, cpu], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and not r.stderr.strip(), r.stderr[-2000:]
        n = int(r.stdout.split()[0])
        assert n == int(dict(a.split("=") for a in args.split())["pics"])
        a = open(cpu, "rb").read()
        for mode in ({}, dict(F1_PIPELINE="3", F1_THREADS="4", F1_CHECK_HASH="0")):       # synchronous; pipelined with worker threads
            r = subprocess.run([F1_DEC, bits, hip], env=dict(env, F1_MODE="hip", F1_HIP_LIB=backend.SO_PATH, **mode), capture_output=True, text=True, timeout=120)
            assert r.returncode == 0 and not warnings_of(r.stderr), r.stderr[-2000:]
            assert int(r.stdout.split()[0]) == n
            b = open(hip, "rb").read()
            assert len(a) == len(b) and len(a) > 0
            if a != b:
                A, B = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
                first = int(np.nonzero(A != B)[0][0])
                raise AssertionError("%s %s: HIP-backed decode differs from the CPU decode: %d bytes, first at offset %d of %d (picture %d)"
                                     % (name, mode, int((A != B).sum()), first, len(a), first // (len(a) // n)))


# ---------------------------------------------------------------- f1 robustness (VERDICT r3 "What's missing" 1)
# (8-bit streams: for wider samples the reference's synthesised picture is half undefined - image.cc:510-523 fill_image is a
#  byte memset over stride x height BYTES -, so two runs of the reference itself need not agree behind the loss; the 10-bit
#  case below checks that the decode goes through and that the pictures ahead of the loss are the same)
LOSSY = [
    ("lost_p_picture", "gop=P pics=7 nref=2 w=416 h=240 seed=5", 2),
    ("lost_b_reference", "gop=B pics=9 w=416 h=240 seed=6 wp=1", 1),
    ("lost_ldb_picture_tiles", "gop=LDB pics=6 nref=3 w=832 h=480 tile_cols=2 tile_rows=2 seed=7", 3),
    ("lost_b_reference_10bit", "gop=B pics=9 bits=10 w=416 h=240 seed=6", 1),
]


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(F1_DEC) and os.path.exists(F2_WRITER)), reason="f1_dec / f2_writer (make -C oracle f1 f2) did not travel")
@pytest.mark.parametrize("name,args,lost", LOSSY, ids=[n for n, _, _ in LOSSY])
def test_a_stream_with_a_lost_picture_decodes_as_on_the_cpu(name, args, lost):
    """A picture's slices never arrive: libde265 synthesises the missing reference (generate_unavailable_reference_picture,
    decctx.cc:1408-1434: mid-grey planes) and decodes on.  With the HIP back end the recorder mirrors that picture into the
    device-resident DPB (de265hip_dpb_fill); the output must be byte-identical to the CPU decode of the same damaged stream."""
    import drop_picture
    from libde265_amd import backend
    assert backend.device_count() > 0
    with tempfile.TemporaryDirectory() as td:
        whole, bits, cpu, hip = (os.path.join(td, f) for f in ("w.bin", "s.bin", "cpu.yuv", "hip.yuv"))
        subprocess.check_call([F2_WRITER, "out=" + whole] + args.split())
        data, n_all = drop_picture.drop_picture(open(whole, "rb").read(), lost)
        open(bits, "wb").write(data)
        env = {k: v for k, v in os.environ.items() if k not in ("F1_OUT", "F1_MODE")}
        env["F1_CHECK_HASH"] = "0"                       # (the hashes of the pictures behind the loss cannot match: not an error of the decoder)
        r = subprocess.run([F1_DEC, bits, cpu], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        n = int(r.stdout.split()[0])
        assert 0 < n < n_all
        a = open(cpu, "rb").read()
        for mode in ({}, dict(F1_PIPELINE="3", F1_THREADS="4")):
            r = subprocess.run([F1_DEC, bits, hip], env=dict(env, F1_MODE="hip", F1_HIP_LIB=backend.SO_PATH, **mode), capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr[-2000:]
            assert int(r.stdout.split()[0]) == n
            b = open(hip, "rb").read()
            assert len(a) == len(b) and len(a) > 0
            a_, b_ = a, b
            if "bits=10" in args:                        # the first picture in output order is decoded ahead of the loss
                a_, b_ = a[:len(a) // n], b[:len(b) // n]
            if a_ != b_:
                A, B = np.frombuffer(a_, np.uint8), np.frombuffer(b_, np.uint8)
                first = int(np.nonzero(A != B)[0][0])
                raise AssertionError("%s %s: HIP-backed decode differs from the CPU decode: %d bytes, first at offset %d of %d (picture %d)"
                                     % (name, mode, int((A != B).sum()), first, len(a), first // (len(a) // n)))


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(F1_DEC) and os.path.exists(F2_WRITER)), reason="f1_dec / f2_writer (make -C oracle f1 f2) did not travel")
@pytest.mark.parametrize("mode", [{}, dict(F1_PIPELINE="2", F1_THREADS="2")], ids=["synchronous", "pipelined"])
def test_a_back_end_failure_becomes_the_result_of_de265_decode(mode):
    """The one failure the device side admits (a dependency wait of k_run that expires; injected with F1_FAULT): the patched
    libde265 must hand it to the application as the result of de265_decode (de265.h:82-139) - f1_dec prints it, frees the
    decoder and exits with 4 - instead of ending the process inside the library."""
    from libde265_amd import backend
    assert backend.device_count() > 0
    with tempfile.TemporaryDirectory() as td:
        bits, hip = os.path.join(td, "s.bin"), os.path.join(td, "hip.yuv")
        subprocess.check_call([F2_WRITER, "out=" + bits] + "gop=I pics=4 w=832 h=480 seed=9".split())
        env = {k: v for k, v in os.environ.items() if k not in ("F1_OUT", "F1_MODE")}
        r = subprocess.run([F1_DEC, bits, hip], env=dict(env, F1_MODE="hip", F1_HIP_LIB=backend.SO_PATH, F1_FAULT="2000", F1_CHECK_HASH="0", **mode),
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 4, (r.returncode, r.stderr[-2000:])
        assert "decode error" in r.stderr and "pictures" in r.stdout            # (reported, and the clean-up ran)
        # ... and the same decoder binary still works without the fault
        r = subprocess.run([F1_DEC, bits, hip], env=dict(env, F1_MODE="hip", F1_HIP_LIB=backend.SO_PATH, **mode), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and not warnings_of(r.stderr), r.stderr[-2000:]
