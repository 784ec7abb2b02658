"""real bitstreams.  tests/golden/stream_*.bin are HEVC streams (the reference's enc265 on seeded
synthetic YUV, all-intra: the only structure its encoder emits); tests/golden/stream_*.npz hold, per picture, what the
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


def md5(planes):
    m = hashlib.md5()
    for p in planes:
        m.update(np.ascontiguousarray(p).tobytes())
    return m.hexdigest()


def test_fixtures_present():
    assert len(FIXTURES) >= 4 and "720p_intra_q30" in IDS


@pytest.mark.parametrize("fx", FIXTURES, ids=IDS)
def test_oracle_replays_recorded_stream_pictures(fx):
    for i, (rp, dg) in enumerate(f1_stream.load_fixture(fx)):
        P = rp.params
        d = rp.to_desc()
        for stage, key in ((_abi.STAGE_PREFILTER, "prefilter"), (_abi.STAGE_FINAL, "final")):
            out = pyoracle.alloc_planes(P.width, P.height, P.bit_depth_luma)
            pyoracle.reconstruct(d, None, {}, out, stage)
            assert md5(out) == dg[key], "%s picture %d: %s differs from libde265's decoder" % (os.path.basename(fx), i, key)


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
@pytest.mark.parametrize("fx", FIXTURES, ids=IDS)
def test_recording_decoder_still_reproduces_the_fixture(fx):
    bits = fx[:-4] + ".bin"
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call([F1_DEC, bits], env=dict(os.environ, F1_OUT=td), stdout=subprocess.DEVNULL)
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
            slot = rp.meta["dst_slot"] % _abi.MAX_DPB_SLOTS
            dec.dpb_alloc(slot, P.width, P.height, P.bit_depth_luma)
            pic = rec.submit(dec, slot)
            try:
                for stage, key in ((_abi.STAGE_PREFILTER, "prefilter"), (_abi.STAGE_FINAL, "final")):
                    dec.upload(slot, pyoracle.alloc_planes(P.width, P.height, P.bit_depth_luma))
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
@pytest.mark.parametrize("fx", FIXTURES, ids=IDS)
def test_libde265_decodes_the_bitstream_with_the_hip_back_end(fx):
    """SURVEY 8(f1) end to end: the PATCHED libde265 (oracle/f1_recorder.patch: 9 lines) parses the real bitstream on the
    host - NAL, SPS/PPS, slice headers, CABAC - and every decode_TU / generate_inter_prediction_samples / post-filter call
    is replaced by the MI355X back end through de265hip_record_* -> recorder_submit -> picture_run -> dpb_download
    (F1_MODE=hip, oracle/f1_recorder.cc).  What de265_get_next_picture then hands out must be byte-identical to what
    the unpatched CPU path decodes (the fixtures' MD5s): `dec265 --accel hip` in everything but the option parser."""
    from libde265_amd import backend
    assert backend.device_count() > 0
    bits = fx[:-4] + ".bin"
    fixture = f1_stream.load_fixture(fx)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "out.yuv")
        env = dict(os.environ, F1_MODE="hip", F1_HIP_LIB=backend.SO_PATH)
        r = subprocess.run([F1_DEC, bits, out], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.split()[0] == str(len(fixture)), r.stdout
        data = open(out, "rb").read()
    off = 0
    for i, (rp, dg) in enumerate(fixture):                 # all-intra streams: output order == decode order
        P = rp.params
        bpp = 2 if P.bit_depth_luma > 8 else 1
        n = (P.width * P.height + 2 * (P.width // 2) * (P.height // 2)) * bpp
        m = hashlib.md5(data[off:off + n]).hexdigest()
        off += n
        assert m == dg["final"], "%s picture %d: the HIP-backed decoder's output differs from the CPU decoder's" % (os.path.basename(fx), i)
    assert off == len(data)
