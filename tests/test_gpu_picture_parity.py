"""GPU parity tests proper: whole-picture reconstruction through the C ABI
(de265hip_picture_build/run) against the CPU oracle on the same seeded
synthetic command buffers.  Bit-exact at every stage."""
import os

import numpy as np
import pytest

import pyoracle
import pysynth
from libde265_amd import _abi, backend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    from libde265_amd import backend
    assert backend.device_count() > 0, "no GPU visible: HIP path cannot run"
    d = backend.Decoder()
    yield d
    d.close()


def run_case(dec, w, h, bd, slice_type, seed, stages=(0, 1, 2), drop_motion_plane=False, **over):
    cfg = pysynth.default_config(w, h, bd, slice_type, seed=seed, **over)
    sp = pysynth.SynthPicture(cfg)
    refs = {0: pysynth.fill_planes(w, h, bd, 100 + seed), 1: pysynth.fill_planes(w, h, bd, 200 + seed)}
    for slot, pl in refs.items():
        dec.dpb_alloc(slot, w, h, bd)
        dec.upload(slot, pl)
    if drop_motion_plane:                                # the device makes the plane from the PU records (blk_motion = NULL);
        import ctypes as C                               # the oracle keeps reading the generator's
        keep = C.cast(sp.d.blk_motion, C.c_void_p).value
        sp.d.blk_motion = None
    pic = dec.build(2, sp.desc)
    if drop_motion_plane:
        sp.d.blk_motion = C.cast(keep, C.POINTER(_abi.Motion))
    try:
        for stage in stages:
            # start both sides from the same garbage so untouched samples compare equal
            init = pysynth.fill_planes(w, h, bd, 999)
            exp = [p.copy() for p in init]
            pyoracle.reconstruct(sp.desc, sp.order, refs, exp, last_stage=stage)
            dec.upload(2, init)
            dec.run(pic, stage)
            dec.sync()
            got = dec.download(2, w, h, bd)
            for c in range(3):
                bad = np.argwhere(got[c] != exp[c])
                assert bad.size == 0, "stage %d comp %d: %d mismatches, first at (y,x)=%s got %d exp %d" % (
                    stage, c, len(bad), tuple(bad[0]), got[c][tuple(bad[0])], exp[c][tuple(bad[0])])
    finally:
        pic.free()
    return sp


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("slice_type", [2, 1, 0])
def test_small_picture(dec, bd, slice_type):
    run_case(dec, 416, 240, bd, slice_type, seed=3 + bd + slice_type)


@pytest.mark.parametrize("seed", range(4))
def test_feature_mix(dec, seed):
    run_case(dec, 352, 288, 8 if seed % 2 == 0 else 10, seed % 3, seed=40 + seed, tskip_pct=30, bypass_pct=8,
             pcm_pct=10, pcm_loop_filter_disable=seed & 1, scaling_list=(seed >> 1) & 1, n_slices=3,
             lf_across_slices_pct=50, weighted_pred=seed & 1, big_coeff_pct=3, constrained_intra_pred=seed == 3)


def test_tiles(dec):
    run_case(dec, 640, 384, 8, 0, seed=77, tile_cols=3, tile_rows=2, slice_per_tile=1, lf_across_tiles=0,
             lf_across_slices_pct=0)
    run_case(dec, 640, 384, 10, 2, seed=78, tile_cols=2, tile_rows=2, slice_per_tile=0, lf_across_tiles=0)


def test_ctb16_and_32(dec):
    run_case(dec, 208, 120, 8, 0, seed=5, log2_ctb_size=4, log2_max_tb_size=4)
    run_case(dec, 264, 136, 10, 0, seed=6, log2_ctb_size=5)


def test_strong_smoothing_decision_samples_are_dependencies(dec):
    """Regression (found by tools/exp/sweep.py 31 60, iteration 45): a 32x32 luma TU whose border is flat enough for the
    bilinear smoothing variant (here: weighted prediction with negative weights saturates whole regions to 0) reads
    p[+-32] for the decision (intrapred.cc:847-852) whatever its mode reads; the mode-aware dependency sets missed
    those two samples, so the TU could run before their producers."""
    run_case(dec, 1416, 536, 10, 1, seed=9045, stages=(0,), log2_ctb_size=5, log2_max_tb_size=5, log2_min_tb_size=3,
             intra_pct=40, tskip_pct=20, bypass_pct=0, pcm_pct=0, scaling_list=0, constrained_intra_pred=0,
             strong_intra_smoothing=1, weighted_pred=1, n_slices=4, split_bias=0, cbf_pct=60, mv_sigma_qpel=12)


@pytest.mark.parametrize("seed", range(6))
def test_motion_plane_made_on_the_device_from_the_pu_records(dec, seed):
    """de265hip_picture_desc::blk_motion = NULL: the boundary strengths of the deblocking read a motion plane the device derives
    from the PU records (k_motion_from_pus) - same picture as with the generator's plane, which the oracle reads.  P and B
    pictures, several slices (each with its own reference lists), AMP partitions, intra CUs in between, large vectors."""
    st = 1 if seed % 3 == 0 else 0
    run_case(dec, 416 + 64 * (seed % 2), 240, 8 if seed % 2 else 10, st, seed=600 + seed, stages=(1, 2), drop_motion_plane=True,
             n_slices=1 + seed % 4, amp=seed & 1, intra_pct=[0, 15, 50][seed % 3], cbf_pct=[0, 30, 100][seed % 3], bi_pct=[0, 60, 100][(seed + 1) % 3],
             mv_sigma_qpel=[2, 12, 60][seed % 3], weighted_pred=seed & 1, sao=0)


def test_1080p_b_picture(dec):
    run_case(dec, 1920, 1080, 8, 0, seed=11, stages=(2,))


# ---------------------------------------------------------------- edge cases
def test_empty_picture_and_errors(dec):
    import ctypes as C
    from libde265_amd import backend
    w, h, bd = 128, 64, 8
    sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 2, seed=1))
    d = sp.d
    n_tus, n_pus = d.n_tus, d.n_pus
    d.n_tus = 0                                     # nothing to reconstruct: picture passes through SAO/deblock only
    init = pysynth.fill_planes(w, h, bd, 5)
    exp = [p.copy() for p in init]
    pyoracle.reconstruct(sp.desc, None, {}, exp)
    dec.dpb_alloc(2, w, h, bd); dec.upload(2, init)
    pic = dec.build(2, sp.desc); dec.run(pic, 2); dec.sync()
    got = dec.download(2, w, h, bd)
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    assert pic.stats().n_tu_tasks == 0
    pic.free()
    d.n_tus = n_tus
    # unsupported / invalid parameters surface as de265_error codes, never as a fallback
    # what is still refused of the range extensions: monochrome pictures WITH prediction units (the reference's inter path has
    # no chroma planes to read there: no defined result) and extended_precision_processing (the reference hard-codes it off)
    spb = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 0, seed=2))
    assert spb.d.n_pus > 0
    spb.d.params.chroma_format_idc = 0
    with pytest.raises(backend.De265HipError) as e:
        dec.build(2, spb.desc)
    assert e.value.code == _abi.ERROR_NOT_IMPLEMENTED
    spb.close()
    d.params.extended_precision_processing_flag = 1
    with pytest.raises(backend.De265HipError) as e:
        dec.build(2, sp.desc)
    assert e.value.code == _abi.ERROR_NOT_IMPLEMENTED
    d.params.extended_precision_processing_flag = 0
    # a CTB that no slice covers (a damaged stream: the reference's SAO returns for it, sao.cc:140, and decode_some marks it done,
    # decctx.cc:751-757): de265hip_ctb_info has no "no slice" state, the description is refused - defined, not undefined
    keep = d.ctbs[1].slice_idx
    d.ctbs[1].slice_idx = 0xFFFF
    with pytest.raises(backend.De265HipError) as e:
        dec.build(2, sp.desc)
    assert e.value.code == _abi.ERROR_PARAMETER_OUT_OF_RANGE
    d.ctbs[1].slice_idx = keep
    d.tus[0].x0 = 4000
    with pytest.raises(backend.De265HipError) as e:      # (a malformed TU record: found by the device-side scan, reported when
        bad = dec.build(2, sp.desc)                      #  the picture is launched - or asked for its statistics)
        try:
            dec.run(bad, 2)
        finally:
            bad.free()
    assert e.value.code == _abi.ERROR_PARAMETER_OUT_OF_RANGE
    dec.sync()
    # the host mirror refuses planes that do not match what the slot holds (the C entry point trusts its caller)
    assert dec.dpb_info(2) == (w, h, bd, bd)
    with pytest.raises(ValueError):
        dec.upload(2, pysynth.fill_planes(w + 8, h, bd, 5))
    with pytest.raises(ValueError):
        dec.download(2, w, h, 10)
    with pytest.raises(backend.De265HipError):
        dec.dpb_info(15)                             # never allocated
    del n_pus, C


def test_mc_far_outside_picture_and_shortcuts(dec):
    """64x64 PUs at the picture corners with MVs pointing far outside (clamping path),
    identical-MV bi-prediction (bi->uni shortcut), full-pel vectors."""
    w, h, bd = 256, 128, 10
    sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 0, seed=31, intra_pct=0, split_bias=0, cbf_pct=0,
                                                     mv_sigma_qpel=400))
    d = sp.d
    for i in range(d.n_pus):
        pu = d.pus[i]
        if i % 3 == 0 and pu.pred_flag == 3:
            pu.mv[1][0], pu.mv[1][1] = pu.mv[0][0], pu.mv[0][1]
            pu.ref_idx[1] = pu.ref_idx[0]
        if i % 5 == 0:
            pu.mv[0][0] &= ~3; pu.mv[0][1] &= ~3
    refs = {0: pysynth.fill_planes(w, h, bd, 1), 1: pysynth.fill_planes(w, h, bd, 2)}
    for s, pl in refs.items():
        dec.dpb_alloc(s, w, h, bd); dec.upload(s, pl)
    init = pysynth.fill_planes(w, h, bd, 9)
    exp = [p.copy() for p in init]
    pyoracle.reconstruct(sp.desc, sp.order, refs, exp, last_stage=0)
    dec.dpb_alloc(2, w, h, bd); dec.upload(2, init)
    pic = dec.build(2, sp.desc); dec.run(pic, 0); dec.sync()
    got = dec.download(2, w, h, bd)
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    pic.free()


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_all_intra_worst_case_small_blocks(dec, bd):
    """All 4x4/8x8 intra TUs: the longest dependency chains, every availability pattern."""
    run_case(dec, 192, 128, bd, 2, seed=60 + bd, split_bias=100, log2_max_tb_size=3, stages=(0,))


def test_4k_main10_full_size(dec):
    """BASELINE config 4 at full size: one I and one B picture, bit-exact against the oracle."""
    w, h, bd = 3840, 2160, 10
    i_cfg = pysynth.default_config(w, h, bd, 2, seed=0xDE265004)
    b_cfg = pysynth.default_config(w, h, bd, 0, seed=0xDE265005, ref_slots=[0, 1])
    refs = {1: pysynth.fill_planes(w, h, bd, 77)}
    dec.dpb_alloc(1, w, h, bd); dec.upload(1, refs[1])
    spi = pysynth.SynthPicture(i_cfg)
    exp_i = pyoracle.alloc_planes(w, h, bd)
    pyoracle.reconstruct(spi.desc, spi.order, {}, exp_i)
    dec.dpb_alloc(0, w, h, bd)
    pic = dec.build(0, spi.desc); dec.run(pic, 2); dec.sync()
    got_i = dec.download(0, w, h, bd)
    assert all(np.array_equal(g, e) for g, e in zip(got_i, exp_i))
    pic.free()
    refs[0] = exp_i
    spb = pysynth.SynthPicture(b_cfg)
    exp_b = pyoracle.alloc_planes(w, h, bd)
    pyoracle.reconstruct(spb.desc, spb.order, refs, exp_b)
    pic = dec.build(2, spb.desc); dec.run(pic, 2); dec.sync()      # references the device-resident I picture
    got_b = dec.download(2, w, h, bd)
    assert all(np.array_equal(g, e) for g, e in zip(got_b, exp_b))
    # idempotence: re-running the resident command buffers reproduces the picture
    dec.run(pic, 2); dec.sync()
    assert all(np.array_equal(g, e) for g, e in zip(dec.download(2, w, h, bd), exp_b))
    pic.free()


def test_intra_run_kernel_equals_level_launches(dec):
    """The single-launch run kernel (default) and the one-launch-per-dependency-level path
    (DE265HIP_INTRA_MODE=levels) are two schedules of the same work: identical pictures."""
    import os
    from libde265_amd import backend
    w, h, bd = 416, 240, 10
    refs = {0: pysynth.fill_planes(w, h, bd, 5), 1: pysynth.fill_planes(w, h, bd, 6)}
    outs = []
    for mode in ("levels", None):
        if mode:
            os.environ["DE265HIP_INTRA_MODE"] = mode
        else:
            os.environ.pop("DE265HIP_INTRA_MODE", None)
        d2 = backend.Decoder()
        try:
            for s, pl in refs.items():
                d2.dpb_alloc(s, w, h, bd); d2.upload(s, pl)
            res = []
            for st, seed in ((2, 71), (0, 72)):
                sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=seed, tskip_pct=20, pcm_pct=10,
                                                                 scaling_list=1, n_slices=2))
                d2.dpb_alloc(2, w, h, bd); d2.upload(2, pysynth.fill_planes(w, h, bd, 9))
                pic = d2.build(2, sp.desc); d2.run(pic, 2); d2.sync()
                res.append(d2.download(2, w, h, bd))
                stats = pic.stats()
                assert stats.n_runs > 0 and (stats.n_levels > 1 if mode else stats.n_run_levels >= 1)   # (TU levels: the level-launch schedule's)
                pic.free()
            outs.append(res)
        finally:
            d2.close()
    os.environ.pop("DE265HIP_INTRA_MODE", None)
    for a, b in zip(outs[0], outs[1]):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_concurrent_decoders_do_not_interfere(dec):
    """Several decoders (HIP streams) in flight at once, as bench.py --streams does."""
    from libde265_amd import backend
    w, h, bd = 352, 288, 8
    decs, pics, exps = [], [], []
    for s in range(3):
        d = backend.Decoder()
        sp_i = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 2, seed=300 + s))
        sp_b = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 0, seed=400 + s, ref_slots=[0, 0]))
        e_i = pyoracle.alloc_planes(w, h, bd); pyoracle.reconstruct(sp_i.desc, sp_i.order, {}, e_i)
        e_b = pyoracle.alloc_planes(w, h, bd); pyoracle.reconstruct(sp_b.desc, sp_b.order, {0: e_i}, e_b)
        d.dpb_alloc(0, w, h, bd); d.dpb_alloc(1, w, h, bd)
        decs.append(d); pics.append((d.build(0, sp_i.desc), d.build(1, sp_b.desc), sp_i, sp_b)); exps.append(e_b)
    for rep in range(3):
        for k in range(2):
            for s in range(3):
                decs[s].run(pics[s][k], 2)
    for s in range(3):
        decs[s].sync()
        got = decs[s].download(1, w, h, bd)
        assert all(np.array_equal(g, e) for g, e in zip(got, exps[s])), s
        pics[s][0].free(); pics[s][1].free(); decs[s].close()


def test_randomized_small_pictures(dec):
    """Many small random configurations (sizes down to one CTB, every feature switch drawn at random):
    shakes out corner cases of availability, window clipping and partial strips."""
    import os
    rng = np.random.default_rng(int(os.environ.get("DE265HIP_TEST_SEED", "20260104")))
    for it in range(int(os.environ.get("DE265HIP_TEST_RANDOM", "60"))):       # a longer sweep: DE265HIP_TEST_RANDOM=1000
        log2_ctb = int(rng.choice([4, 5, 6]))
        w = int(rng.integers(1, 26)) * 8 if it % 7 else int(rng.integers(26, 80)) * 8
        h = int(rng.integers(1, 18)) * 8 if it % 7 else int(rng.integers(18, 48)) * 8
        bd = int(rng.choice([8, 9, 10, 12]))
        st = int(rng.choice([0, 1, 2]))
        cols = int(rng.integers(1, 3)) if w >= 128 else 1
        rows = int(rng.integers(1, 3)) if h >= 128 else 1
        ctbs_w = -(-w // (1 << log2_ctb))
        ctbs_h = -(-h // (1 << log2_ctb))
        cols, rows = min(cols, ctbs_w), min(rows, ctbs_h)
        over = dict(log2_ctb_size=log2_ctb, log2_max_tb_size=min(5, log2_ctb),
                    log2_min_tb_size=int(rng.choice([2, 2, 3])),
                    intra_pct=int(rng.choice([0, 15, 50, 100])), tskip_pct=int(rng.choice([0, 30])),
                    bypass_pct=int(rng.choice([0, 10])), pcm_pct=int(rng.choice([0, 20])),
                    pcm_loop_filter_disable=int(rng.integers(0, 2)), scaling_list=int(rng.integers(0, 2)),
                    constrained_intra_pred=int(rng.integers(0, 2)), strong_intra_smoothing=int(rng.integers(0, 2)),
                    weighted_pred=int(rng.integers(0, 2)), n_slices=int(rng.integers(1, 4)),
                    tile_cols=cols, tile_rows=rows, slice_per_tile=int(rng.integers(0, 2)),
                    lf_across_tiles=int(rng.integers(0, 2)), lf_across_slices_pct=int(rng.choice([0, 50, 100])),
                    deblocking=int(rng.integers(0, 4) > 0), sao=int(rng.integers(0, 4) > 0),
                    big_coeff_pct=int(rng.choice([0, 5])), mv_sigma_qpel=int(rng.choice([4, 12, 60])),
                    split_bias=int(rng.choice([0, 50, 100])), cbf_pct=int(rng.choice([0, 60, 100])))
        try:
            run_case(dec, w, h, bd, st, seed=5000 + it, stages=(2,), **over)
        except AssertionError as e:
            raise AssertionError("config %d: %dx%d bd=%d st=%d %r: %s" % (it, w, h, bd, st, over, e))


def random_midsize_config(rng):
    """One random mid-size configuration (320..1928 x 240..1088, every feature switch drawn at random); also used by
    tools/exp/sweep.py for long sweeps."""
    log2_ctb = int(rng.choice([4, 5, 6, 6]))
    w = int(rng.integers(40, 241)) * 8; h = int(rng.integers(30, 137)) * 8
    bd = int(rng.choice([8, 10, 10, 12])); st = int(rng.choice([0, 0, 1, 2]))
    over = dict(log2_ctb_size=log2_ctb, log2_max_tb_size=min(5, log2_ctb), log2_min_tb_size=int(rng.choice([2, 2, 3])),
                intra_pct=int(rng.choice([5, 15, 40, 100])), tskip_pct=int(rng.choice([0, 20])), bypass_pct=int(rng.choice([0, 5])),
                pcm_pct=int(rng.choice([0, 10])), scaling_list=int(rng.integers(0, 2)), constrained_intra_pred=int(rng.integers(0, 2)),
                strong_intra_smoothing=int(rng.integers(0, 2)), weighted_pred=int(rng.integers(0, 2)), n_slices=int(rng.integers(1, 5)),
                split_bias=int(rng.choice([0, 30, 50, 80, 100])), cbf_pct=int(rng.choice([30, 60, 100])), mv_sigma_qpel=int(rng.choice([4, 12, 80])),
                pcm_loop_filter_disable=int(rng.integers(0, 2)), lf_across_slices_pct=int(rng.choice([0, 50, 100])),
                lf_across_tiles=int(rng.integers(0, 2)), big_coeff_pct=int(rng.choice([0, 0, 2])), bi_pct=int(rng.choice([0, 60, 100])),
                amp=int(rng.integers(0, 2)), deblocking=int(rng.choice([1, 1, 1, 0])), sao=int(rng.choice([1, 1, 1, 0])))
    if log2_ctb >= 5 and rng.integers(0, 3) == 0:        # tiles (the slice layout then follows the tiles)
        over.update(tile_cols=int(rng.integers(1, 4)), tile_rows=int(rng.integers(1, 3)), slice_per_tile=int(rng.integers(0, 2)))
    qlo = int(rng.integers(0, 40)); over.update(qp_min=qlo, qp_max=int(rng.integers(qlo, 52)))
    return w, h, bd, st, over


def test_randomized_midsize_pictures(dec):
    """Random pictures of everyday sizes: long dependency chains, many runs per picture, partial CTBs at both edges
    (a sweep of this kind found the missing dependency of test_strong_smoothing_decision_samples_are_dependencies).
    Longer: DE265HIP_TEST_RANDOM_MID=500, or tools/exp/sweep.py <seed> <n>."""
    import os
    rng = np.random.default_rng(int(os.environ.get("DE265HIP_TEST_SEED", "20261004")))
    for it in range(int(os.environ.get("DE265HIP_TEST_RANDOM_MID", "24"))):
        w, h, bd, st, over = random_midsize_config(rng)
        try:
            run_case(dec, w, h, bd, st, seed=9000 + it, stages=(2,), **over)
        except AssertionError as e:
            raise AssertionError("config %d: %dx%d bd=%d st=%d %r: %s" % (it, w, h, bd, st, over, e))


def test_recorder_submit_matches_oracle(dec):
    """Incremental interface (de265hip_record_* + recorder_submit) on the GPU: same picture as the oracle."""
    w, h, bd = 352, 288, 10
    cfg = pysynth.default_config(w, h, bd, 0, seed=77, pcm_pct=10, tskip_pct=20, n_slices=2, scaling_list=1)
    sp = pysynth.SynthPicture(cfg)
    refs = {0: pysynth.fill_planes(w, h, bd, 177), 1: pysynth.fill_planes(w, h, bd, 277)}
    for slot, pl in refs.items():
        dec.dpb_alloc(slot, w, h, bd)
        dec.upload(slot, pl)
    sf = np.ctypeslib.as_array(sp.d.scaling_factors, shape=(_abi.SCALING_BLOB_BYTES,)).copy()
    rec = backend.Recorder(sp.d.params, sf)
    rec.record_desc(sp.d)
    init = pysynth.fill_planes(w, h, bd, 999)
    exp = [p.copy() for p in init]
    pyoracle.reconstruct(sp.desc, sp.order, refs, exp)
    dec.dpb_alloc(2, w, h, bd)                # (the slot may hold a picture of another size from an earlier test)
    dec.upload(2, init)
    pic = rec.submit(dec, 2)
    dec.run(pic)
    dec.sync()
    got = dec.download(2, w, h, bd)
    pic.free()
    rec.free()
    for c in range(3):
        assert np.array_equal(got[c], exp[c])


@pytest.mark.parametrize("env", [{"DE265HIP_RUN_WAVES": "1"}, {"DE265HIP_RUN_WAVES": "2"}, {"DE265HIP_NO_MICRO": "1"},
                                 {"DE265HIP_MICRO_TUS": "4"}, {"DE265HIP_TICKET_BATCH": "4"}, {"DE265HIP_SEPARATE_BS": "1"},
                                 {"DE265HIP_TWO_PASS_DEBLOCK": "1"}, {"DE265HIP_MICRO16": "0"}, {"DE265HIP_RESID16_BIG": "1"}, {"DE265HIP_RESID_ONE_LAUNCH": "0"},
                                 {"DE265HIP_LF_TILE": "1"}, {"DE265HIP_NO_MERGE": "1"}, {"DE265HIP_RUN_DIRECT": "1"}, {"DE265HIP_SAO_STRIPS": "1"},
                                 {"DE265HIP_NO_MAILBOX": "1"}, {"DE265HIP_NO_MB_PHASES": "1"}, {"DE265HIP_NO_FRONT": "1"}, {"DE265HIP_MC_PATHS": "0"}, {"DE265HIP_MC_PATHS": "1"},
                                 {"DE265HIP_MC_PATHS": "2"}, {"DE265HIP_HOST_SCAN": "1"}, {"DE265HIP_HOST_SCAN": "1", "DE265HIP_NO_MB_PHASES": "1"},
                                 {"DE265HIP_NO_MODE_DEPS": "1"}, {"DE265HIP_NO_DENSE": "1"}])
def test_run_kernel_schedule_variants(env):
    """k_run's schedule knobs (wavefronts per workgroup, micro runs on/off, tickets per draw, edge mailboxes with and without the phased hand-over, front runs) and the forms MC tasks
    take (round 2's tiles only, + quads of small blocks, + chunks) only change who does what when: every variant is bit-exact against the oracle."""
    import os
    from libde265_amd import backend
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    d2 = backend.Decoder()
    try:
        run_case(d2, 416, 240, 10, 2, seed=91, stages=(2,), tskip_pct=20, pcm_pct=8, n_slices=2)
        run_case(d2, 352, 288, 8, 0, seed=92, stages=(2,), tskip_pct=10, scaling_list=1, constrained_intra_pred=1)
        run_case(d2, 512, 320, 10, 0, seed=93, stages=(2,), intra_pct=60, split_bias=80)
        run_case(d2, 448, 256, 10, 2, seed=94, stages=(2,), split_bias=40)            # all-intra, big TUs: dense CTB runs (edge mailboxes)
        run_case(d2, 1024, 576, 10, 2, seed=95, stages=(2,), split_bias=70, strong_intra_smoothing=0)      # many CTBs side by side: the phased hand-over at work
    finally:
        d2.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_decoder_freed_before_its_pictures():
    """Lifetime rule of include/de265_hip.h: decoder_free() with pictures alive orphans them; an orphan can still
    be asked for its stats and freed, never run.  (Round 1: picture_free dereferenced the dead decoder.)"""
    from libde265_amd import backend
    w, h, bd = 128, 64, 8
    sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 2, seed=5))
    d1 = backend.Decoder()
    d1.dpb_alloc(2, w, h, bd)
    pics = [d1.build(2, sp.desc) for _ in range(3)]
    d1.run(pics[0], 2)
    pics[1].free()                                   # the normal order for one of them
    d1.close()                                       # decoder first ...
    assert pics[0].stats().n_tu_tasks > 0            # ... the handles stay valid
    d2 = backend.Decoder()
    try:
        with pytest.raises(backend.De265HipError) as e:
            d2.run(pics[0], 2)                       # neither on its dead decoder nor on another one
        assert e.value.code == _abi.ERROR_PARAMETER_OUT_OF_RANGE
    finally:
        pics[0].free(); pics[2].free()               # ... pictures second
        d2.close()


def test_arena_pool_reuse_keeps_pictures_apart(dec):
    """build -> run -> free in a loop recycles pooled device arenas and pinned staging buffers without any host-side
    wait: every picture must still come out right although its arena belonged to another picture a moment ago."""
    w, h, bd = 352, 288, 10
    refs = {0: pysynth.fill_planes(w, h, bd, 1), 1: pysynth.fill_planes(w, h, bd, 2)}
    for s, pl in refs.items():
        dec.dpb_alloc(s, w, h, bd); dec.upload(s, pl)
    dec.dpb_alloc(2, w, h, bd)
    sps, exps = [], []
    for k in range(6):
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, k % 3, seed=700 + k, n_slices=1 + k % 2))
        exp = pyoracle.alloc_planes(w, h, bd)
        pyoracle.reconstruct(sp.desc, sp.order, refs, exp)
        sps.append(sp); exps.append(exp)
    for rep in range(3):
        for k in range(6):
            pic = dec.build(2, sps[k].desc)
            try:
                dec.upload(2, pyoracle.alloc_planes(w, h, bd))
                dec.run(pic, 2)
                if (rep + k) % 2:                    # sometimes free while the kernels are still queued
                    pic.free()
                dec.sync()
                got = dec.download(2, w, h, bd)
            finally:
                pic.free()
            assert all(np.array_equal(g, e) for g, e in zip(got, exps[k])), (rep, k)


def test_reference_picture_exchange_on_dpb_planes(dec):
    """SURVEY 8e on the product's own picture store: slot -> slot device copy between two decoders
    (de265hip_dpb_copy), and the zero-copy torch views RCCL sends/receives on (farm.dpb_plane_tensors):
    a B picture decoded by the second decoder from the handed-over references equals the reference's."""
    import torch
    from libde265_amd import backend, farm
    w, h, bd = 352, 288, 10
    refs = {0: pysynth.fill_planes(w, h, bd, 11), 1: pysynth.fill_planes(w, h, bd, 12)}
    for s, pl in refs.items():
        dec.dpb_alloc(s, w, h, bd); dec.upload(s, pl)
    d2 = backend.Decoder()
    try:
        dec.copy_slot_to(0, d2, 0)                       # C ABI, device to device, stream-ordered
        d2.dpb_alloc(1, w, h, bd)
        src, dst = farm.dpb_plane_tensors(dec, 1), farm.dpb_plane_tensors(d2, 1)
        dec.sync()
        for a, b in zip(src, dst):                       # what dist.send / dist.recv do between ranks, on one GPU
            assert a.is_cuda and a.dtype == torch.uint8 and a.numel() == b.numel()
            b.copy_(a)
        torch.cuda.synchronize()
        for s in (0, 1):
            got = d2.download(s, w, h, bd)
            assert all(np.array_equal(g, e) for g, e in zip(got, refs[s]))
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 0, seed=4242))
        exp = pyoracle.alloc_planes(w, h, bd)
        pyoracle.reconstruct(sp.desc, sp.order, refs, exp)
        d2.dpb_alloc(2, w, h, bd)
        pic = d2.build(2, sp.desc)
        try:
            d2.run(pic, 2); d2.sync()
            got = d2.download(2, w, h, bd)
        finally:
            pic.free()
        assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    finally:
        d2.close()


@pytest.mark.parametrize("cf", [2, 3, 0])
def test_reference_picture_exchange_of_other_chroma_formats(cf):
    """The hand-over views follow the slot's chroma format (4:2:2 / 4:4:4 chroma planes are as high as the luma plane, a
    monochrome slot has none): the planes of a slot, and the slot as ONE message (farm.dpb_slot_tensor), arrive whole."""
    import torch
    from libde265_amd import backend, farm
    w, h, bd = 352, 288, 10
    pl = pysynth.fill_planes(w, h, bd, 31 + cf)
    shapes = farm.plane_shapes(w, h, cf)
    planes = [np.ascontiguousarray(np.resize(p, sh)) if sh[0] else np.zeros((0, 0), p.dtype) for p, sh in zip([pl[0], pl[1], pl[2]], shapes)]
    a, b = backend.Decoder(), backend.Decoder()
    try:
        for d in (a, b):
            for s in (1, 2):
                d.dpb_alloc(s, w, h, bd, chroma_format=cf)
        a.upload(1, planes)
        assert farm.plane_rows(a, 1) == [sh[0] for sh in shapes]
        src, dst = farm.dpb_plane_tensors(a, 1), farm.dpb_plane_tensors(b, 1)
        assert len(src) == (1 if cf == 0 else 3)
        a.sync()
        for x, y in zip(src, dst):
            assert x.numel() == y.numel()
            y.copy_(x)
        one_src, one_dst = farm.dpb_slot_tensor(a, 1), farm.dpb_slot_tensor(b, 2)      # the whole slot in one message
        assert one_src.numel() == one_dst.numel() >= sum(t.numel() for t in src)
        one_dst.copy_(one_src)
        torch.cuda.synchronize()
        for s in (1, 2):
            got = b.download(s, w, h, bd)
            assert [g.shape for g in got] == [tuple(sh) for sh in shapes]
            assert all(np.array_equal(g, e) for g, e in zip(got, planes))
        pp = backend.PinnedPlanes(w, h, bd, chroma_format=cf)
        assert [p.shape for p in pp.planes] == [tuple(sh) for sh in shapes]
        pp.free()
    finally:
        a.close(); b.close()


def test_concurrent_builds_on_one_decoder(dec):
    """de265hip_picture_build / _free from several host threads on ONE decoder (include/de265_hip.h THREADS; what
    bench.py's host_inclusive leg does): the pools and the live list are shared, every picture still comes out right."""
    from concurrent.futures import ThreadPoolExecutor
    w, h, bd = 352, 288, 8
    refs = {0: pysynth.fill_planes(w, h, bd, 21), 1: pysynth.fill_planes(w, h, bd, 22)}
    for s, pl in refs.items():
        dec.dpb_alloc(s, w, h, bd); dec.upload(s, pl)
    dec.dpb_alloc(2, w, h, bd)
    sps, exps = [], []
    for k in range(8):
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, (0, 1, 2, 0)[k % 4], seed=900 + k, tskip_pct=10, n_slices=1 + k % 3))
        exp = pyoracle.alloc_planes(w, h, bd)
        pyoracle.reconstruct(sp.desc, sp.order, refs, exp)
        sps.append(sp); exps.append(exp)
    with ThreadPoolExecutor(8) as pool:
        for rep in range(4):
            futs = [pool.submit(dec.build, 2, sps[k].desc) for k in range(8)]       # eight builds in flight at once
            for k, f in enumerate(futs):
                pic = f.result()
                try:
                    dec.upload(2, pyoracle.alloc_planes(w, h, bd))
                    dec.run(pic, 2)
                    dec.sync()
                    got = dec.download(2, w, h, bd)
                finally:
                    pool.submit(pic.free).result()                                   # freed on a pool thread as well
                assert all(np.array_equal(g, e) for g, e in zip(got, exps[k])), (rep, k)


def test_extreme_picture_sizes(dec):
    """Smallest (one 8x8 CU) and a very large picture (8K UHD, 7680x4320 Main10, I and B): coordinates, dependency levels and
    per-picture tables at their extremes, bit-exact against the live compiled reference where it is present, else the restatement."""
    import pyref
    for (w, h, bd, st, seed, over) in [(8, 8, 8, 2, 1, dict(log2_ctb_size=4, log2_max_tb_size=3)),
                                       (8, 8, 10, 0, 2, dict(log2_ctb_size=4, log2_max_tb_size=3)),
                                       (16, 8, 8, 1, 3, dict(log2_ctb_size=4, log2_max_tb_size=4)),
                                       (7680, 4320, 10, 0, 0xDE265008, dict()),
                                       (7680, 4320, 10, 2, 0xDE265009, dict())]:
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=seed, **over))
        refs = {0: pysynth.fill_planes(w, h, bd, 31), 1: pysynth.fill_planes(w, h, bd, 32)}
        init = pysynth.fill_planes(w, h, bd, 33)
        exp = [p.copy() for p in init]
        if pyref.available():
            pyref.reconstruct(sp.desc, sp.order, refs, exp, sp.structure())
        else:
            pyoracle.reconstruct(sp.desc, sp.order, refs, exp)
        for s, pl in refs.items():
            dec.dpb_alloc(s, w, h, bd); dec.upload(s, pl)
        dec.dpb_alloc(2, w, h, bd); dec.upload(2, init)
        pic = dec.build(2, sp.desc)
        try:
            dec.run(pic, 2); dec.sync()
            got = dec.download(2, w, h, bd)
        finally:
            pic.free()
        for c in range(3):
            bad = np.argwhere(got[c] != exp[c])
            assert bad.size == 0, "%dx%d bd=%d st=%d comp %d: %d samples differ, first %s" % (w, h, bd, st, c, len(bad), tuple(bad[0]))
        for s in (0, 1, 2):
            dec.dpb_alloc(s, 64, 64, 8)                # (give the 8K slots back)


def test_async_copy_out_overlaps_later_pictures_and_survives_slot_reuse(dec):
    """SURVEY 8(f3), picture-level pipelining through the C ABI: pictures are enqueued back to back with
    de265hip_dpb_download_async behind each (pinned planes from de265hip_host_alloc), nothing waits on the host until
    the end; one slot is decoded into twice in a row, so the second picture's kernels must wait for the first one's
    copy-out by themselves.  Every picture that arrives must be the one the oracle computes."""
    w, h, bd = 352, 288, 10
    refs = {0: pysynth.fill_planes(w, h, bd, 21), 1: pysynth.fill_planes(w, h, bd, 22)}
    for s, pl in refs.items():
        dec.dpb_alloc(s, w, h, bd); dec.upload(s, pl)
    slots = [2, 3, 3, 4, 2, 2]                       # 3,3 and 2,2: immediate reuse of a slot whose copy-out is still in flight
    for s in set(slots):
        dec.dpb_alloc(s, w, h, bd); dec.upload(s, pyoracle.alloc_planes(w, h, bd))
    sps, exps = [], []
    for k in range(len(slots)):
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, k % 3, seed=900 + k, n_slices=1 + k % 2))
        exp = pyoracle.alloc_planes(w, h, bd)
        pyoracle.reconstruct(sp.desc, sp.order, refs, exp)
        sps.append(sp); exps.append(exp)
    pics, pend = [], []
    try:
        for k, s in enumerate(slots):
            pics.append(dec.build(s, sps[k].desc))
            dec.run(pics[-1], 2)
            pend.append(dec.download_async(s, w, h, bd))
        for k in reversed(range(len(slots))):        # any order of waiting
            got = pend[k].wait()
            assert all(np.array_equal(g, e) for g, e in zip(got, exps[k])), k
    finally:
        dec.sync()
        for p in pend:
            p.free()
        for p in pics:
            p.free()


def test_copy_out_of_all_planes_in_one_call(dec):
    """de265hip_dpb_download_planes_async: pinned planes leave by the library's copy-out kernel (contiguous rows, padded rows,
    a plane whose byte count is no multiple of 16), pageable planes and odd strides by the runtime's copy - every form must
    deliver what de265hip_dpb_download does, and bytes outside the rows stay as they were."""
    import ctypes as C
    from libde265_amd import backend
    L = backend.lib()
    for (w, h, bd, cf) in [(352, 288, 10, 1), (200, 104, 8, 1), (64, 72, 8, 3), (136, 64, 10, 2), (72, 40, 8, 0)]:
        planes = pysynth.fill_planes(w, h, bd, 77) if cf == 1 else None
        dec.dpb_alloc(5, w, h, bd, chroma_format=cf)
        if planes is None:
            rng = np.random.default_rng(w * h + cf)
            cw, ch = dec._chroma_dims(5, w, h)
            dt = np.uint16 if bd > 8 else np.uint8
            planes = [rng.integers(0, 1 << bd, size=sh, dtype=dt) for sh in ((h, w), (ch, cw), (ch, cw))]
        dec.upload(5, planes)
        bpp = planes[0].dtype.itemsize
        for pad, pinned in [(0, True), (32, True), (6, True), (0, False), (10, False)]:
            bufs, views, ptrs, strides = [], [], [], []
            for pl in planes:
                rows, cols = pl.shape
                if rows == 0 or cols == 0:
                    ptrs.append(None); strides.append(0); views.append(None); bufs.append(None); continue
                stride = cols * bpp + pad
                nbytes = stride * rows
                if pinned:
                    ptr = L.de265hip_host_alloc(nbytes); assert ptr
                    raw = np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), np.uint8)
                else:
                    raw = np.empty(nbytes, np.uint8); ptr = raw.ctypes.data
                raw[:] = 0xA5
                bufs.append((ptr, raw)); ptrs.append(ptr); strides.append(stride); views.append(raw.reshape(rows, stride))
            try:
                dec.download_planes_async(5, ptrs, strides)
                dec.wait_slot(5)
                for c, pl in enumerate(planes):
                    if views[c] is None:
                        continue
                    rb = pl.shape[1] * bpp
                    got = views[c][:, :rb].copy().view(pl.dtype)
                    assert np.array_equal(got, pl), (w, h, bd, cf, pad, pinned, c)
                    assert (views[c][:, rb:] == 0xA5).all(), "bytes behind the rows were written"
            finally:
                for b in bufs:
                    if b is not None and pinned:
                        L.de265hip_host_free(b[0])
    dec.dpb_alloc(5, 64, 64, 8)


def test_pipeline_builds_concurrently_and_launches_in_order(dec):
    """SURVEY 8(f3) through the C ABI's own pipeline (de265hip_pipeline_*): a chain of pictures in which every picture
    predicts from the one before it (its DPB slot) is submitted without waiting; three worker threads prepare and build them
    concurrently, the pipeline launches them in submission order and copies each out into pinned planes.  Every picture must be
    what the oracle computes from the ORACLE's own chain of reference pictures - any launch out of order would read a slot too early."""
    from libde265_amd import backend
    w, h, bd = 352, 288, 10
    n = 7
    first = pysynth.fill_planes(w, h, bd, 31)
    dec.dpb_alloc(0, w, h, bd); dec.upload(0, first)
    for s in range(1, n + 1):
        dec.dpb_alloc(s, w, h, bd)
    sps, exps, prev = [], [], first
    for k in range(n):
        # a P picture whose only reference is slot k = where picture k-1 is decoded into (picture k goes into slot k+1)
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 1, seed=950 + k, ref_slots=[k]))
        exp = pyoracle.alloc_planes(w, h, bd)
        pyoracle.reconstruct(sp.desc, sp.order, {k: prev}, exp)
        sps.append(sp); exps.append(exp); prev = exp
    pipe = backend.Pipeline(dec, 3)
    pins = [backend.PinnedPlanes(w, h, bd) for _ in range(n)]
    try:
        tickets = []
        for k in range(n):
            def make(k=k):
                d = sps[k].d
                sf = np.ctypeslib.as_array(d.scaling_factors, shape=(_abi.SCALING_BLOB_BYTES,)).copy() if d.params.scaling_list_enable_flag else None
                rec = backend.Recorder(d.params, sf)
                rec.record_desc(d)
                return rec
            tickets.append(pipe.submit(k + 1, make, pins[k]))
        for k in reversed(range(n)):
            pipe.wait(tickets[k])
            assert all(np.array_equal(g, e) for g, e in zip(pins[k].planes, exps[k])), k
        pipe.drain()
    finally:
        pipe.close()
        for p in pins:
            p.free()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_UNGATED_SCRIPT = r"""
import os, sys
sys.path[:0] = [ROOT, ROOT + "/oracle", ROOT + "/tools"]
import torch, numpy as np
import pysynth, pyoracle
from libde265_amd import backend
w, h, bd = 352, 288, 10
refs = {0: pysynth.fill_planes(w, h, bd, 21), 1: pysynth.fill_planes(w, h, bd, 22)}
sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 0, seed=4242))
exp = pyoracle.alloc_planes(w, h, bd)
pyoracle.reconstruct(sp.desc, sp.order, refs, exp)
dec = backend.Decoder()
for s_, pl in refs.items():
    dec.dpb_alloc(s_, w, h, bd); dec.upload(s_, pl)
dec.dpb_alloc(2, w, h, bd)
pipe = backend.Pipeline(dec, 2)
pin = backend.PinnedPlanes(w, h, bd)
t = pipe.submit_desc(2, sp.desc, pin)
pipe.wait(t)
ok = all(np.array_equal(g, e) for g, e in zip(pin.planes, exp))
pipe.drain(); pipe.close(); pin.free(); dec.close()
print("UNGATED_OK" if ok else "UNGATED_MISMATCH")
"""


def test_switches_are_ignored_without_the_tuning_gate():
    """csrc/env.h: the DE265HIP_* switches - some of which make a decoder's results invalid - are honoured only in a process that
    sets DE265HIP_TUNING=1.  A process WITHOUT it, with three such switches in its environment, decodes and delivers a picture
    exactly as the oracle computes it (with the gate open the same switches skip the reconstruction and the copy-out)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "DE265HIP_TUNING"}
    env.update(DE265HIP_PIPE_NO_RUN="1", DE265HIP_OUT_COPY="none", DE265HIP_NO_FRONT="1")
    script = "ROOT = %r\n" % ROOT + _UNGATED_SCRIPT
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert "UNGATED_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    env["DE265HIP_TUNING"] = "1"                            # ... and with the gate open they do what they say
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert "UNGATED_MISMATCH" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    # the earlier forms of upload and copy-out (streams instead of the HSA runtime / the output thread) stay selectable and correct
    env = {k: v for k, v in os.environ.items()}
    env.update(DE265HIP_TUNING="1", DE265HIP_UPLOAD="stream", DE265HIP_OUT_COPY="dma")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert "UNGATED_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    for form in ("kernel", "deferred"):
        env.update(DE265HIP_OUT_COPY=form)
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
        assert "UNGATED_OK" in r.stdout, (form, r.stdout[-500:], r.stderr[-1500:])


def test_pipeline_output_queue_deeper_than_the_dpb_cycle(dec):
    """A chain of ten pictures that cycles through three DPB slots, each copied out into pinned planes of its own, collected only
    after all have been submitted: by then every slot has been decoded into and copied out again up to three times.  A ticket
    waits for ITS picture's copy-out (de265hip_dpb_wait_copy_out), and every picture arrives as the oracle computes it."""
    from libde265_amd import backend
    w, h, bd = 352, 288, 10
    n = 10
    first = pysynth.fill_planes(w, h, bd, 41)
    dec.dpb_alloc(0, w, h, bd); dec.upload(0, first)
    for s in (1, 2, 3):
        dec.dpb_alloc(s, w, h, bd)
    slot_of = lambda k: 1 + k % 3
    sps, exps, prev = [], [], first
    for k in range(n):
        ref = 0 if k == 0 else slot_of(k - 1)
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 1, seed=1950 + k, ref_slots=[ref]))
        exp = pyoracle.alloc_planes(w, h, bd)
        pyoracle.reconstruct(sp.desc, sp.order, {ref: prev}, exp)
        sps.append(sp); exps.append(exp); prev = exp
    pipe = backend.Pipeline(dec, 3)
    pins = [backend.PinnedPlanes(w, h, bd) for _ in range(n)]
    try:
        tickets = [pipe.submit(slot_of(k), _recorder_maker(sps[k]), pins[k]) for k in range(n)]
        for k in range(n):
            pipe.wait(tickets[k])
            assert all(np.array_equal(g, e) for g, e in zip(pins[k].planes, exps[k])), k
        pipe.drain()
        # the numbered wait by itself: two copy-outs of one slot, the first one waited for by its number
        a, b = backend.PinnedPlanes(w, h, bd), backend.PinnedPlanes(w, h, bd)
        id1 = dec.download_planes_async(1, a.ptrs, a.strides)
        id2 = dec.download_planes_async(1, b.ptrs, b.strides)
        assert id2 == id1 + 1
        dec.wait_slot(1, id1)
        assert all(np.array_equal(g, e) for g, e in zip(a.planes, exps[9]))      # slot 1 holds picture 9
        dec.wait_slot(1, id2)
        assert all(np.array_equal(g, e) for g, e in zip(b.planes, exps[9]))
        a.free(); b.free()
    finally:
        pipe.close()
        for p in pins:
            p.free()


def _recorder_maker(sp):
    def make():
        d = sp.d
        sf = np.ctypeslib.as_array(d.scaling_factors, shape=(_abi.SCALING_BLOB_BYTES,)).copy() if d.params.scaling_list_enable_flag else None
        rec = backend.Recorder(d.params, sf)
        rec.record_desc(d)
        return rec
    return make


def test_picture_size_change_with_pictures_still_in_flight():
    """A new sequence with another picture size while pictures of the old size are built but not launched yet (the pipeline
    builds ahead of its launches): the old pictures must keep their planes - slots and the SAO target are re-allocated in
    launch order, by the first picture of the new size - and every picture of both sizes must come out as the oracle has it.
    A picture of the new size that refers to a slot still holding the old size is refused when it is launched."""
    wa, ha, wb, hb, bd = 416, 240, 208, 120, 10
    d = backend.Decoder()
    first = pysynth.fill_planes(wa, ha, bd, 41)
    d.dpb_alloc(0, wa, ha, bd); d.upload(0, first)
    jobs, prev = [], first                           # (slot, SynthPicture, expected planes, (w, h))
    for k in range(4):                               # size A: a chain of P pictures, picture k -> slot k + 1
        sp = pysynth.SynthPicture(pysynth.default_config(wa, ha, bd, 1, seed=1200 + k, ref_slots=[k]))
        exp = pyoracle.alloc_planes(wa, ha, bd)
        pyoracle.reconstruct(sp.desc, sp.order, {k: prev}, exp)
        jobs.append((k + 1, sp, exp, (wa, ha))); prev = exp
    spi = pysynth.SynthPicture(pysynth.default_config(wb, hb, bd, 2, seed=1210))       # size B: an I picture into slot 1 ...
    expi = pyoracle.alloc_planes(wb, hb, bd)
    pyoracle.reconstruct(spi.desc, spi.order, {}, expi)
    jobs.append((1, spi, expi, (wb, hb)))
    spp = pysynth.SynthPicture(pysynth.default_config(wb, hb, bd, 1, seed=1211, ref_slots=[1]))   # ... and a P picture from it into slot 2
    expp = pyoracle.alloc_planes(wb, hb, bd)
    pyoracle.reconstruct(spp.desc, spp.order, {1: expi}, expp)
    jobs.append((2, spp, expp, (wb, hb)))
    bad = pysynth.SynthPicture(pysynth.default_config(wb, hb, bd, 1, seed=1212, ref_slots=[4]))   # slot 4 still holds size A
    pipe = backend.Pipeline(d, 3)
    pins = [backend.PinnedPlanes(w, h, bd) for _, _, _, (w, h) in jobs]
    try:
        tickets = [pipe.submit(slot, _recorder_maker(sp), pins[i]) for i, (slot, sp, _, _) in enumerate(jobs)]
        t_bad = pipe.submit(5, _recorder_maker(bad), None)
        for i in reversed(range(len(jobs))):
            pipe.wait(tickets[i])
            assert all(np.array_equal(g, e) for g, e in zip(pins[i].planes, jobs[i][2])), i
        with pytest.raises(backend.De265HipError) as e:
            pipe.wait(t_bad)
        assert e.value.code == _abi.ERROR_PARAMETER_OUT_OF_RANGE
        assert d.dpb_info(1)[:2] == (wb, hb) and d.dpb_info(3)[:2] == (wa, ha)
        pipe.drain()
    finally:
        pipe.close()
        for p in pins:
            p.free()
        d.close()


def test_dpb_copy_into_a_slot_that_queued_pictures_still_read():
    """Open-GOP hand-over into a LIVE reference slot: the destination decoder has pictures queued that predict
    from slot 0 when de265hip_dpb_copy overwrites slot 0 from another decoder.  The queued pictures must see the old content,
    the picture enqueued after the copy the new one."""
    w, h, bd = 1280, 720, 8
    old0, ref1, new0 = (pysynth.fill_planes(w, h, bd, s) for s in (51, 52, 53))
    sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 0, seed=1300, ref_slots=[0, 1], intra_pct=5))
    exp_old, exp_new = pyoracle.alloc_planes(w, h, bd), pyoracle.alloc_planes(w, h, bd)
    pyoracle.reconstruct(sp.desc, sp.order, {0: old0, 1: ref1}, exp_old)
    pyoracle.reconstruct(sp.desc, sp.order, {0: new0, 1: ref1}, exp_new)
    src, dst = backend.Decoder(), backend.Decoder()
    try:
        src.dpb_alloc(7, w, h, bd); src.upload(7, new0)
        for s, pl in ((0, old0), (1, ref1)):
            dst.dpb_alloc(s, w, h, bd); dst.upload(s, pl)
        for s in (2, 3):
            dst.dpb_alloc(s, w, h, bd); dst.upload(s, pyoracle.alloc_planes(w, h, bd))
        pic2, pic3 = dst.build(2, sp.desc), dst.build(3, sp.desc)
        for _ in range(8):                             # a queue of pictures that read slot 0 ...
            dst.run(pic2, 2)
        pend = dst.download_async(0, w, h, bd)         # ... and a copy-out of its old content
        src.copy_slot_to(7, dst, 0)                    # overwrite it, stream-ordered on both sides
        dst.run(pic3, 2)
        dst.sync()
        assert all(np.array_equal(g, e) for g, e in zip(pend.wait(), old0))
        assert all(np.array_equal(g, e) for g, e in zip(dst.download(2, w, h, bd), exp_old))
        assert all(np.array_equal(g, e) for g, e in zip(dst.download(3, w, h, bd), exp_new))
        pend.free(); pic2.free(); pic3.free()
    finally:
        src.close(); dst.close()


@pytest.mark.parametrize("host_scan", [0, 1])
def test_expired_dependency_wait_surfaces_as_decoding_error(monkeypatch, host_scan):
    """The one device-side failure mode of the design: a k_run wavefront waits for a producer run whose flag never comes.
    Fault injection (de265hip_debug_fault_injection, an explicit test entry point - no environment switch of the shipped library
    makes a decoder fail: one run that others depend on is left out of the ticket list, with a short bound on the waits): the
    picture must FAIL - de265hip_decoder_sync returns DE265_ERROR_UNSPECIFIED_DECODING_ERROR - within the bound instead of
    hanging or passing as a wrong picture, and the next picture on the same decoder must decode correctly.  Both scans."""
    import time
    w, h, bd = 416, 240, 8
    if host_scan:
        monkeypatch.setenv("DE265HIP_HOST_SCAN", "1")
    monkeypatch.setenv("DE265HIP_TEST_DROP_PRODUCER", "1")      # (the shipped library ignores it: checked below)
    monkeypatch.setenv("DE265HIP_TEST_SPIN_LIMIT", "400")
    d = backend.Decoder()
    try:
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 2, seed=1400))
        exp = pyoracle.alloc_planes(w, h, bd)
        pyoracle.reconstruct(sp.desc, sp.order, {}, exp)
        d.dpb_alloc(2, w, h, bd)
        ok = d.build(2, sp.desc)                         # the environment alone changes nothing
        d.run(ok, 2); d.sync()
        assert all(np.array_equal(g, e_) for g, e_ in zip(d.download(2, w, h, bd), exp))
        ok.free()
        assert backend.lib().de265hip_debug_fault_injection(d._h, 1, 400) == 0
        bad = d.build(2, sp.desc)
        assert backend.lib().de265hip_debug_fault_injection(d._h, 0, 400) == 0
        assert bad.stats().n_run_levels > 1              # there is something to wait for
        t0 = time.perf_counter()
        d.run(bad, 2)
        with pytest.raises(backend.De265HipError) as e:
            d.sync()
        assert e.value.code == _abi.ERROR_DECODING
        assert time.perf_counter() - t0 < 5.0
        bad.free()
        good = d.build(2, sp.desc)                       # same decoder, same slot: the error state is gone
        d.run(good, 2); d.sync()
        assert all(np.array_equal(g, e_) for g, e_ in zip(d.download(2, w, h, bd), exp))
        good.free()
    finally:
        d.close()


@pytest.mark.parametrize("host_scan", [0, 1])
def test_coefficient_position_beyond_its_block_surfaces_as_decoding_error(monkeypatch, host_scan):
    """A coefficient position outside its TU's nT x nT block (no parser of the reference produces one) is caught on the device,
    behind the upload: the position is folded into its block (no kernel ever indexes beyond it) and the picture FAILS with
    DE265_ERROR_UNSPECIFIED_DECODING_ERROR - at its launch when the device-side scan found it (the scan's verdict is read
    there), at the next synchronisation with the round-3 host scan (k_check_coeffs) - and the next picture is fine."""
    w, h, bd = 416, 240, 8
    if host_scan:
        monkeypatch.setenv("DE265HIP_HOST_SCAN", "1")
    d = backend.Decoder()
    try:
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 2, seed=1401, log2_max_tb_size=3))
        exp = pyoracle.alloc_planes(w, h, bd)
        pyoracle.reconstruct(sp.desc, sp.order, {}, exp)
        dd = sp.d
        assert dd.n_coeffs > 10
        keep = dd.coeff_pos[5]
        dd.coeff_pos[5] = 60000                          # every TU is at most 8x8: beyond any block
        d.dpb_alloc(2, w, h, bd)
        bad = d.build(2, sp.desc)                        # (not refused at build: the check travels with the upload)
        with pytest.raises(backend.De265HipError) as e:
            d.run(bad, 2)
            d.sync()
        assert e.value.code == _abi.ERROR_DECODING
        bad.free()
        dd.coeff_pos[5] = keep
        good = d.build(2, sp.desc)
        d.run(good, 2); d.sync()
        assert all(np.array_equal(g, e_) for g, e_ in zip(d.download(2, w, h, bd), exp))
        good.free()
    finally:
        d.close()


def test_one_failing_picture_in_a_pipeline_fails_only_its_own_ticket(dec):
    """a picture's error belongs to that picture.  Five pictures through a pipeline, the third with a coefficient
    position beyond its block: only its ticket reports an error, the others deliver correct pictures - although the pipeline
    builds (and checks) pictures ahead of the launches."""
    w, h, bd = 416, 240, 8
    sps, exps = [], []
    for k in range(5):
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 2, seed=1500 + k, log2_max_tb_size=3))
        exp = pyoracle.alloc_planes(w, h, bd)
        pyoracle.reconstruct(sp.desc, sp.order, {}, exp)
        sps.append(sp); exps.append(exp)
    sps[2].d.coeff_pos[7] = 61000
    for host_scan in (0, 1):
        import os
        if host_scan:
            os.environ["DE265HIP_HOST_SCAN"] = "1"
        d = backend.Decoder()
        os.environ.pop("DE265HIP_HOST_SCAN", None)
        try:
            for k in range(5):
                d.dpb_alloc(k, w, h, bd)
            pipe = backend.Pipeline(d, 3)
            pins = [backend.PinnedPlanes(w, h, bd) for _ in range(5)]
            tickets = [pipe.submit_desc(k, sps[k].desc, pins[k]) for k in range(5)]
            for k, t in enumerate(tickets):
                if k == 2:
                    with pytest.raises(backend.De265HipError) as e:
                        pipe.wait(t)
                    assert e.value.code == _abi.ERROR_DECODING
                else:
                    pipe.wait(t)
                    assert all(np.array_equal(g, e_) for g, e_ in zip(pins[k].planes, exps[k])), (host_scan, k)
            try:
                pipe.drain()                              # (the failed picture's word may be reported once more here)
            except backend.De265HipError as e2:
                assert e2.code == _abi.ERROR_DECODING
            pipe.close()
            for pn in pins:
                pn.free()
        finally:
            d.close()


# decode order of two hierarchical GOPs: (slice type, reference slots, destination slot).  Pictures 4 / 5, 6 and 7 / 8 of the
# first GOP do not depend on each other; the second GOP re-uses the slots of the first while its last pictures still read them
_DAG = [(2, [], 0), (1, [0], 1), (0, [0, 1], 2), (0, [0, 2], 3), (0, [0, 3], 4), (0, [3, 2], 5), (0, [2, 1], 6), (0, [2, 6], 7), (0, [6, 1], 8),
        (2, [], 0), (1, [0], 3), (0, [0, 3], 2), (0, [0, 2], 4), (0, [2, 3], 5), (0, [4, 5], 1), (1, [1], 6)]


_DAG_CACHE = {}


def _dag_pictures(w, h, bd):
    """The DAG's pictures and what the oracle makes of them, decoded one after the other (once per session)."""
    if (w, h, bd) not in _DAG_CACHE:
        sps, exp, planes = [], [], {}
        for k, (st, refs, dst) in enumerate(_DAG):
            over = dict(ref_slots=refs) if refs else {}
            sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=7100 + k, intra_pct=25, n_slices=2, weighted_pred=k & 1, **over))
            out = pysynth.fill_planes(w, h, bd, 50 + k)            # (every sample of a picture is covered by a PU, an intra TU or a PCM block: what it starts from does not show)
            pyoracle.reconstruct(sp.desc, sp.order, {s: planes[s] for s in refs}, out)
            planes[dst] = out
            sps.append(sp); exp.append([p.copy() for p in out])
        _DAG_CACHE[(w, h, bd)] = (sps, exp)
    return _DAG_CACHE[(w, h, bd)]


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_lanes_overlap_independent_pictures_and_keep_the_result(lanes):
    """Picture-level concurrency inside one decoder (de265hip_decoder_set_lanes): the pictures of a reference DAG are launched in
    decode order without any host-side wait in between, each copied out asynchronously right behind its launch; every picture
    must come out as the oracle decodes it one after the other - whatever overlaps on the device.  Covers read-after-write
    (references decoded on another lane), write-after-read and write-after-write on re-used slots, and the copy-out.
    (Full HD pictures, and the all-intra one queued several times first: the device is still busy with it when the host has
    queued the others, so a missing wait shows - checked by taking the waits out.)"""
    w, h, bd = 1920, 1080, 10
    d = backend.Decoder()
    try:
        for bad in (0, 5, -1):
            with pytest.raises(backend.De265HipError) as e:
                d.set_lanes(bad)
            assert e.value.code == _abi.ERROR_PARAMETER_OUT_OF_RANGE
        d.set_lanes(lanes)
        sps, exp = _dag_pictures(w, h, bd)
        for s in set(x[2] for x in _DAG):
            d.dpb_alloc(s, w, h, bd)
        pics = [d.build(dst, sp.desc) for sp, (_, _, dst) in zip(sps, _DAG)]
        for rep in range(2):                                   # (the second pass re-runs the same picture objects: the replay of bench.py)
            handles = []
            for _ in range(6):                                  # (the all-intra picture a few times: the device falls behind the host)
                d.run(pics[0], 2)
            for pic, (_, _, dst) in zip(pics, _DAG):
                d.run(pic, 2)
                handles.append(d.download_async(dst, w, h, bd))
            d.sync()
            for k, hd in enumerate(handles):
                got = hd.wait()
                for c in range(3):
                    bad = np.argwhere(got[c] != exp[k][c])
                    assert bad.size == 0, "lanes %d pass %d picture %d comp %d: %d mismatches, first at %s" % (lanes, rep, k, c, len(bad), tuple(bad[0]))
                hd.free()
        for pic in pics:
            pic.free()
    finally:
        d.close()


@pytest.mark.parametrize("seed", range(8))
def test_monochrome_intra_pictures_against_the_oracle(seed):
    """chroma_format_idc 0: intra pictures (the kind the reference has a defined result for), every stage, every bit depth, with
    tiles / slices / PCM / transform skip / bypass / scaling lists / implicit RDPCM / rotation drawn at random.  The two chroma
    planes of such a picture are empty on both sides; a TU record with c_idx > 0 is out of range."""
    rng = np.random.default_rng([seed, 0, 78])
    w, h = [(208, 120), (352, 288), (416, 240), (832, 480)][seed % 4]
    bd = [8, 10, 12, 9][seed % 4]
    over = dict(monochrome=1, tskip_pct=int(rng.integers(0, 50)), bypass_pct=int(rng.integers(0, 20)), pcm_pct=int(rng.integers(0, 15)),
                implicit_rdpcm=int(rng.integers(0, 2)), rotation=int(rng.integers(0, 2)), log2_max_tskip_size=int(rng.integers(2, 6)),
                intra_smoothing_disabled=int(rng.integers(0, 2)), scaling_list=int(rng.integers(0, 3)), n_slices=int(rng.integers(1, 4)),
                lf_across_slices_pct=50, log2_ctb_size=int(rng.choice([4, 5, 6])), big_coeff_pct=int(rng.choice([0, 3])),
                constrained_intra_pred=int(rng.integers(0, 2)), pcm_loop_filter_disable=int(rng.integers(0, 2)),
                tile_cols=int(rng.integers(1, 4)), tile_rows=int(rng.integers(1, 3)), lf_across_tiles=int(rng.integers(0, 2)))
    if over["log2_ctb_size"] == 4:
        over["log2_max_tb_size"] = 4
    sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 2, seed=5200 + seed, **over))
    assert sp.d.params.chroma_format_idc == 0 and sp.d.n_pus == 0
    d = backend.Decoder()
    try:
        d.dpb_alloc(2, w, h, bd, chroma_format=0)
        pic = d.build(2, sp.desc)
        for stage in (0, 1, 2):
            init = pysynth.fill_planes(w, h, bd, 999, 0)
            assert init[1].size == 0 and init[2].size == 0
            exp = [p.copy() for p in init]
            pyoracle.reconstruct(sp.desc, sp.order, {}, exp, last_stage=stage)
            d.upload(2, init); d.run(pic, stage); d.sync()
            got = d.download(2, w, h, bd)
            assert got[1].shape == (0, 0) and got[2].shape == (0, 0)
            bad = np.argwhere(got[0] != exp[0])
            assert bad.size == 0, "seed %d stage %d: %d mismatches, first at %s (%s)" % (seed, stage, len(bad), tuple(bad[0]), over)
        pic.free()
        # a chroma TU in a monochrome picture is refused
        sp.d.tus[0].c_idx = 1
        with pytest.raises(backend.De265HipError) as e:
            bad = d.build(2, sp.desc)
            try:
                d.run(bad, 2)
            finally:
                bad.free()
        assert e.value.code == _abi.ERROR_PARAMETER_OUT_OF_RANGE
    finally:
        d.close()


@pytest.mark.parametrize("cf", [2, 3])
@pytest.mark.parametrize("seed", range(6))
def test_range_extension_pictures_against_the_oracle(seed, cf):
    """SURVEY 8 f4 on random small pictures: 4:2:2 / 4:4:4, every stage, with the range-extension sample tools drawn at random
    (cross-component prediction in 4:4:4, implicit / explicit RDPCM, rotation, transform skip up to 32x32, intra smoothing off,
    high-precision offsets), next to tiles / slices / PCM / bypass / scaling lists / weighted prediction."""
    rng = np.random.default_rng([seed, cf, 77])
    w, h = [(208, 120), (352, 288), (416, 240)][seed % 3]
    bd = int(rng.choice([8, 10, 12]))
    over = dict(chroma_format=cf, tskip_pct=int(rng.integers(0, 50)), bypass_pct=int(rng.integers(0, 20)), pcm_pct=int(rng.integers(0, 10)),
                implicit_rdpcm=int(rng.integers(0, 2)), explicit_rdpcm_pct=int(rng.choice([0, 50])), rotation=int(rng.integers(0, 2)),
                log2_max_tskip_size=int(rng.integers(2, 6)), intra_smoothing_disabled=int(rng.integers(0, 2)),
                high_precision_offsets=int(rng.integers(0, 2)), weighted_pred=int(rng.integers(0, 2)),
                cross_component_pct=int(rng.choice([0, 60])) if cf == 3 else 0, scaling_list=int(rng.integers(0, 2)),
                n_slices=int(rng.integers(1, 4)), lf_across_slices_pct=50, log2_ctb_size=int(rng.choice([4, 5, 6])),
                intra_pct=int(rng.choice([15, 50])), big_coeff_pct=int(rng.choice([0, 3])))
    if over["log2_ctb_size"] == 4:
        over["log2_max_tb_size"] = 4
    cfg = pysynth.default_config(w, h, bd, seed % 3, seed=5000 + 10 * seed + cf, **over)
    sp = pysynth.SynthPicture(cfg)
    d = backend.Decoder()
    try:
        refs = {0: pysynth.fill_planes(w, h, bd, 100 + seed, cf), 1: pysynth.fill_planes(w, h, bd, 200 + seed, cf)}
        for slot, pl in refs.items():
            d.dpb_alloc(slot, w, h, bd, chroma_format=cf); d.upload(slot, pl)
        pic = d.build(2, sp.desc)
        for stage in (0, 1, 2):
            init = pysynth.fill_planes(w, h, bd, 999, cf)
            exp = [p.copy() for p in init]
            pyoracle.reconstruct(sp.desc, sp.order, refs, exp, last_stage=stage)
            d.upload(2, init); d.run(pic, stage); d.sync()
            got = d.download(2, w, h, bd)
            for c in range(3):
                bad = np.argwhere(got[c] != exp[c])
                assert bad.size == 0, "cf %d seed %d stage %d comp %d: %d mismatches, first at %s (%s)" % (cf, seed, stage, c, len(bad), tuple(bad[0]), over)
        pic.free()
    finally:
        d.close()


@pytest.mark.parametrize("bd,cf", [(8, 1), (10, 1), (10, 3)])
def test_dpb_fill_makes_the_unavailable_reference_picture_of_libde265(dec, bd, cf):
    """de265hip_dpb_fill (generate_unavailable_reference_picture, decctx.cc:1408-1434: fill_image(1 << (bitDepth - 1)) on the
    host): the slot holds the constant picture, a B picture that predicts from it equals the oracle's with that reference,
    values beyond the bit depth and unallocated slots are refused."""
    w, h = 416, 240
    grey = 1 << (bd - 1)
    dec.dpb_alloc(0, w, h, bd, chroma_format=cf)
    dec.upload(0, pysynth.fill_planes(w, h, bd, 41, chroma_format=cf) if cf != 1 else pysynth.fill_planes(w, h, bd, 41))
    dec.fill(0, grey, grey, grey)
    got = dec.download(0, w, h, bd)
    assert all((p == grey).all() for p in got)
    dec.fill(0, 0, (1 << bd) - 1, 3)
    got = dec.download(0, w, h, bd)
    assert (got[0] == 0).all() and (got[1] == (1 << bd) - 1).all() and (got[2] == 3).all()
    with pytest.raises(backend.De265HipError):
        dec.fill(0, 1 << bd, 0, 0)
    d2 = backend.Decoder()
    try:
        with pytest.raises(backend.De265HipError):
            d2.fill(0, grey, grey, grey)                 # never allocated
    finally:
        d2.close()
    if cf != 1:
        return
    # a picture predicted from the synthesised reference
    sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, 0, seed=7100 + bd, intra_pct=10))
    refs = {}
    for slot in (0, 1):
        dec.dpb_alloc(slot, w, h, bd)
        dec.fill(slot, grey, grey, grey)
        refs[slot] = [np.full_like(p, grey) for p in pysynth.fill_planes(w, h, bd, 1)]
    init = pysynth.fill_planes(w, h, bd, 999)
    exp = [p.copy() for p in init]
    pyoracle.reconstruct(sp.desc, sp.order, refs, exp)
    dec.dpb_alloc(2, w, h, bd)
    dec.upload(2, init)
    pic = dec.build(2, sp.desc)
    try:
        dec.run(pic, _abi.STAGE_FINAL)
        dec.sync()
        got = dec.download(2, w, h, bd)
        for c in range(3):
            assert np.array_equal(got[c], exp[c]), "component %d differs from the oracle" % c
    finally:
        pic.free()
