"""GPU parity tests proper: whole-picture reconstruction through the C ABI
(de265hip_picture_build/run) against the CPU oracle on the same seeded
synthetic command buffers.  Bit-exact at every stage."""
import numpy as np
import pytest

import pyoracle
import pysynth
from libde265_amd import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    from libde265_amd import backend
    assert backend.device_count() > 0, "no GPU visible: HIP path cannot run"
    d = backend.Decoder()
    yield d
    d.close()


def run_case(dec, w, h, bd, slice_type, seed, stages=(0, 1, 2), **over):
    cfg = pysynth.default_config(w, h, bd, slice_type, seed=seed, **over)
    sp = pysynth.SynthPicture(cfg)
    refs = {0: pysynth.fill_planes(w, h, bd, 100 + seed), 1: pysynth.fill_planes(w, h, bd, 200 + seed)}
    for slot, pl in refs.items():
        dec.dpb_alloc(slot, w, h, bd)
        dec.upload(slot, pl)
    pic = dec.build(2, sp.desc)
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


def test_1080p_b_picture(dec):
    run_case(dec, 1920, 1080, 8, 0, seed=11, stages=(2,))
