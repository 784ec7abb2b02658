"""The round-4 intra scan (scan_core.h: per-TU, per-CTB and per-run passes, kernels on the device) against the round-3 host
scan (host.hip) on random pictures: both build the run-side structures of a picture - runs, their TU lists in chain order,
producers, mailbox segments with ready / need epochs, level-0 task lists, ticket slots - and the two results must be the
same structure (tools/scan_canon.py compares by content: the layouts differ).

CPU part (this file without the gpu marker): the passes compiled for the host (the CPU rehearsal, de265hip_debug_build_host_only_ex
mode 2) against the host scan (mode 0), no GPU involved.  GPU part: the kernels' output read back from the arena against the
rehearsal's."""
import os

import numpy as np
import pytest

import pysynth
import scan_canon
from libde265_amd import backend


def small_config(rng, it):
    log2_ctb = int(rng.choice([4, 5, 6]))
    w = int(rng.integers(1, 26)) * 8 if it % 7 else int(rng.integers(26, 80)) * 8
    h = int(rng.integers(1, 18)) * 8 if it % 7 else int(rng.integers(18, 48)) * 8
    bd = int(rng.choice([8, 10]))
    st = int(rng.choice([0, 1, 2]))
    cols = int(rng.integers(1, 3)) if w >= 128 else 1
    rows = int(rng.integers(1, 3)) if h >= 128 else 1
    cols, rows = min(cols, -(-w // (1 << log2_ctb))), min(rows, -(-h // (1 << log2_ctb)))
    over = dict(log2_ctb_size=log2_ctb, log2_max_tb_size=min(5, log2_ctb), log2_min_tb_size=int(rng.choice([2, 2, 3])),
                intra_pct=int(rng.choice([5, 15, 50, 100])), tskip_pct=int(rng.choice([0, 30])), bypass_pct=int(rng.choice([0, 10])),
                pcm_pct=int(rng.choice([0, 20])), constrained_intra_pred=int(rng.integers(0, 2)),
                strong_intra_smoothing=int(rng.integers(0, 2)), n_slices=int(rng.integers(1, 4)), tile_cols=cols, tile_rows=rows,
                slice_per_tile=int(rng.integers(0, 2)), split_bias=int(rng.choice([0, 50, 100])), cbf_pct=int(rng.choice([0, 60, 100])))
    return w, h, bd, st, over


def rext_config(rng, it):
    cf = int(rng.choice([2, 3, 0]))
    w, h = [(208, 120), (352, 288), (416, 240), (136, 72)][it % 4]
    over = dict(tskip_pct=int(rng.integers(0, 50)), bypass_pct=int(rng.integers(0, 20)), pcm_pct=int(rng.integers(0, 10)),
                implicit_rdpcm=int(rng.integers(0, 2)), rotation=int(rng.integers(0, 2)), log2_max_tskip_size=int(rng.integers(2, 6)),
                intra_smoothing_disabled=int(rng.integers(0, 2)), n_slices=int(rng.integers(1, 4)), log2_ctb_size=int(rng.choice([4, 5, 6])),
                constrained_intra_pred=int(rng.integers(0, 2)), big_coeff_pct=int(rng.choice([0, 3])))
    if cf == 0:
        over.update(monochrome=1)
        st = 2
    else:
        over.update(chroma_format=cf, explicit_rdpcm_pct=int(rng.choice([0, 50])), cross_component_pct=int(rng.choice([0, 60])) if cf == 3 else 0,
                    intra_pct=int(rng.choice([15, 50, 100])))
        st = int(rng.choice([0, 1, 2]))
    if over["log2_ctb_size"] == 4:
        over["log2_max_tb_size"] = 4
    return w, h, int(rng.choice([8, 10])), st, over


def compare(desc, what):
    L = backend.lib()
    h0 = scan_canon.build_dry(desc, 0)
    h2 = scan_canon.build_dry(desc, 2)
    try:
        a, b = scan_canon.canon(h0), scan_canon.canon(h2)
        d = scan_canon.diff(a, b)
        assert d is None, "%s: host scan vs passes: %s" % (what, d)
        return a
    finally:
        L.de265hip_picture_free(h0); L.de265hip_picture_free(h2)


def test_passes_equal_the_host_scan_on_small_pictures():
    rng = np.random.default_rng(20261005)
    n_runs = n_mb = 0
    for it in range(int(os.environ.get("DE265HIP_TEST_SCAN_SMALL", "120"))):
        w, h, bd, st, over = small_config(rng, it)
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=31000 + it, **over))
        a = compare(sp.desc, "small %d: %dx%d st=%d %r" % (it, w, h, st, over))
        n_runs += a["counts"]["n_runs"]; n_mb += sum(1 for r in a["runs"].values() if "reads" in r)
        sp.close()
    assert n_runs > 1000 and n_mb > 20          # (the sweep does exercise runs and mailbox readers)


def test_passes_equal_the_host_scan_on_other_chroma_formats_and_tools():
    rng = np.random.default_rng(20261006)
    for it in range(int(os.environ.get("DE265HIP_TEST_SCAN_REXT", "60"))):
        w, h, bd, st, over = rext_config(rng, it)
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=32000 + it, **over))
        compare(sp.desc, "rext %d: %dx%d st=%d %r" % (it, w, h, st, over))
        sp.close()


def test_passes_equal_the_host_scan_on_everyday_sizes():
    """Full HD all-intra (dense runs, phased mailboxes on every CTB) and a B picture with isolated intra CUs (micro and front runs)."""
    for st, w, h, bd, over in ((2, 1920, 1080, 10, {}), (0, 1920, 1080, 8, dict(n_slices=3)), (2, 1280, 720, 8, dict(tile_cols=3, tile_rows=2, log2_ctb_size=5))):
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=33000 + st + w, **over))
        a = compare(sp.desc, "%dx%d st=%d" % (w, h, st))
        assert a["counts"]["n_runs"] > 300
        if st == 0:
            assert a["counts"]["n_front"] > 100
        sp.close()


def test_schedule_switches_mean_the_same_in_both_scans(monkeypatch):
    """every environment switch that shapes the runs reaches the passes as it reaches the host scan"""
    sp = pysynth.SynthPicture(pysynth.default_config(832, 480, 10, 0, seed=34001, intra_pct=40, n_slices=2))
    spi = pysynth.SynthPicture(pysynth.default_config(832, 480, 8, 2, seed=34002))
    for env in ({}, {"DE265HIP_NO_MODE_DEPS": "1"}, {"DE265HIP_NO_MERGE": "1"}, {"DE265HIP_NO_MAILBOX": "1"}, {"DE265HIP_NO_MB_PHASES": "1"},
                {"DE265HIP_NO_MICRO": "1"}, {"DE265HIP_NO_DENSE": "1"}, {"DE265HIP_MICRO16": "0"}, {"DE265HIP_NO_FRONT": "1"},
                {"DE265HIP_MICRO_TUS": "8"}, {"DE265HIP_TEST_DROP_PRODUCER": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for s_, name in ((sp, "B"), (spi, "I")):
            a = compare(s_.desc, "%s picture with %r" % (name, env))
            if "DE265HIP_TEST_DROP_PRODUCER" in env:
                assert len(a["no_ticket"]) == 1
        for k in env:
            monkeypatch.delenv(k)


def test_malformed_records_are_refused_by_the_passes():
    from libde265_amd import _abi
    import ctypes as C
    sp = pysynth.SynthPicture(pysynth.default_config(416, 240, 8, 2, seed=35001))
    L = backend.lib()
    d = sp.d
    keep = (d.tus[5].x0, d.tus[5].log2_size)
    for field, val in (("x0", 4000), ("log2_size", 7), ("c_idx", 3)):
        old = getattr(d.tus[5], field)
        setattr(d.tus[5], field, val)
        for mode in (0, 2):
            assert L.de265hip_debug_build_host_only_ex(sp.desc, 1, mode, None) == _abi.ERROR_PARAMETER_OUT_OF_RANGE, (field, mode)
        setattr(d.tus[5], field, old)
    # the records of a CTB must be contiguous: moving one TU record far away breaks that (the host scan does not look at it:
    # it trusts the decode order; the passes need the property and check it)
    assert (d.tus[5].x0, d.tus[5].log2_size) == keep
    del C


# ---------------------------------------------------------------- the kernels against the rehearsal
def _device_vs_rehearsal(dec, desc, what):
    L = backend.lib()
    pic = dec.build(2, desc)
    h2 = scan_canon.build_dry(desc, 2)
    try:
        a, b = scan_canon.canon(pic._h), scan_canon.canon(h2)
        d = scan_canon.diff(a, b)
        assert d is None, "%s: kernels vs CPU rehearsal: %s" % (what, d)
        return a
    finally:
        pic.free(); L.de265hip_picture_free(h2)


@pytest.mark.gpu
def test_kernels_equal_the_rehearsal():
    """k_scan.hip's kernels leave in the arena what the same passes leave when they run as loops on the host (which the CPU
    tests above hold to the round-3 host scan): small pictures of every kind, the other chroma formats, Full HD, 4K."""
    assert backend.device_count() > 0
    dec = backend.Decoder()
    try:
        rng = np.random.default_rng(20261007)
        for it in range(40):
            w, h, bd, st, over = small_config(rng, it)
            sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=36000 + it, **over))
            dec.dpb_alloc(2, w, h, bd)
            _device_vs_rehearsal(dec, sp.desc, "small %d: %dx%d st=%d %r" % (it, w, h, st, over))
            sp.close()
        for it in range(24):
            w, h, bd, st, over = rext_config(rng, it)
            sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=37000 + it, **over))
            dec.dpb_alloc(2, w, h, bd, chroma_format=sp.d.params.chroma_format_idc)
            _device_vs_rehearsal(dec, sp.desc, "rext %d: %dx%d st=%d %r" % (it, w, h, st, over))
            sp.close()
        for st, w, h, bd in ((2, 1920, 1080, 10), (0, 1920, 1080, 8), (2, 3840, 2160, 10), (0, 3840, 2160, 10)):
            sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=38000 + st + w))
            dec.dpb_alloc(2, w, h, bd)
            a = _device_vs_rehearsal(dec, sp.desc, "%dx%d st=%d" % (w, h, st))
            assert a["counts"]["n_runs"] > 1000
            sp.close()
    finally:
        dec.close()
