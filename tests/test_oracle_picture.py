"""Picture-level oracle and host-logic checks (no GPU): order equivalence, stage
isolation flags, in-loop filter properties, edge-flag derivation (oracle + the
product's host helper) against the generator's own marks."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import pyoracle
import pysynth
from libde265_amd import _abi, backend

L = pyoracle.lib()


def make(w, h, bd, st, seed, **over):
    sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=seed, **over))
    refs = {0: pysynth.fill_planes(w, h, bd, 100 + seed), 1: pysynth.fill_planes(w, h, bd, 200 + seed)}
    return sp, refs


def recon(sp, refs, stage=_abi.STAGE_FINAL, order="decode", init_seed=7):
    P = sp.d.params
    out = pysynth.fill_planes(P.width, P.height, P.bit_depth_luma, init_seed)
    pyoracle.reconstruct(sp.desc, sp.order if order == "decode" else None, refs, out, last_stage=stage)
    return out


def digest(planes):
    m = hashlib.md5()
    for p in planes:
        m.update(np.ascontiguousarray(p).tobytes())
    return m.hexdigest()


@pytest.mark.parametrize("bd,st", [(8, 0), (10, 0), (8, 1), (10, 2)])
def test_phase_order_equals_decode_order(bd, st):
    """MC -> PCM -> TUs (the device's phase order) gives the reference's interleaved decode order result."""
    sp, refs = make(416, 240, bd, st, 3, pcm_pct=10, tskip_pct=20, bypass_pct=5, n_slices=2)
    assert digest(recon(sp, refs, order="decode")) == digest(recon(sp, refs, order="phase"))


def test_generator_is_deterministic_and_covers_features():
    a, _ = make(352, 288, 8, 0, 9, pcm_pct=40, tskip_pct=30, bypass_pct=10)
    b, _ = make(352, 288, 8, 0, 9, pcm_pct=40, tskip_pct=30, bypass_pct=10)
    da, db = a.d, b.d
    assert da.n_tus == db.n_tus and da.n_pus == db.n_pus and da.n_coeffs == db.n_coeffs
    ta = np.ctypeslib.as_array(C.cast(da.tus, C.POINTER(C.c_uint8)), shape=(da.n_tus * C.sizeof(_abi.TU),))
    tb = np.ctypeslib.as_array(C.cast(db.tus, C.POINTER(C.c_uint8)), shape=(db.n_tus * C.sizeof(_abi.TU),))
    assert np.array_equal(ta, tb)
    flags = np.array([da.tus[i].flags for i in range(da.n_tus)])
    sizes = np.array([da.tus[i].log2_size for i in range(da.n_tus)])
    assert (flags & _abi.TU_INTRA).any() and (~flags & _abi.TU_INTRA).any()
    assert (flags & _abi.TU_TSKIP).any() and (flags & _abi.TU_BYPASS).any() and da.n_pcms > 0
    assert set(sizes) == {2, 3, 4, 5}
    pf = np.array([da.pus[i].pred_flag for i in range(da.n_pus)])
    assert set(pf) == {1, 2, 3}
    widths = {da.pus[i].w for i in range(da.n_pus)}
    assert {4, 8, 12, 16, 32, 64} <= widths or {8, 16, 32, 64} <= widths


def test_monochrome_intra_pictures_and_what_is_refused():
    """chroma_format_idc 0: the generator emits luma records only, PCM blocks carry n*n samples, both checkers decode it into a
    luma plane next to two EMPTY chroma planes, and a monochrome picture with prediction units is refused by both (the
    reference's inter path addresses chroma planes whatever the format, motion.cc:296-305: no defined result)."""
    w, h = 208, 120
    sp = pysynth.SynthPicture(pysynth.default_config(w, h, 10, 2, seed=31, monochrome=1, pcm_pct=20, tskip_pct=20, n_slices=2))
    d = sp.d
    assert d.params.chroma_format_idc == 0 and d.n_pus == 0 and d.n_pcms > 0
    assert all(d.tus[i].c_idx == 0 for i in range(d.n_tus))
    assert d.n_pcm_samples == sum((1 << d.pcms[i].log2_cb_size) ** 2 for i in range(d.n_pcms))
    out = pyoracle.alloc_planes(w, h, 10, chroma_format=0)
    assert out[1].shape == (0, 0) and out[2].shape == (0, 0)
    pre = [p.copy() for p in out]
    pyoracle.reconstruct(sp.desc, sp.order, {}, pre, last_stage=_abi.STAGE_PREFILTER)
    fin = [p.copy() for p in out]
    pyoracle.reconstruct(sp.desc, sp.order, {}, fin)
    assert not np.array_equal(pre[0], fin[0]) and fin[0].max() < 1024
    # the product's host stage (no device needed) accepts it ...
    assert backend.lib().de265hip_debug_build_host_only(sp.desc, 1) == 0
    # ... and both refuse monochrome with prediction units
    sp2 = pysynth.SynthPicture(pysynth.default_config(w, h, 8, 0, seed=32, monochrome=1))
    assert sp2.d.n_pus > 0
    with pytest.raises(RuntimeError):
        pyoracle.reconstruct(sp2.desc, sp2.order, {}, pyoracle.alloc_planes(w, h, 8, chroma_format=0))
    assert backend.lib().de265hip_debug_build_host_only(sp2.desc, 1) == _abi.ERROR_NOT_IMPLEMENTED
    # a chroma TU record in a monochrome picture is out of range
    d.tus[0].c_idx = 1
    assert backend.lib().de265hip_debug_build_host_only(sp.desc, 1) == _abi.ERROR_PARAMETER_OUT_OF_RANGE


def test_host_stage_survives_corrupted_descriptions():
    """tools/fuzz_desc.py: random corruptions of valid picture descriptions - wild values and values that pass every range
    check (moved, resized, duplicated TUs, arbitrary flags and modes) - come back from the host stage as an error code or as a
    normal build, never as a crash.  (The same under an address-sanitizer build of the host code: see the tool's header;
    1 800 descriptions without a report.)"""
    import fuzz_desc
    codes = fuzz_desc.run(20261004, 120)
    assert set(codes) <= {0, _abi.ERROR_PARAMETER_OUT_OF_RANGE, _abi.ERROR_NOT_IMPLEMENTED, _abi.ERROR_OUT_OF_MEMORY}, codes
    assert codes.get(0, 0) > 10 and codes.get(_abi.ERROR_PARAMETER_OUT_OF_RANGE, 0) > 10


def test_host_stage_cr_shortcut_and_its_fallback(monkeypatch):
    """The TU scan of de265hip_picture_build takes a Cr intra TU's availability / levels / run from its Cb twin; a descriptor
    without that pattern (here: a Cr TU whose mode differs from its Cb TU's, and a Cb TU whose Cr TU is missing) is built the
    long way.  Either way the uploaded arena is the one the long way builds (FNV hash of everything the device would get)."""
    Lh = backend.lib()

    def both(desc):
        monkeypatch.delenv("DE265HIP_NO_CR_MIRROR", raising=False)
        rc_a = Lh.de265hip_debug_build_host_only(desc, 1); a = Lh.de265hip_debug_last_build_hash()
        monkeypatch.setenv("DE265HIP_NO_CR_MIRROR", "1")
        rc_b = Lh.de265hip_debug_build_host_only(desc, 1); b = Lh.de265hip_debug_last_build_hash()
        monkeypatch.delenv("DE265HIP_NO_CR_MIRROR")
        return rc_a, a, rc_b, b

    for st, cf in ((2, 1), (0, 1), (2, 2), (0, 3)):
        sp = pysynth.SynthPicture(pysynth.default_config(416, 240, 8, st, seed=90 + st + cf, chroma_format=cf, tskip_pct=20, n_slices=2))
        d = sp.d
        rc_a, a, rc_b, b = both(sp.desc)
        assert rc_a == 0 and rc_b == 0 and a == b
        cr = [i for i in range(d.n_tus) if d.tus[i].c_idx == 2 and (d.tus[i].flags & _abi.TU_INTRA)]
        assert cr
        k = cr[len(cr) // 2]
        d.tus[k].intra_mode = (d.tus[k].intra_mode + 7) % 35          # no longer its Cb twin's mode
        rc_a, a2, rc_b, b2 = both(sp.desc)
        assert rc_a == 0 and rc_b == 0 and a2 == b2 and a2 != a
        d.tus[k].intra_mode = (d.tus[k].intra_mode - 7) % 35
        d.tus[k].flags &= ~_abi.TU_INTRA & 0xFF                        # the Cb TU before it is left alone
        rc_a, a3, rc_b, b3 = both(sp.desc)
        assert rc_a == rc_b and a3 == b3
        sp.close()


def test_disable_flags_and_stages():
    sp, refs = make(416, 240, 8, 0, 5)
    pre = recon(sp, refs, _abi.STAGE_PREFILTER)
    dbk = recon(sp, refs, _abi.STAGE_DEBLOCKED)
    fin = recon(sp, refs, _abi.STAGE_FINAL)
    assert digest(pre) != digest(dbk) != digest(fin)
    sp.d.params.disable_sao = 1
    assert digest(recon(sp, refs)) == digest(dbk)
    sp.d.params.disable_deblocking = 1
    assert digest(recon(sp, refs)) == digest(pre)


def test_deblock_changes_only_near_edges_and_sao_is_bounded():
    sp, refs = make(416, 240, 10, 0, 6)
    pre = recon(sp, refs, _abi.STAGE_PREFILTER)
    dbk = recon(sp, refs, _abi.STAGE_DEBLOCKED)
    fin = recon(sp, refs, _abi.STAGE_FINAL)
    ch = np.argwhere(pre[0] != dbk[0])
    assert len(ch) > 0
    # luma deblocking touches at most 3 samples each side of the 8x8 grid
    dist = np.minimum(np.minimum(ch % 8, 7 - ch % 8).min(axis=1), 3)
    assert (np.minimum(ch[:, 0] % 8, 7 - ch[:, 0] % 8) <= 2).__or__(np.minimum(ch[:, 1] % 8, 7 - ch[:, 1] % 8) <= 2).all()
    # chroma: only p0/q0 of the 8-sample chroma grid
    cc = np.argwhere(pre[1] != dbk[1])
    assert ((cc[:, 0] % 8 == 0) | (cc[:, 0] % 8 == 7) | (cc[:, 1] % 8 == 0) | (cc[:, 1] % 8 == 7)).all()
    # SAO moves a sample by at most the largest offset (31 at 10 bit)
    assert np.abs(fin[0].astype(int) - dbk[0].astype(int)).max() <= 31
    del dist


def test_flat_picture_is_a_fixed_point_of_the_filters():
    """All-intra DC picture with no residual: deblocking decisions see zero gradients, SAO edge sees no edges."""
    sp, refs = make(256, 128, 8, 2, 8, cbf_pct=0)
    d = sp.d
    for i in range(d.n_tus):
        d.tus[i].intra_mode = 1
    for i in range(d.n_ctbs):                 # edge offset only (band offsets would shift a flat picture)
        t = d.ctbs[i].sao_type_idx
        d.ctbs[i].sao_type_idx = sum((2 if (t >> (2 * c)) & 3 else 0) << (2 * c) for c in range(3))
    out = recon(sp, refs)
    assert all((p == 128).all() for p in out)


def test_sao_band_numpy_model():
    sp, refs = make(128, 128, 8, 2, 4, log2_ctb_size=5, deblocking=0)
    d = sp.d
    dbk = recon(sp, refs, _abi.STAGE_DEBLOCKED)
    for i in range(d.n_ctbs):
        d.ctbs[i].sao_type_idx = 1 | (1 << 2) | (1 << 4)
    for s in range(d.n_slices):
        d.slices[s].slice_sao_luma_flag = d.slices[s].slice_sao_chroma_flag = 1
    fin = recon(sp, refs)
    for c, (ctb, plane_in, plane_out) in enumerate([(32, dbk[0], fin[0]), (16, dbk[1], fin[1]), (16, dbk[2], fin[2])]):
        exp = plane_in.astype(int).copy()
        for cy in range(128 // 32):
            for cx in range(128 // 32):
                ci = d.ctbs[cx + cy * 4]
                blk = plane_in[cy * ctb:(cy + 1) * ctb, cx * ctb:(cx + 1) * ctb].astype(int)
                k = ((blk >> 3) - ci.sao_band_position[c]) & 31
                off = np.zeros_like(blk)
                for j in range(4):
                    off[k == j] = ci.sao_offset_val[c][j]
                exp[cy * ctb:(cy + 1) * ctb, cx * ctb:(cx + 1) * ctb] = np.clip(blk + off, 0, 255)
        assert np.array_equal(plane_out, exp), c


@pytest.mark.parametrize("over", [dict(), dict(n_slices=4, lf_across_slices_pct=0),
                                  dict(tile_cols=3, tile_rows=2, slice_per_tile=1, lf_across_tiles=0,
                                       lf_across_slices_pct=0),
                                  dict(tile_cols=2, tile_rows=2, lf_across_tiles=0), dict(n_slices=3)])
def test_edge_flag_derivation_three_ways(over):
    """a11: generator marks == oracle derive_edgeFlags restatement == product host helper."""
    sp, _ = make(640, 384, 8, 0, 21, pcm_pct=5, **over)
    d = sp.d
    cb_log2, cb_part, tu_split, noedge = sp.structure()
    want = sp.blk_flags().ravel()
    got_o = noedge.copy()
    assert L.oracle_derive_edge_flags(C.byref(d.params), d.slices, d.n_slices, d.ctbs, cb_log2.ctypes.data,
                                      cb_part.ctypes.data, tu_split.ctypes.data, got_o.ctypes.data) == 0
    assert np.array_equal(got_o, want)
    got_h = noedge.copy()
    backend.derive_edge_flags(d.params, d.slices, d.n_slices, d.ctbs, cb_log2, cb_part, tu_split, got_h)
    assert np.array_equal(got_h, want)
    assert (want & 0xF0).any()


def test_bs_values_and_intra_rule():
    sp, _ = make(416, 240, 8, 0, 12)
    d = sp.d
    w4, h4 = 104, 60
    flags = sp.blk_flags()
    for vertical in (1, 0):
        bs = np.zeros((h4, w4), np.uint8)
        L.oracle_derive_bs(sp.desc, vertical, bs.ctypes.data)
        assert set(np.unique(bs)) <= {0, 1, 2} and (bs == 2).any() and (bs == 1).any()
        mask = (_abi.BLK_EDGE_TU_V | _abi.BLK_EDGE_PB_V) if vertical else (_abi.BLK_EDGE_TU_H | _abi.BLK_EDGE_PB_H)
        ys, xs = np.nonzero(bs)
        assert ((flags[ys, xs] & mask) != 0).all()
        assert ((xs % 2 == 0).all() if vertical else (ys % 2 == 0).all())       # 8x8 grid only
        nb = flags[ys, xs - 1] if vertical else flags[ys - 1, xs]
        intra = ((flags[ys, xs] | nb) & _abi.BLK_INTRA) != 0
        assert (bs[ys, xs][intra] == 2).all() and (bs[ys, xs][~intra] < 2).all()
    del d


def test_recorder_accumulates_the_same_description():
    """de265hip_recorder_*: replaying a description call by call (TU / PU / PCM / slice / CTB / planes)
    yields a description the oracle reconstructs to the identical picture (host logic, no GPU)."""
    sp, refs = make(352, 288, 10, 0, 33, pcm_pct=15, tskip_pct=20, n_slices=2, scaling_list=1)
    d = sp.d
    sf = np.ctypeslib.as_array(d.scaling_factors, shape=(_abi.SCALING_BLOB_BYTES,)).copy()
    rec = backend.Recorder(d.params, sf)
    rec.record_desc(d)
    rd = rec.desc.contents
    assert (rd.n_tus, rd.n_pus, rd.n_pcms, rd.n_coeffs, rd.n_slices, rd.n_ctbs) == \
           (d.n_tus, d.n_pus, d.n_pcms, d.n_coeffs, d.n_slices, d.n_ctbs)
    P = d.params
    a = pysynth.fill_planes(P.width, P.height, P.bit_depth_luma, 3)
    b = [p.copy() for p in a]
    pyoracle.reconstruct(sp.desc, None, refs, a)
    pyoracle.reconstruct(rec.desc, None, refs, b)
    assert digest(a) == digest(b)
    with pytest.raises(backend.De265HipError):
        backend._chk(backend.lib().de265hip_record_ctb(rec._h, 10 ** 6, C.byref(d.ctbs[0])), "record_ctb")
    rec.free()


def test_mode_aware_dependency_table_covers_every_sample_the_predictors_read():
    """Host logic behind the run kernel's schedule (de265hip_intra_used_units): an intra TU only waits for the producers
    of the neighbour units its mode can read.  Checked against the oracle's predictors by brute force: for every size,
    mode and component type, a border sample whose change alters the prediction (random borders, and smooth ramps that
    switch the strong bilinear smoothing on) must lie in a unit the table names.  This is the check that would have
    caught the strong-smoothing decision samples p[+-32] missing from the table."""
    import ctypes as C
    from libde265_amd import backend
    import pyoracle
    L, O = backend.lib(), pyoracle.lib()
    rng = np.random.default_rng(77)
    bd = 10
    for log2 in (2, 3, 4, 5):
        nT = 1 << log2
        NB, Cc, corner = 4 * nT + 1, 2 * nT, nT >> 1
        unit_of = lambda q: (q >> 2) if q < Cc else (corner if q == Cc else corner + 1 + ((q - Cc - 1) >> 2))
        bases = [rng.integers(0, 1 << bd, NB) for _ in range(3)]
        bases.append(np.linspace(400, 460, NB).astype(np.int64))          # smooth: bilinear variant at 32x32 luma
        bases.append(np.full(NB, 512, np.int64))                          # flat
        for luma in (1, 0):
            for mode in range(35):
                u = C.c_uint64()
                assert L.de265hip_intra_used_units(log2, mode, luma, C.byref(u)) == 0
                used = u.value
                for base in bases:
                    border = base.astype(np.uint16)
                    ref = np.zeros((nT, nT), np.uint16)
                    O.oracle_intra_predict(bd, 1, ref.ctypes.data, nT, nT, 0 if luma else 1, mode,
                                           border.ctypes.data + Cc * 2)
                    for q in range(NB):
                        if (used >> unit_of(q)) & 1:
                            continue                                      # named by the table: nothing to prove
                        for delta in (37, -41, 300):
                            b2 = border.copy()
                            b2[q] = np.uint16(min(max(int(b2[q]) + delta, 0), (1 << bd) - 1))
                            out = np.zeros((nT, nT), np.uint16)
                            O.oracle_intra_predict(bd, 1, out.ctypes.data, nT, nT, 0 if luma else 1, mode,
                                                   b2.ctypes.data + Cc * 2)
                            assert np.array_equal(out, ref), \
                                "size %d mode %d luma %d: border entry %d (unit %d) changes the prediction but is not in the table %x" % (
                                    nT, mode, luma, q, unit_of(q), used)
    bad = C.c_uint64()
    assert L.de265hip_intra_used_units(6, 0, 1, C.byref(bad)) != 0
