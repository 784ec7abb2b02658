"""Seeded cases behind tests/golden/ref_*.json -- the fixtures GENERATED FROM THE COMPILED REFERENCE
(oracle/_ref, tools/make_ref_golden.py) -- and runners that push one case through the CPU restatement
(oracle/) or the compiled reference (oracle/_ref).  Test infrastructure.

Picture cases exercise SURVEY.md 8a rows a1-a16 at picture level (the "metadata-driven oracle" of
SURVEY.md section 4); function cases exercise the acceleration_functions slots (acceleration.h:29-201) of
rows a2-a4, a7-a9 one block at a time.  Inputs are regenerated from seeds wherever the tests run; only
digests (and three small full pictures) are stored.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import pyoracle  # noqa: E402
import pysynth  # noqa: E402

# --------------------------------------------------------------------------------------- pictures
_F = dict(tskip_pct=30, bypass_pct=8, pcm_pct=10, big_coeff_pct=3)
PICTURE_CASES = [
    # name, geometry, slice type (0 B, 1 P, 2 I), seed, generator overrides
    dict(name="qcif8_I", w=176, h=144, bd=8, st=2, seed=2001),
    dict(name="qcif10_I", w=176, h=144, bd=10, st=2, seed=2002),
    dict(name="qcif8_P", w=176, h=144, bd=8, st=1, seed=2003),
    dict(name="qcif10_B", w=176, h=144, bd=10, st=0, seed=2004),
    dict(name="cif8_I_nocbf_smooth", w=352, h=288, bd=8, st=2, seed=2005, cbf_pct=0, strong_intra_smoothing=1),
    dict(name="cif10_I_nocbf_nosmooth", w=352, h=288, bd=10, st=2, seed=2006, cbf_pct=0, strong_intra_smoothing=0),
    dict(name="cif8_I_small_tus", w=352, h=288, bd=8, st=2, seed=2007, split_bias=100, log2_max_tb_size=3),
    dict(name="cif10_I_big_tus", w=352, h=288, bd=10, st=2, seed=2008, split_bias=0),
    dict(name="cif8_P_weighted", w=352, h=288, bd=8, st=1, seed=2009, weighted_pred=1),
    dict(name="cif10_B_weighted", w=352, h=288, bd=10, st=0, seed=2010, weighted_pred=1),
    dict(name="cif8_B_constrained_intra", w=352, h=288, bd=8, st=0, seed=2011, constrained_intra_pred=1, intra_pct=40),
    dict(name="cif10_P_constrained_intra", w=352, h=288, bd=10, st=1, seed=2012, constrained_intra_pred=1, intra_pct=50),
    dict(name="cif8_B_features", w=352, h=288, bd=8, st=0, seed=2013, pcm_loop_filter_disable=1, scaling_list=1,
         n_slices=3, lf_across_slices_pct=50, **_F),
    dict(name="cif10_I_features", w=352, h=288, bd=10, st=2, seed=2014, pcm_loop_filter_disable=0, scaling_list=1,
         n_slices=3, lf_across_slices_pct=50, **_F),
    dict(name="cif10_B_features_pcmlf", w=352, h=288, bd=10, st=0, seed=2015, pcm_loop_filter_disable=1, scaling_list=0,
         n_slices=2, lf_across_slices_pct=0, weighted_pred=1, **_F),
    # the chroma edge-offset / slice-address behaviour of sao.cc:55: two and four slices,
    # slice_loop_filter_across_slices_enabled_flag 0 everywhere, SAO on
    dict(name="cif8_B_2slices_noacross", w=352, h=288, bd=8, st=0, seed=2016, n_slices=2, lf_across_slices_pct=0),
    dict(name="wvga10_I_4slices_noacross", w=832, h=480, bd=10, st=2, seed=2017, n_slices=4, lf_across_slices_pct=0),
    dict(name="wvga8_B_tiles_noacross", w=832, h=480, bd=8, st=0, seed=2018, tile_cols=3, tile_rows=2, slice_per_tile=1,
         lf_across_tiles=0, lf_across_slices_pct=0),
    dict(name="wvga10_B_tiles_one_slice", w=832, h=480, bd=10, st=0, seed=2019, tile_cols=2, tile_rows=2,
         slice_per_tile=0, lf_across_tiles=0),
    dict(name="wvga10_I_tiles_across", w=832, h=480, bd=10, st=2, seed=2020, tile_cols=3, tile_rows=2, slice_per_tile=1,
         lf_across_tiles=1, lf_across_slices_pct=100),
    dict(name="ctb16_8_B", w=208, h=120, bd=8, st=0, seed=2021, log2_ctb_size=4, log2_max_tb_size=4),
    dict(name="ctb32_10_B", w=264, h=136, bd=10, st=0, seed=2022, log2_ctb_size=5),
    dict(name="ctb16_12_B", w=208, h=120, bd=12, st=0, seed=2023, log2_ctb_size=4, log2_max_tb_size=4),
    dict(name="mintb8_9_P", w=264, h=136, bd=9, st=1, seed=2024, log2_min_tb_size=3),
    dict(name="cif8_B_nodeblock_nosao", w=352, h=288, bd=8, st=0, seed=2025, deblocking=0, sao=0),
    dict(name="cif10_B_longmv_uni", w=352, h=288, bd=10, st=0, seed=2026, mv_sigma_qpel=300, bi_pct=0, amp=1),
    dict(name="cif8_B_bi_noamp", w=352, h=288, bd=8, st=0, seed=2027, bi_pct=100, amp=0, mv_sigma_qpel=4),
    dict(name="cif10_B_qp_extremes", w=352, h=288, bd=10, st=0, seed=2028, qp_min=0, qp_max=51, big_coeff_pct=5),
    # the pictures round 1 froze as digests of its own oracle (tests/golden/oracle_digests.json, now removed):
    # same seeds, now pinned by the reference
    dict(name="r1_cif8_I", w=352, h=288, bd=8, st=2, seed=1001),
    dict(name="r1_cif8_P_weighted", w=352, h=288, bd=8, st=1, seed=1002, weighted_pred=1),
    dict(name="r1_cif8_B_slices", w=352, h=288, bd=8, st=0, seed=1003, n_slices=3, lf_across_slices_pct=0),
    dict(name="r1_wvga10_B_tiles", w=832, h=480, bd=10, st=0, seed=1004, tile_cols=3, tile_rows=2, slice_per_tile=1,
         lf_across_tiles=0),
    dict(name="r1_wvga10_I_features", w=832, h=480, bd=10, st=2, seed=1005, tskip_pct=30, pcm_pct=10, bypass_pct=5,
         pcm_loop_filter_disable=1, scaling_list=1, constrained_intra_pred=0),
    dict(name="r1_720p8_B", w=1280, h=720, bd=8, st=0, seed=1006, weighted_pred=1, big_coeff_pct=2),
    dict(name="r1_1080p10_B", w=1920, h=1080, bd=10, st=0, seed=1008),
    # BASELINE.json configs at full size
    dict(name="720p8_I_config1", w=1280, h=720, bd=8, st=2, seed=0xDE265001, log2_ctb_size=5),      # config 1 workload
    dict(name="1080p8_I_config2", w=1920, h=1080, bd=8, st=2, seed=0xDE265002),
    dict(name="1080p8_B_config3", w=1920, h=1080, bd=8, st=0, seed=0xDE265003, weighted_pred=1),
    dict(name="4k10_I_config4", w=3840, h=2160, bd=10, st=2, seed=0xDE265004),
    dict(name="4k10_B_config4", w=3840, h=2160, bd=10, st=0, seed=0xDE265005),
    # 4:2:2 / 4:4:4 at 8 and 10 bit, I and B; the range-extension sample tools the reference implements
    # (cross-component prediction, implicit / explicit RDPCM, transform-skip rotation, transform skip up to 32x32,
    # intra smoothing switched off, high-precision weighting offsets)
    dict(name="cif8_I_422", w=352, h=288, bd=8, st=2, seed=3001, chroma_format=2),
    dict(name="cif10_B_422", w=352, h=288, bd=10, st=0, seed=3002, chroma_format=2, weighted_pred=1),
    dict(name="cif8_B_444", w=352, h=288, bd=8, st=0, seed=3003, chroma_format=3),
    dict(name="cif10_I_444", w=352, h=288, bd=10, st=2, seed=3004, chroma_format=3),
    dict(name="cif8_P_422_features", w=352, h=288, bd=8, st=1, seed=3005, chroma_format=2, pcm_loop_filter_disable=1, scaling_list=1,
         n_slices=3, lf_across_slices_pct=50, **_F),
    dict(name="wvga10_B_444_tiles_slices", w=832, h=480, bd=10, st=0, seed=3006, chroma_format=3, tile_cols=3, tile_rows=2,
         slice_per_tile=1, lf_across_tiles=0, lf_across_slices_pct=0, **_F),
    dict(name="cif8_I_444_rext", w=352, h=288, bd=8, st=2, seed=3007, chroma_format=3, tskip_pct=40, bypass_pct=15, pcm_pct=5,
         implicit_rdpcm=1, rotation=1, log2_max_tskip_size=5, cross_component_pct=60, big_coeff_pct=2),
    dict(name="cif10_B_444_rext", w=352, h=288, bd=10, st=0, seed=3008, chroma_format=3, tskip_pct=40, bypass_pct=15,
         implicit_rdpcm=1, explicit_rdpcm_pct=50, rotation=1, log2_max_tskip_size=5, cross_component_pct=60, weighted_pred=1,
         high_precision_offsets=1, n_slices=2, lf_across_slices_pct=50),
    dict(name="cif10_P_422_rext", w=352, h=288, bd=10, st=1, seed=3009, chroma_format=2, tskip_pct=40, bypass_pct=15,
         implicit_rdpcm=1, explicit_rdpcm_pct=50, rotation=1, log2_max_tskip_size=4, intra_smoothing_disabled=1, intra_pct=40),
    dict(name="cif8_B_420_rext", w=352, h=288, bd=8, st=0, seed=3010, tskip_pct=40, bypass_pct=15, implicit_rdpcm=1,
         explicit_rdpcm_pct=50, rotation=1, log2_max_tskip_size=5, intra_smoothing_disabled=1, intra_pct=40, scaling_list=1),
    dict(name="1080p10_B_444_rext", w=1920, h=1080, bd=10, st=0, seed=3011, chroma_format=3, tskip_pct=20, bypass_pct=5,
         implicit_rdpcm=1, explicit_rdpcm_pct=30, rotation=1, log2_max_tskip_size=5, cross_component_pct=40),
    # monochrome (chroma_format_idc 0), intra pictures: the only kind the reference has a defined result for
    # (its inter path addresses chroma planes whatever the format, motion.cc:296-305)
    dict(name="cif8_I_mono", w=352, h=288, bd=8, st=2, seed=3101, monochrome=1),
    dict(name="cif10_I_mono_features", w=352, h=288, bd=10, st=2, seed=3102, monochrome=1, pcm_loop_filter_disable=1, scaling_list=1,
         n_slices=3, lf_across_slices_pct=50, **_F),
    dict(name="wvga12_I_mono_tiles_rext", w=832, h=480, bd=12, st=2, seed=3103, monochrome=1, tile_cols=3, tile_rows=2, slice_per_tile=1,
         lf_across_tiles=0, lf_across_slices_pct=0, tskip_pct=40, bypass_pct=15, pcm_pct=5, implicit_rdpcm=1, rotation=1,
         log2_max_tskip_size=5, intra_smoothing_disabled=1),
    dict(name="1080p8_I_mono", w=1920, h=1080, bd=8, st=2, seed=3104, monochrome=1, constrained_intra_pred=1),
]
# three cases whose final planes are stored in full (tests/golden/ref_small_pictures.npz)
FULL_PICTURE_CASES = ["qcif8_I", "qcif10_B", "ctb16_12_B"]
STAGES = ((0, "prefilter"), (1, "deblocked"), (2, "final"))


def digest(arrays):
    m = hashlib.md5()
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()


def make_picture(c):
    """-> SynthPicture, {slot: planes}, initial planes of the picture being decoded"""
    over = {k: v for k, v in c.items() if k not in ("name", "w", "h", "bd", "st", "seed")}
    sp = pysynth.SynthPicture(pysynth.default_config(c["w"], c["h"], c["bd"], c["st"], seed=c["seed"], **over))
    s = c["seed"] & 0xFFFF
    cf = 0 if c.get("monochrome") else c.get("chroma_format", 1)
    refs = {0: pysynth.fill_planes(c["w"], c["h"], c["bd"], s + 1, cf), 1: pysynth.fill_planes(c["w"], c["h"], c["bd"], s + 2, cf)}
    init = pysynth.fill_planes(c["w"], c["h"], c["bd"], s + 3, cf)
    return sp, refs, init


def picture_record(c, impl, keep_final=False):
    """All digests of one picture case through impl = 'oracle' | 'ref'."""
    sp, refs, init = make_picture(c)
    out = {}
    final = None
    for stage, key in STAGES:
        planes = [p.copy() for p in init]
        if impl == "oracle":
            pyoracle.reconstruct(sp.desc, sp.order, refs, planes, last_stage=stage)
        else:
            import pyref
            pyref.reconstruct(sp.desc, sp.order, refs, planes, sp.structure(), last_stage=stage)
        out[key] = digest(planes)
        final = planes
    w4, h4 = (c["w"] + 3) // 4, (c["h"] + 3) // 4
    sel = {1: np.zeros((h4, w4), bool), 0: np.zeros((h4, w4), bool)}
    sel[1][:, ::2] = True                       # units a direction's pass visits (deblock.cc:244-245)
    sel[0][::2, :] = True
    if impl == "oracle":
        L = pyoracle.lib()
        cb_log2, cb_part, tu_split, noedge = sp.structure()
        P = sp.d.params
        ef = noedge.copy()
        import ctypes as _C
        rc = L.oracle_derive_edge_flags(_C.byref(P), sp.d.slices, sp.d.n_slices, sp.d.ctbs, cb_log2.ctypes.data,
                                        cb_part.ctypes.data, tu_split.ctypes.data, ef.ctypes.data)
        assert rc == 0
        out["edge_flags"] = digest([ef])
        for v, key in ((1, "bs_v"), (0, "bs_h")):
            bs = np.zeros((h4, w4), np.uint8)
            L.oracle_derive_bs(sp.desc, v, bs.ctypes.data)
            out[key] = digest([bs[sel[v]]])
    else:
        import pyref
        st = sp.structure()
        out["edge_flags"] = digest([pyref.derive_edge_flags(sp.desc, st)])
        for v, key in ((1, "bs_v"), (0, "bs_h")):
            out[key] = digest([pyref.derive_bs(sp.desc, st, v)[sel[v]]])
    d = sp.d
    out["n_tus"], out["n_pus"], out["n_pcms"], out["n_coeffs"] = d.n_tus, d.n_pus, d.n_pcms, d.n_coeffs
    sp.close()
    return (out, final) if keep_final else out


# --------------------------------------------------------------------------------------- functions
def _px(bd):
    return np.uint16 if bd > 8 else np.uint8


def _coeffs(rng, n, nT):
    c = np.zeros((n, nT, nT), np.int16)
    for i in range(n):
        m = i % 6
        if m == 0:                                   # sparse low-frequency
            k = min(nT, 4)
            c[i, :k, :k] = rng.integers(-600, 601, (k, k))
        elif m == 1:                                 # dense moderate
            c[i] = rng.integers(-300, 301, (nT, nT))
        elif m == 2:                                 # extremes (every clip path)
            c[i] = rng.choice(np.array([-32768, 32767, 0, 0, 0], np.int16), (nT, nT))
        elif m == 3:                                 # a single coefficient anywhere
            c[i, rng.integers(nT), rng.integers(nT)] = rng.integers(-32768, 32768)
        elif m == 4:                                 # dense full range
            c[i] = rng.integers(-32768, 32768, (nT, nT))
        else:                                        # DC only
            c[i, 0, 0] = rng.integers(-2000, 2001)
    return c


QPEL_SIZES = [(4, 4), (8, 8), (16, 16), (12, 16), (32, 8), (64, 64), (24, 32), (8, 4), (4, 8), (16, 64)]
EPEL_SIZES = [(2, 2), (4, 4), (8, 8), (6, 8), (32, 32), (16, 4), (2, 4), (4, 2), (12, 16)]
PRED_SIZES = [(4, 4), (8, 4), (16, 16), (64, 32), (2, 2), (6, 8), (12, 16)]
PRED_PARAMS = [(0, (0, 0, 0, 0, 1)), (2, (0, 0, 0, 0, 1)), (1, (77, -20, 0, 0, 8)), (1, (-32, 63, 0, 0, 2)),
               (1, (64, 0, 0, 0, 6)), (3, (90, 12, -31, -64, 7)), (3, (64, 0, 64, 0, 12)), (3, (-32, -128, 95, 127, 3))]


def function_cases():
    """Deterministic list of dicts: key, kind, parameters and numpy inputs."""
    cases = []
    for bd in (8, 9, 10, 12):
        rng = np.random.default_rng([265, bd])
        for log2 in (2, 3, 4, 5):
            nT = 1 << log2
            n = 24 if log2 < 5 else 12
            kinds = ["transform_add"] + (["transform_dst_add"] if log2 == 2 else []) + ["tskip_add", "bypass_add"]
            for kind in kinds:
                cases.append(dict(key="%s/bd%d/n%d" % (kind, bd, nT), kind=kind, bd=bd, log2=log2,
                                  pred=rng.integers(0, 1 << bd, (n, nT, nT)).astype(_px(bd)),
                                  coeffs=_coeffs(rng, n, nT)))
        plane = rng.integers(0, 1 << bd, (160, 192)).astype(_px(bd))
        plane[:40, :40] = (1 << bd) - 1              # a saturated corner: int16 truncation of the first stage
        plane[100:, 150:] = 0
        for (w, h) in QPEL_SIZES:
            pos = np.stack([rng.integers(3, 192 - w - 4, 3), rng.integers(3, 160 - h - 4, 3)], axis=1)
            pos[0] = (3, 3)
            cases.append(dict(key="qpel/bd%d/%dx%d" % (bd, w, h), kind="qpel", bd=bd, w=w, h=h, plane=plane, pos=pos))
        for (w, h) in EPEL_SIZES:
            pos = np.stack([rng.integers(1, 192 - w - 2, 2), rng.integers(1, 160 - h - 2, 2)], axis=1)
            pos[0] = (1, 1)
            cases.append(dict(key="epel/bd%d/%dx%d" % (bd, w, h), kind="epel", bd=bd, w=w, h=h, plane=plane, pos=pos))
        for (w, h) in PRED_SIZES:
            n = 3
            cases.append(dict(key="pred/bd%d/%dx%d" % (bd, w, h), kind="pred", bd=bd, w=w, h=h,
                              dst=rng.integers(0, 1 << bd, (len(PRED_PARAMS), n, h, w)).astype(_px(bd)),
                              s0=rng.integers(-9000, 16384, (n, h, w)).astype(np.int16),
                              s1=rng.integers(-9000, 16384, (n, h, w)).astype(np.int16)))
    return cases


def _lib(impl):
    if impl == "oracle":
        return pyoracle.lib(), "oracle_"
    import pyref
    return pyref.lib(), "ref_"


def run_function_case(case, impl):
    """Output array(s) of one function case through impl = 'oracle' | 'ref' (one C call per block)."""
    L, pre = _lib(impl)
    k, bd = case["kind"], case["bd"]
    if k in ("transform_add", "transform_dst_add", "tskip_add", "bypass_add"):
        out = case["pred"].copy()
        nT = 1 << case["log2"]
        for i in range(len(out)):
            co = np.ascontiguousarray(case["coeffs"][i])
            if k == "transform_add":
                getattr(L, pre + "transform_add")(case["log2"], 0, bd, out[i].ctypes.data, nT, co.ctypes.data)
            elif k == "transform_dst_add":
                getattr(L, pre + "transform_add")(case["log2"], 1, bd, out[i].ctypes.data, nT, co.ctypes.data)
            elif k == "tskip_add":
                getattr(L, pre + "transform_skip_add")(case["log2"], bd, out[i].ctypes.data, nT, co.ctypes.data)
            else:
                getattr(L, pre + "transform_bypass_add")(case["log2"], bd, out[i].ctypes.data, nT, co.ctypes.data)
        return out
    if k in ("qpel", "epel"):
        w, h, plane = case["w"], case["h"], case["plane"]
        nf = 4 if k == "qpel" else 8
        out = np.zeros((len(case["pos"]), nf, nf, h, w), np.int16)
        fn = getattr(L, pre + ("put_qpel" if k == "qpel" else "put_epel"))
        for i, (x, y) in enumerate(case["pos"]):
            src = plane.ctypes.data + (int(y) * plane.shape[1] + int(x)) * plane.itemsize
            for fx in range(nf):
                for fy in range(nf):
                    fn(bd, out[i, fx, fy].ctypes.data, w, src, plane.shape[1], w, h, fx, fy)
        return out
    if k == "pred":
        w, h = case["w"], case["h"]
        out = case["dst"].copy()
        fn = getattr(L, pre + "put_pred")
        for j, (mode, (w0, o0, w1, o1, wd)) in enumerate(PRED_PARAMS):
            for i in range(out.shape[1]):
                fn(mode, bd, out[j, i].ctypes.data, w, case["s0"][i].ctypes.data, case["s1"][i].ctypes.data, w, w, h,
                   w0, o0, w1, o1, wd)
        return out
    raise KeyError(k)
