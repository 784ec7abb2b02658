"""Parity with libde265 ITSELF.  tests/golden/ref_*.json were produced by the compiled reference
(oracle/_ref/libde265_ref.so, built from /root/reference by oracle/Makefile; generator: tools/make_ref_golden.py).

CPU tests (always):  the plain-C restatement reproduces every fixture -> the oracle is PINNED.
CPU tests (where the compiled reference is present): it still reproduces the fixtures, and restatement ==
                     reference on fresh random pictures and function inputs that no fixture holds.
GPU tests:           the HIP path reproduces the same fixtures through the C ABI (and, where the compiled
                     reference travelled along, matches it live on further random pictures)."""
import json
import os

import numpy as np
import pytest

import pyoracle
import pyref
import ref_cases
from libde265_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD_P = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_pictures.json")))["cases"]
GOLD_F = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_functions.json")))["cases"]
HAVE_REF = pyref.available()
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="compiled reference (oracle/_ref) not present here")
PIC_IDS = [c["name"] for c in ref_cases.PICTURE_CASES]


def test_fixtures_cover_the_case_lists():
    assert sorted(GOLD_P) == sorted(PIC_IDS)
    assert sorted(GOLD_F) == sorted(c["key"] for c in ref_cases.function_cases())


# ------------------------------------------------------------------ CPU: restatement == reference fixtures
@pytest.mark.parametrize("case", ref_cases.PICTURE_CASES, ids=PIC_IDS)
def test_oracle_reproduces_reference_pictures(case):
    """a1-a16 at picture level: planes after every stage, derived edge flags, both bS passes."""
    assert ref_cases.picture_record(case, "oracle") == GOLD_P[case["name"]]


def test_oracle_reproduces_reference_function_slots():
    """a2-a4, a7-a9: the fallback vtable slots on seeded blocks, 8/9/10/12 bit."""
    bad = [c["key"] for c in ref_cases.function_cases()
           if ref_cases.digest([ref_cases.run_function_case(c, "oracle")]) != GOLD_F[c["key"]]]
    assert not bad, bad


def test_full_small_pictures_match():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_small_pictures.npz"))
    for name in ref_cases.FULL_PICTURE_CASES:
        case = next(c for c in ref_cases.PICTURE_CASES if c["name"] == name)
        sp, refs, init = ref_cases.make_picture(case)
        planes = [p.copy() for p in init]
        pyoracle.reconstruct(sp.desc, sp.order, refs, planes)
        for i in range(3):
            exp = z["%s/%d" % (name, i)]
            bad = np.argwhere(planes[i] != exp)
            assert bad.size == 0, "%s comp %d: %d samples differ, first (y,x)=%s" % (name, i, len(bad), tuple(bad[0]))


# ------------------------------------------------------------------ CPU, build container: live reference
@needs_ref
def test_reference_still_reproduces_fixtures():
    """Guards against stale fixtures (generator or case list changed without re-running make_ref_golden.py)."""
    for case in ref_cases.PICTURE_CASES:
        assert ref_cases.picture_record(case, "ref") == GOLD_P[case["name"]], case["name"]
    for c in ref_cases.function_cases():
        assert ref_cases.digest([ref_cases.run_function_case(c, "ref")]) == GOLD_F[c["key"]], c["key"]


@needs_ref
def test_oracle_equals_reference_on_fresh_random_pictures():
    """Every feature switch at random, all three stages + edge flags + bS, inputs no fixture holds."""
    import ref_sweep
    rng = np.random.default_rng(int(os.environ.get("DE265HIP_TEST_SEED", "77")))
    for it in range(int(os.environ.get("DE265HIP_REF_RANDOM", "60"))):
        w, h, bd, st, over = ref_sweep.small_config(rng, it)
        bad = ref_sweep.compare(w, h, bd, st, 31000 + it, over)
        assert not bad, "config %d: %dx%d bd=%d st=%d %r: %s" % (it, w, h, bd, st, over, bad)
    for it in range(3):
        w, h, bd, st, over = ref_sweep.mid_config(rng)
        bad = ref_sweep.compare(w, h, bd, st, 32000 + it, over)
        assert not bad, "mid config %d: %dx%d bd=%d st=%d %r: %s" % (it, w, h, bd, st, over, bad)


@needs_ref
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_oracle_equals_reference_on_fresh_random_blocks(bd):
    """Many more blocks than the fixtures hold, straight against the fallback slots."""
    LO, LR = pyoracle.lib(), pyref.lib()
    rng = np.random.default_rng([99, bd])
    dt = np.uint16 if bd > 8 else np.uint8
    for log2 in (2, 3, 4, 5):
        nT = 1 << log2
        co = ref_cases._coeffs(rng, 600 if log2 < 5 else 120, nT)
        pred = rng.integers(0, 1 << bd, co.shape).astype(dt)
        for kind in (0, 1) if log2 == 2 else (0,):
            a, b = pred.copy(), pred.copy()
            for i in range(len(co)):
                LO.oracle_transform_add(log2, kind, bd, a[i].ctypes.data, nT, co[i].ctypes.data)
                LR.ref_transform_add(log2, kind, bd, b[i].ctypes.data, nT, co[i].ctypes.data)
            assert np.array_equal(a, b), (log2, kind)
        for fo, fr in ((LO.oracle_transform_skip_add, LR.ref_transform_skip_add),
                       (LO.oracle_transform_bypass_add, LR.ref_transform_bypass_add)):
            a, b = pred.copy(), pred.copy()
            for i in range(len(co)):
                fo(log2, bd, a[i].ctypes.data, nT, co[i].ctypes.data)
                fr(log2, bd, b[i].ctypes.data, nT, co[i].ctypes.data)
            assert np.array_equal(a, b), log2
    plane = rng.integers(0, 1 << bd, (96, 128)).astype(dt)
    for it in range(400):
        w, h = int(rng.choice([4, 8, 12, 16, 24, 32, 64])), int(rng.choice([4, 8, 12, 16, 24, 32, 64]))
        x, y = int(rng.integers(3, 128 - w - 4 + 1)), int(rng.integers(3, 96 - h - 4 + 1))
        src = plane.ctypes.data + (y * 128 + x) * plane.itemsize
        dx, dy = int(rng.integers(4)), int(rng.integers(4))
        a, b = np.zeros((h, w), np.int16), np.zeros((h, w), np.int16)
        LO.oracle_put_qpel(bd, a.ctypes.data, w, src, 128, w, h, dx, dy)
        LR.ref_put_qpel(bd, b.ctypes.data, w, src, 128, w, h, dx, dy)
        assert np.array_equal(a, b), ("qpel", w, h, dx, dy)
        w2, h2, mx, my = w // 2, h // 2, int(rng.integers(8)), int(rng.integers(8))
        a, b = np.zeros((h2, w2), np.int16), np.zeros((h2, w2), np.int16)
        LO.oracle_put_epel(bd, a.ctypes.data, w2, src, 128, w2, h2, mx, my)
        LR.ref_put_epel(bd, b.ctypes.data, w2, src, 128, w2, h2, mx, my)
        assert np.array_equal(a, b), ("epel", w2, h2, mx, my)
        s0 = rng.integers(-16384, 16384, (h, w)).astype(np.int16)
        s1 = rng.integers(-16384, 16384, (h, w)).astype(np.int16)
        mode = int(rng.integers(4))
        w0, w1 = int(rng.integers(-128, 128)), int(rng.integers(-128, 128))
        o0, o1 = (int(rng.integers(-128, 128)) << (bd - 8) for _ in range(2))
        wd = int(rng.integers(0, 8)) + max(2, 14 - bd)
        a = rng.integers(0, 1 << bd, (h, w)).astype(dt)
        b = a.copy()
        LO.oracle_put_pred(mode, bd, a.ctypes.data, w, s0.ctypes.data, s1.ctypes.data, w, w, h, w0, o0, w1, o1, wd)
        LR.ref_put_pred(mode, bd, b.ctypes.data, w, s0.ctypes.data, s1.ctypes.data, w, w, h, w0, o0, w1, o1, wd)
        assert np.array_equal(a, b), ("pred", mode, w, h, w0, o0, w1, o1, wd)


# ------------------------------------------------------------------ GPU: HIP path == reference fixtures
@pytest.fixture(scope="module")
def dec():
    from libde265_amd import backend
    assert backend.device_count() > 0, "no GPU visible: HIP path cannot run"
    d = backend.Decoder()
    yield d
    d.close()


def _gpu_picture_digests(dec, case):
    sp, refs, init = ref_cases.make_picture(case)
    w, h, bd, cf = case["w"], case["h"], case["bd"], 0 if case.get("monochrome") else case.get("chroma_format", 1)
    for s, pl in refs.items():
        dec.dpb_alloc(s, w, h, bd, chroma_format=cf)
        dec.upload(s, pl)
    dec.dpb_alloc(2, w, h, bd, chroma_format=cf)
    pic = dec.build(2, sp.desc)
    out = {}
    try:
        for stage, key in ref_cases.STAGES:
            dec.upload(2, init)
            dec.run(pic, stage)
            dec.sync()
            out[key] = ref_cases.digest(dec.download(2, w, h, bd))
    finally:
        pic.free()
    return out, sp


@pytest.mark.gpu
@pytest.mark.parametrize("case", ref_cases.PICTURE_CASES, ids=PIC_IDS)
def test_gpu_reproduces_reference_pictures(dec, case):
    """The HIP path against libde265's own output: every stage of every picture case, incl. the full-size
    BASELINE configurations (720p 8-bit all-intra, 1080p 8-bit I and B, 4K Main10 I and B)."""
    from libde265_amd import backend
    got, sp = _gpu_picture_digests(dec, case)
    gold = GOLD_P[case["name"]]
    for _, key in ref_cases.STAGES:
        assert got[key] == gold[key], "%s: %s differs from the reference" % (case["name"], key)
    # a11 through the product's host helper
    cb_log2, cb_part, tu_split, noedge = sp.structure()
    d = sp.d
    ef = noedge.copy()
    backend.derive_edge_flags(d.params, d.slices, d.n_slices, d.ctbs, cb_log2, cb_part, tu_split, ef)
    assert ref_cases.digest([ef]) == gold["edge_flags"]


def _gpu_function_output(be, case):
    """One function case through the batched de265hip_fn_* entry points: the blocks are laid out on a grid in one plane."""
    k, bd = case["kind"], case["bd"]
    dt = np.uint16 if bd > 8 else np.uint8
    if k in ("transform_add", "transform_dst_add", "tskip_add", "bypass_add"):
        pred, co = case["pred"], case["coeffs"]
        n, nT = len(pred), pred.shape[1]
        cols = 8
        rows = (n + cols - 1) // cols
        plane = np.zeros((rows * nT, cols * nT), dt)
        xy = np.array([((i % cols) * nT, (i // cols) * nT) for i in range(n)], np.int32)
        for i, (x, y) in enumerate(xy):
            plane[y:y + nT, x:x + nT] = pred[i]
        if k == "transform_add":
            be.fn_transform_add(plane, bd, case["log2"], xy, co)
        elif k == "transform_dst_add":
            be.fn_transform_add(plane, bd, case["log2"], xy, co, dst=True)
        elif k == "tskip_add":
            be.fn_transform_skip_add(plane, bd, case["log2"], xy, co)
        else:
            be.fn_transform_bypass_add(plane, bd, case["log2"], xy, co)
        return np.stack([plane[y:y + nT, x:x + nT] for x, y in xy])
    if k in ("qpel", "epel"):
        w, h, plane, pos = case["w"], case["h"], case["plane"], case["pos"].astype(np.int32)
        nf = 4 if k == "qpel" else 8
        out = np.zeros((len(pos), nf, nf, h, w), np.int16)
        fn = be.fn_put_qpel if k == "qpel" else be.fn_put_epel
        for fx in range(nf):
            for fy in range(nf):
                out[:, fx, fy] = fn(plane, bd, w, h, fx, fy, pos)
        return out
    if k == "pred":
        w, h = case["w"], case["h"]
        out = case["dst"].copy()
        n = out.shape[1]
        for j, (mode, (w0, o0, w1, o1, wd)) in enumerate(ref_cases.PRED_PARAMS):
            plane = np.zeros((h, n * w), dt)
            xy = np.array([(i * w, 0) for i in range(n)], np.int32)
            for i in range(n):
                plane[:, i * w:(i + 1) * w] = out[j, i]
            be.fn_put_pred(plane, bd, mode, xy, case["s0"], case["s1"], w0, o0, w1, o1, wd)
            for i in range(n):
                out[j, i] = plane[:, i * w:(i + 1) * w]
        return out
    raise KeyError(k)


@pytest.mark.gpu
def test_gpu_reproduces_reference_function_slots():
    from libde265_amd import backend as be
    assert be.device_count() > 0
    bad = [c["key"] for c in ref_cases.function_cases()
           if ref_cases.digest([_gpu_function_output(be, c)]) != GOLD_F[c["key"]]]
    assert not bad, bad


@pytest.mark.gpu
@needs_ref
def test_gpu_equals_live_reference_on_random_pictures(dec):
    """Where the compiled reference travelled to the GPU box: HIP path against it on random pictures no fixture holds."""
    import pysynth
    import ref_sweep
    rng = np.random.default_rng(20261005)
    for it in range(40):
        w, h, bd, st, over = ref_sweep.small_config(rng, it) if it % 4 else ref_sweep.mid_config(rng)
        cfg = pysynth.default_config(w, h, bd, st, seed=41000 + it, **over)
        sp = pysynth.SynthPicture(cfg)
        refs = {0: pysynth.fill_planes(w, h, bd, 100 + it), 1: pysynth.fill_planes(w, h, bd, 200 + it)}
        init = pysynth.fill_planes(w, h, bd, 999)
        exp = [p.copy() for p in init]
        pyref.reconstruct(sp.desc, sp.order, refs, exp, sp.structure(), _abi.STAGE_FINAL)
        for s, pl in refs.items():
            dec.dpb_alloc(s, w, h, bd)
            dec.upload(s, pl)
        dec.dpb_alloc(2, w, h, bd)
        dec.upload(2, init)
        pic = dec.build(2, sp.desc)
        try:
            dec.run(pic, _abi.STAGE_FINAL)
            dec.sync()
            got = dec.download(2, w, h, bd)
        finally:
            pic.free()
        for c in range(3):
            bad = np.argwhere(got[c] != exp[c])
            assert bad.size == 0, "config %d: %dx%d bd=%d st=%d %r comp %d: %d samples differ from libde265, first %s" % (
                it, w, h, bd, st, over, c, len(bad), tuple(bad[0]))
