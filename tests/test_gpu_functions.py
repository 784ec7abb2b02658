"""Function-level (vtable-slot) parity on the GPU: de265hip_fn_* through the C ABI
against the oracle's restatement of the fallback slots, bit-exact, 8 and 10 bit."""
import numpy as np
import pytest

import pyoracle

pytestmark = pytest.mark.gpu
L = pyoracle.lib()
RNG = np.random.default_rng(950)


@pytest.fixture(scope="module")
def be():
    from libde265_amd import backend
    assert backend.device_count() > 0, "no GPU visible"
    return backend


def px(bd):
    return np.uint16 if bd > 8 else np.uint8


def grid_blocks(nT, pw, ph, n):
    cols, rows = pw // nT, ph // nT
    idx = RNG.choice(cols * rows, size=min(n, cols * rows), replace=False)
    return np.stack([(idx % cols) * nT, (idx // cols) * nT], axis=1).astype(np.int32)


def rand_coeffs(n, nT, kind):
    c = np.zeros((n, nT, nT), np.int16)
    for i in range(n):
        m = i % 4
        if m == 0:   # sparse low-frequency
            k = min(nT, 4)
            c[i, :k, :k] = RNG.integers(-600, 601, (k, k))
        elif m == 1:  # dense moderate
            c[i] = RNG.integers(-300, 301, (nT, nT))
        elif m == 2:  # extremes (clipping paths)
            c[i] = RNG.choice(np.array([-32768, 32767, 0, 0, 0], np.int16), (nT, nT))
        else:        # single coefficient anywhere
            c[i, RNG.integers(nT), RNG.integers(nT)] = RNG.integers(-32768, 32768)
    return c


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_transform_add(be, bd, log2):
    nT = 1 << log2
    pw, ph = 1024, 512
    blocks = grid_blocks(nT, pw, ph, 20000 if log2 <= 3 else 512)
    n = len(blocks)
    plane = RNG.integers(0, 1 << bd, (ph, pw)).astype(px(bd))
    for kind in ([0, 1] if log2 == 2 else [0]):
        co = rand_coeffs(n, nT, kind)
        exp = plane.copy()
        for (x, y), c in zip(blocks, co):
            L.oracle_transform_add(log2, kind, bd, exp.ctypes.data + (int(y) * pw + int(x)) * exp.itemsize, pw,
                                   np.ascontiguousarray(c).ctypes.data)
        got = plane.copy()
        be.fn_transform_add(got, bd, log2, blocks, co, dst=bool(kind))
        assert np.array_equal(got, exp), "kind %d" % kind


@pytest.mark.parametrize("bd", [8, 10])
def test_transform_skip_and_bypass(be, bd):
    for log2 in (2, 3, 4, 5):
        nT = 1 << log2
        pw, ph = 256, 256
        blocks = grid_blocks(nT, pw, ph, 64)
        co = rand_coeffs(len(blocks), nT, 0)
        plane = RNG.integers(0, 1 << bd, (ph, pw)).astype(px(bd))
        for fn, ofn in ((be.fn_transform_skip_add, L.oracle_transform_skip_add),
                        (be.fn_transform_bypass_add, L.oracle_transform_bypass_add)):
            exp = plane.copy()
            for (x, y), c in zip(blocks, co):
                ofn(log2, bd, exp.ctypes.data + (int(y) * pw + int(x)) * exp.itemsize, pw,
                    np.ascontiguousarray(c).ctypes.data)
            got = plane.copy()
            fn(got, bd, log2, blocks, co)
            assert np.array_equal(got, exp)


@pytest.mark.parametrize("bd", [8, 10])
def test_put_qpel_all_fractions(be, bd):
    pw, ph = 512, 256
    plane = RNG.integers(0, 1 << bd, (ph, pw)).astype(px(bd))
    plane[:40, :40] = (1 << bd) - 1
    for (w, h) in [(4, 4), (8, 8), (16, 16), (12, 16), (32, 8), (64, 64), (24, 32)]:
        blocks = np.stack([RNG.integers(3, pw - w - 4, 12), RNG.integers(3, ph - h - 4, 12)], axis=1).astype(np.int32)
        blocks[0] = (3, 3)
        for dx in range(4):
            for dy in range(4):
                got = be.fn_put_qpel(plane, bd, w, h, dx, dy, blocks)
                exp = np.zeros_like(got)
                for i, (x, y) in enumerate(blocks):
                    L.oracle_put_qpel(bd, exp[i].ctypes.data, w, plane.ctypes.data + (int(y) * pw + int(x)) * plane.itemsize,
                                      pw, w, h, dx, dy)
                assert np.array_equal(got, exp), (w, h, dx, dy)


@pytest.mark.parametrize("bd", [8, 10])
def test_put_epel_all_fractions(be, bd):
    pw, ph = 256, 128
    plane = RNG.integers(0, 1 << bd, (ph, pw)).astype(px(bd))
    for (w, h) in [(2, 2), (4, 4), (8, 8), (6, 8), (32, 32), (16, 4)]:
        blocks = np.stack([RNG.integers(1, pw - w - 2, 8), RNG.integers(1, ph - h - 2, 8)], axis=1).astype(np.int32)
        for mx in range(8):
            for my in range(8):
                got = be.fn_put_epel(plane, bd, w, h, mx, my, blocks)
                exp = np.zeros_like(got)
                for i, (x, y) in enumerate(blocks):
                    L.oracle_put_epel(bd, exp[i].ctypes.data, w, plane.ctypes.data + (int(y) * pw + int(x)) * plane.itemsize,
                                      pw, w, h, mx, my)
                assert np.array_equal(got, exp), (w, h, mx, my)


@pytest.mark.parametrize("bd", [8, 10])
def test_put_pred_modes(be, bd):
    pw, ph = 256, 128
    for (w, h) in [(4, 4), (8, 4), (16, 16), (64, 32)]:
        blocks = grid_blocks(64, pw, ph, 6)
        n = len(blocks)
        s0 = RNG.integers(-9000, 16384, (n, h, w)).astype(np.int16)
        s1 = RNG.integers(-9000, 16384, (n, h, w)).astype(np.int16)
        for mode, (w0, o0, w1, o1, wd) in [(0, (0, 0, 0, 0, 1)), (2, (0, 0, 0, 0, 1)), (1, (77, -20, 0, 0, 8)),
                                           (1, (-32, 63, 0, 0, 2)), (3, (90, 12, -31, -64, 7)), (3, (64, 0, 64, 0, 12))]:
            plane = RNG.integers(0, 1 << bd, (ph, pw)).astype(px(bd))
            exp = plane.copy()
            for i, (x, y) in enumerate(blocks):
                L.oracle_put_pred(mode, bd, exp.ctypes.data + (int(y) * pw + int(x)) * exp.itemsize, pw,
                                  s0[i].ctypes.data, s1[i].ctypes.data, w, w, h, w0, o0, w1, o1, wd)
            got = plane.copy()
            be.fn_put_pred(got, bd, mode, blocks, s0, s1, w0, o0, w1, o1, wd)
            assert np.array_equal(got, exp), (w, h, mode)


def test_argument_validation(be):
    from libde265_amd import _abi
    plane = np.zeros((32, 32), np.uint8)
    co = np.zeros((1, 8, 8), np.int16)
    with pytest.raises(be.De265HipError) as e:
        be.fn_transform_add(plane, 8, 3, [(28, 0)], co)            # block leaves the plane
    assert e.value.code == _abi.ERROR_PARAMETER_OUT_OF_RANGE
    with pytest.raises(be.De265HipError):
        be.fn_transform_add(plane, 8, 3, [(0, 0)], co, dst=True)   # DST exists for 4x4 only
    with pytest.raises(be.De265HipError):
        be.fn_put_qpel(plane, 8, 8, 8, 1, 0, [(1, 4)])             # filter margin outside the plane
    be.fn_transform_add(plane, 8, 3, np.zeros((0, 2), np.int32), np.zeros((0, 8, 8), np.int16))   # empty batch is fine
