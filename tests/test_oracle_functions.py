"""Oracle self-checks (no GPU): the C restatement against (a) hand-derived known
answers from the formulas in SURVEY.md 8a and (b) an independent vectorised numpy
formulation of the same H.265 equations.  The oracle is "parity unpinned" against
the reference binary (see oracle/hevc_oracle.h); these tests are what stands in."""
import ctypes as C

import numpy as np
import pytest

import pyoracle

L = pyoracle.lib()
RNG = np.random.default_rng(265)


def dct_mat():
    return np.array([[L.oracle_dct_coeff(k, n) for n in range(32)] for k in range(32)], dtype=np.int64)


def px(bd):
    return np.uint16 if bd > 8 else np.uint8


def o_transform_add(log2, is_dst, bd, pred, coeffs):
    dst = pred.copy()
    c = np.ascontiguousarray(coeffs, np.int16)
    L.oracle_transform_add(log2, is_dst, bd, dst.ctypes.data, dst.shape[1], c.ctypes.data)
    return dst


def np_idct_add(log2, bd, pred, coeffs):
    nT = 1 << log2
    M = dct_mat()[::32 // nT, :nT]                      # M[j][i] = mat_dct[fact*j][i]
    Cc = coeffs.astype(np.int64).reshape(nT, nT)        # [j][c]
    g = np.clip((M.T @ Cc + 64) >> 7, -32768, 32767)    # g[i][c]
    out = (g @ M + (1 << (19 - bd))) >> (20 - bd)       # out[y][i]
    return np.clip(pred.astype(np.int64) + out, 0, (1 << bd) - 1).astype(pred.dtype)


def np_dst_add(bd, pred, coeffs):
    D = np.array([L.oracle_table(b"dst", i) for i in range(16)], dtype=np.int64).reshape(4, 4)
    Cc = coeffs.astype(np.int64).reshape(4, 4)
    g = np.clip((D.T @ Cc + 64) >> 7, -32768, 32767)
    out = np.clip((g @ D + (1 << (19 - bd))) >> (20 - bd), -32768, 32767)
    return np.clip(pred.astype(np.int64) + out, 0, (1 << bd) - 1).astype(pred.dtype)


def test_idct_dc_known_answer():
    # c=64 at DC, 8 bit: stage1 (64*64+64)>>7 = 32 everywhere in column 0, stage2 (64*32+2048)>>12 = 1
    for log2 in (2, 3, 4, 5):
        nT = 1 << log2
        co = np.zeros((nT, nT), np.int16); co[0, 0] = 64
        out = o_transform_add(log2, 0, 8, np.full((nT, nT), 100, np.uint8), co)
        assert (out == 101).all()
    # 10 bit: stage2 shift is 10: (64*32+512)>>10 = 2
    co = np.zeros((8, 8), np.int16); co[0, 0] = 64
    assert (o_transform_add(3, 0, 10, np.full((8, 8), 500, np.uint16), co) == 502).all()
    # saturation of the pixel clip
    co[0, 0] = 32767
    assert (o_transform_add(3, 0, 8, np.full((8, 8), 250, np.uint8), co) == 255).all()
    co[0, 0] = -32768
    assert (o_transform_add(3, 0, 8, np.full((8, 8), 5, np.uint8), co) == 0).all()


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_idct_matches_numpy_matrix_form(bd, log2):
    nT = 1 << log2
    for it in range(6):
        amp = [20, 300, 4000, 32767][it % 4]
        co = RNG.integers(-amp, amp + 1, (nT, nT)).astype(np.int16)
        if it >= 4:
            co[RNG.random((nT, nT)) < 0.8] = 0
        pred = RNG.integers(0, 1 << bd, (nT, nT)).astype(px(bd))
        assert np.array_equal(o_transform_add(log2, 0, bd, pred, co), np_idct_add(log2, bd, pred, co))


@pytest.mark.parametrize("bd", [8, 10])
def test_dst_matches_numpy_matrix_form(bd):
    for amp in (10, 500, 32767):
        co = RNG.integers(-amp, amp + 1, (4, 4)).astype(np.int16)
        pred = RNG.integers(0, 1 << bd, (4, 4)).astype(px(bd))
        assert np.array_equal(o_transform_add(2, 1, bd, pred, co), np_dst_add(bd, pred, co))


@pytest.mark.parametrize("bd", [8, 10])
def test_transform_skip_and_bypass_closed_form(bd):
    for log2 in (2, 3, 4, 5):
        nT = 1 << log2
        co = RNG.integers(-3000, 3001, (nT, nT)).astype(np.int16)
        pred = RNG.integers(0, 1 << bd, (nT, nT)).astype(px(bd))
        d = pred.copy(); L.oracle_transform_skip_add(log2, bd, d.ctypes.data, nT, co.ctypes.data)
        r = ((co.astype(np.int64) << (5 + log2)) + (1 << (19 - bd))) >> (20 - bd)      # transform.cc:533-537
        assert np.array_equal(d, np.clip(pred.astype(np.int64) + r, 0, (1 << bd) - 1))
        d = pred.copy(); L.oracle_transform_bypass_add(log2, bd, d.ctypes.data, nT, co.ctypes.data)
        assert np.array_equal(d, np.clip(pred.astype(np.int64) + co, 0, (1 << bd) - 1))


def o_dequant(log2, c_idx, intra, qp, bd, vals, pos, sf=None):
    nT = 1 << log2
    buf = np.zeros(nT * nT, np.int16)
    vals = np.ascontiguousarray(vals, np.int16); pos = np.ascontiguousarray(pos, np.uint16)
    L.oracle_dequant(buf.ctypes.data, log2, c_idx, intra, qp, bd, vals.ctypes.data, pos.ctypes.data, len(vals),
                     sf.ctypes.data if sf is not None else None)
    return buf


def test_dequant_flat_known_answers_and_int32_wrap():
    # qP=28 -> levelScale[4]=64 << 4 = 1024; 8x8 8-bit: bdShift = 8+3-9 = 2, offset 2
    out = o_dequant(3, 0, 1, 28, 8, [3, -3], [0, 9])
    assert out[0] == (3 * 1024 + 2) >> 2 and out[9] == (-3 * 1024 + 2) >> 2
    # 32-bit wraparound (transform.cc:464-470): 32767 * (72<<10) overflows int32
    # qP' = 65 (12-bit: QpBdOffset 24) -> levelScale[5]=72 << 10; bdShift = 12+3-9 = 6
    qp, c = 65, 32767
    fact = 72 << 10
    wrapped = ((c * fact + (1 << 5)) + 2**31) % 2**32 - 2**31
    exp = int(np.clip(wrapped >> 6, -32768, 32767))
    assert o_dequant(3, 0, 0, qp, 12, [c], [5])[5] == exp
    assert exp != int(np.clip((c * fact + 32) >> 6, -32768, 32767))    # really exercises the wrap


def test_dequant_scaling_list_numpy():
    sf = RNG.integers(1, 256, 4064).astype(np.uint8)
    offs = {2: 0, 3: 96, 4: 96 + 384, 5: 96 + 384 + 1536}
    for log2 in (2, 3, 4, 5):
        nT = 1 << log2
        for intra in (0, 1):
            for c_idx in ((0,) if log2 == 5 else (0, 1, 2)):
                n = min(nT * nT, 40)
                pos = RNG.choice(nT * nT, n, replace=False)
                vals = RNG.integers(-2000, 2001, n)
                qp, bd = int(RNG.integers(0, 52)), 8
                mid = c_idx + (0 if intra else (3 if nT < 32 else 1))
                scl = sf[offs[log2] + mid * nT * nT:][:nT * nT].astype(np.int64)
                fact = (scl[pos] * [40, 45, 51, 57, 64, 72][qp % 6]) << (qp // 6)
                bdShift = bd + log2 - 5
                exp = np.zeros(nT * nT, np.int64)
                exp[pos] = np.clip((vals * fact + (1 << (bdShift - 1))) >> bdShift, -32768, 32767)
                assert np.array_equal(o_dequant(log2, c_idx, intra, qp, bd, vals, pos, sf), exp)


QPEL = {1: [-1, 4, -10, 58, 17, -5, 1, 0], 2: [-1, 4, -11, 40, 40, -11, 4, -1], 3: [0, 1, -5, 17, 58, -10, 4, -1]}
EPEL = {1: [-2, 58, 10, -2], 2: [-4, 54, 16, -2], 3: [-6, 46, 28, -4], 4: [-4, 36, 36, -4],
        5: [-4, 28, 46, -6], 6: [-2, 16, 54, -4], 7: [-2, 10, 58, -2]}


def np_interp(plane, bd, x0, y0, w, h, fx, fy, taps, before):
    """8.5.3.2.2: separable interpolation, int16 truncation after each stage."""
    n = len(next(iter(taps.values())))
    src = plane.astype(np.int64)
    if fx == 0 and fy == 0:
        return (src[y0:y0 + h, x0:x0 + w] << (14 - bd)).astype(np.int16)
    rows = src[y0 - before:y0 + h + n - 1 - before, :]
    if fx:
        t = sum(taps[fx][k] * rows[:, x0 - before + k:x0 - before + k + w] for k in range(n)) >> (bd - 8)
    else:
        t = rows[:, x0:x0 + w]
    t = t.astype(np.int16).astype(np.int64)
    if fy:
        v = sum(taps[fy][k] * t[k:k + h] for k in range(n)) >> (6 if fx else bd - 8)
    else:
        v = t[before:before + h]
    return v.astype(np.int16)


@pytest.mark.parametrize("bd", [8, 10])
def test_qpel_epel_match_numpy(bd):
    plane = RNG.integers(0, 1 << bd, (96, 128)).astype(px(bd))
    plane[:8, :16] = (1 << bd) - 1                     # saturated corner: largest intermediates
    for (w, h) in [(4, 4), (8, 16), (16, 8), (24, 32), (64, 64)]:
        for fx in range(4):
            for fy in range(4):
                x0, y0 = 8, 8
                out = np.zeros((h, w), np.int16)
                L.oracle_put_qpel(bd, out.ctypes.data, w, plane.ctypes.data + (y0 * 128 + x0) * plane.itemsize, 128,
                                  w, h, fx, fy)
                assert np.array_equal(out, np_interp(plane, bd, x0, y0, w, h, fx, fy, QPEL, 3)), (w, h, fx, fy)
    for (w, h) in [(2, 2), (4, 8), (8, 4), (32, 32)]:
        for fx in range(8):
            for fy in range(8):
                x0, y0 = 4, 4
                out = np.zeros((h, w), np.int16)
                L.oracle_put_epel(bd, out.ctypes.data, w, plane.ctypes.data + (y0 * 128 + x0) * plane.itemsize, 128,
                                  w, h, fx, fy)
                assert np.array_equal(out, np_interp(plane, bd, x0, y0, w, h, fx, fy, EPEL, 1)), (w, h, fx, fy)


def test_qpel_fullpel_and_flat_known_answers():
    flat = np.full((32, 32), 100, np.uint8)
    out = np.zeros((8, 8), np.int16)
    for fx in range(4):
        for fy in range(4):
            L.oracle_put_qpel(8, out.ctypes.data, 8, flat.ctypes.data + 8 * 32 + 8, 32, 8, 8, fx, fy)
            assert (out == 100 * 64).all()            # taps sum to 64: flat input -> 14-bit value
    flat10 = np.full((32, 32), 1000, np.uint16)
    L.oracle_put_qpel(10, out.ctypes.data, 8, flat10.ctypes.data + (8 * 32 + 8) * 2, 32, 8, 8, 2, 2)
    assert (out == 1000 * 16).all()


@pytest.mark.parametrize("bd", [8, 10])
def test_put_pred_closed_forms(bd):
    w, h = 16, 8
    s0 = RNG.integers(-8000, 16384, (h, w)).astype(np.int16)
    s1 = RNG.integers(-8000, 16384, (h, w)).astype(np.int16)
    a, b = s0.astype(np.int64), s1.astype(np.int64)
    mx = (1 << bd) - 1
    dst = np.zeros((h, w), px(bd))

    def run(mode, w0=0, o0=0, w1=0, o1=0, wd=1):
        L.oracle_put_pred(mode, bd, dst.ctypes.data, w, s0.ctypes.data, s1.ctypes.data, w, w, h, w0, o0, w1, o1, wd)
        return dst.astype(np.int64)
    sh = 14 - bd
    assert np.array_equal(run(0), np.clip((a + (1 << (sh - 1))) >> sh, 0, mx))
    assert np.array_equal(run(2), np.clip((a + b + (1 << sh)) >> (sh + 1), 0, mx))
    for (w0, o0, w1, o1, wd) in [(64, 0, 64, 0, 6 + sh), (-20, 5, 90, -7, 9), (37, -100, -3, 20, 3)]:
        assert np.array_equal(run(1, w0, o0, 0, 0, wd), np.clip(((a * w0 + (1 << (wd - 1))) >> wd) + o0, 0, mx))
        assert np.array_equal(run(3, w0, o0, w1, o1, wd),
                              np.clip((a * w0 + b * w1 + ((o0 + o1 + 1) << wd)) >> (wd + 1), 0, mx))


# ---- intra prediction: independent formulation straight from 8.4.4.2.3-6 ----
ANGLE = [0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
         -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32]


def spec_intra(p, nT, c_idx, mode, bd, strong):
    """p: dict index -> sample with p[(x,-1)] x=-1..2nT-1 and p[(-1,y)] y=-1..2nT-1 (spec coordinates)."""
    P = dict(p)
    if c_idx == 0 and mode != 1 and nT != 4:
        mind = min(abs(mode - 26), abs(mode - 10))
        if mind > {8: 7, 16: 1, 32: 0}[nT]:
            F = {}
            bi = (strong and nT == 32 and abs(P[(-1, -1)] + P[(63, -1)] - 2 * P[(31, -1)]) < (1 << (bd - 5)) and
                  abs(P[(-1, -1)] + P[(-1, 63)] - 2 * P[(-1, 31)]) < (1 << (bd - 5)))
            if bi:
                F[(-1, -1)] = P[(-1, -1)]
                for y in range(63):
                    F[(-1, y)] = ((63 - y) * P[(-1, -1)] + (y + 1) * P[(-1, 63)] + 32) >> 6
                F[(-1, 63)] = P[(-1, 63)]
                for x in range(63):
                    F[(x, -1)] = ((63 - x) * P[(-1, -1)] + (x + 1) * P[(63, -1)] + 32) >> 6
                F[(63, -1)] = P[(63, -1)]
            else:
                F[(-1, -1)] = (P[(-1, 0)] + 2 * P[(-1, -1)] + P[(0, -1)] + 2) >> 2
                for y in range(2 * nT - 1):
                    F[(-1, y)] = (P[(-1, y + 1)] + 2 * P[(-1, y)] + P[(-1, y - 1)] + 2) >> 2
                F[(-1, 2 * nT - 1)] = P[(-1, 2 * nT - 1)]
                for x in range(2 * nT - 1):
                    F[(x, -1)] = (P[(x - 1, -1)] + 2 * P[(x, -1)] + P[(x + 1, -1)] + 2) >> 2
                F[(2 * nT - 1, -1)] = P[(2 * nT - 1, -1)]
            P = F
    out = np.zeros((nT, nT), np.int64)
    lg = nT.bit_length() - 1
    if mode == 0:
        for y in range(nT):
            for x in range(nT):
                out[y, x] = ((nT - 1 - x) * P[(-1, y)] + (x + 1) * P[(nT, -1)] + (nT - 1 - y) * P[(x, -1)] +
                             (y + 1) * P[(-1, nT)] + nT) >> (lg + 1)
    elif mode == 1:
        dc = (sum(P[(x, -1)] for x in range(nT)) + sum(P[(-1, y)] for y in range(nT)) + nT) >> (lg + 1)
        out[:] = dc
        if c_idx == 0 and nT < 32:
            out[0, 0] = (P[(-1, 0)] + 2 * dc + P[(0, -1)] + 2) >> 2
            for x in range(1, nT):
                out[0, x] = (P[(x, -1)] + 3 * dc + 2) >> 2
            for y in range(1, nT):
                out[y, 0] = (P[(-1, y)] + 3 * dc + 2) >> 2
    else:
        ang = ANGLE[mode]
        ref = {}
        if mode >= 18:
            for x in range(nT + 1):
                ref[x] = P[(-1 + x, -1)]
            if ang < 0:
                inv = round(8192 / ang)
                if (nT * ang) >> 5 < -1:
                    for x in range((nT * ang) >> 5, 0):
                        ref[x] = P[(-1, -1 + ((x * inv + 128) >> 8))]
            else:
                for x in range(nT + 1, 2 * nT + 1):
                    ref[x] = P[(-1 + x, -1)]
            for y in range(nT):
                i, f = ((y + 1) * ang) >> 5, ((y + 1) * ang) & 31
                for x in range(nT):
                    out[y, x] = ((32 - f) * ref[x + i + 1] + f * ref[x + i + 2] + 16) >> 5 if f else ref[x + i + 1]
            if mode == 26 and c_idx == 0 and nT < 32:
                for y in range(nT):
                    out[y, 0] = min(max(P[(0, -1)] + ((P[(-1, y)] - P[(-1, -1)]) >> 1), 0), (1 << bd) - 1)
        else:
            for x in range(nT + 1):
                ref[x] = P[(-1, -1 + x)]
            if ang < 0:
                inv = round(8192 / ang)
                if (nT * ang) >> 5 < -1:
                    for x in range((nT * ang) >> 5, 0):
                        ref[x] = P[(-1 + ((x * inv + 128) >> 8), -1)]
            else:
                for x in range(nT + 1, 2 * nT + 1):
                    ref[x] = P[(-1, -1 + x)]
            for x in range(nT):
                i, f = ((x + 1) * ang) >> 5, ((x + 1) * ang) & 31
                for y in range(nT):
                    out[y, x] = ((32 - f) * ref[y + i + 1] + f * ref[y + i + 2] + 16) >> 5 if f else ref[y + i + 1]
            if mode == 10 and c_idx == 0 and nT < 32:
                for x in range(nT):
                    out[0, x] = min(max(P[(-1, 0)] + ((P[(x, -1)] - P[(-1, -1)]) >> 1), 0), (1 << bd) - 1)
    return out


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("nT", [4, 8, 16, 32])
def test_intra_predictors_match_spec_formulation(bd, nT):
    dt = px(bd)
    for c_idx in (0, 1):
        for smooth in (0, 1):
            if smooth:                      # smooth ramp: triggers the strong (bilinear) filter at 32x32
                base = np.linspace(200, 260, 4 * nT + 1).astype(np.int64) * (1 << (bd - 8)) // 2
            else:
                base = RNG.integers(0, 1 << bd, 4 * nT + 1)
            border = base.astype(dt)        # index i + 2nT ; i<0: left column (y = -i-1), i>0 top row (x = i-1)
            P = {(-1, -1): int(border[2 * nT])}
            for i in range(1, 2 * nT + 1):
                P[(i - 1, -1)] = int(border[2 * nT + i]); P[(-1, i - 1)] = int(border[2 * nT - i])
            for mode in range(35):
                dst = np.zeros((nT, nT), dt)
                L.oracle_intra_predict(bd, 1, dst.ctypes.data, nT, nT, c_idx, mode,
                                       border.ctypes.data + 2 * nT * border.itemsize)
                exp = spec_intra(P, nT, c_idx, mode, bd, True)
                assert np.array_equal(dst, exp), (bd, nT, c_idx, smooth, mode)
