#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_digests.json: MD5 digests of the oracle's output for a fixed
set of seeded synthetic pictures (all stages).  These are REGRESSION pins of this repository's own
oracle (they freeze its behaviour across rounds and are checked against the GPU path too); they are
not outputs of the reference, which cannot be built under this round's rules.

    python tools/make_golden.py            # rewrite the fixture
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import pyoracle  # noqa: E402
import pysynth  # noqa: E402

CASES = [
    dict(name="cif8_I", w=352, h=288, bd=8, st=2, seed=1001),
    dict(name="cif8_P_weighted", w=352, h=288, bd=8, st=1, seed=1002, weighted_pred=1),
    dict(name="cif8_B_slices", w=352, h=288, bd=8, st=0, seed=1003, n_slices=3, lf_across_slices_pct=0),
    dict(name="wvga10_B_tiles", w=832, h=480, bd=10, st=0, seed=1004, tile_cols=3, tile_rows=2, slice_per_tile=1,
         lf_across_tiles=0),
    dict(name="wvga10_I_features", w=832, h=480, bd=10, st=2, seed=1005, tskip_pct=30, pcm_pct=10, bypass_pct=5,
         pcm_loop_filter_disable=1, scaling_list=1, constrained_intra_pred=0),
    dict(name="720p8_B", w=1280, h=720, bd=8, st=0, seed=1006, weighted_pred=1, big_coeff_pct=2),
    dict(name="ctb16_12bit_B", w=208, h=120, bd=12, st=0, seed=1007, log2_ctb_size=4, log2_max_tb_size=4),
    dict(name="1080p10_B", w=1920, h=1080, bd=10, st=0, seed=1008),
]


def digest(planes):
    m = hashlib.md5()
    for p in planes:
        m.update(np.ascontiguousarray(p).tobytes())
    return m.hexdigest()


def make_case(c):
    over = {k: v for k, v in c.items() if k not in ("name", "w", "h", "bd", "st", "seed")}
    sp = pysynth.SynthPicture(pysynth.default_config(c["w"], c["h"], c["bd"], c["st"], seed=c["seed"], **over))
    refs = {0: pysynth.fill_planes(c["w"], c["h"], c["bd"], c["seed"] + 1),
            1: pysynth.fill_planes(c["w"], c["h"], c["bd"], c["seed"] + 2)}
    init = pysynth.fill_planes(c["w"], c["h"], c["bd"], c["seed"] + 3)
    return sp, refs, init


def oracle_digests(c):
    sp, refs, init = make_case(c)
    out = {}
    for stage, key in ((0, "prefilter"), (1, "deblocked"), (2, "final")):
        planes = [p.copy() for p in init]
        pyoracle.reconstruct(sp.desc, sp.order, refs, planes, last_stage=stage)
        out[key] = digest(planes)
    d = sp.d
    out["n_tus"], out["n_pus"], out["n_coeffs"] = d.n_tus, d.n_pus, d.n_coeffs
    return out


if __name__ == "__main__":
    res = {c["name"]: oracle_digests(c) for c in CASES}
    path = os.path.join(ROOT, "tests", "golden", "oracle_digests.json")
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
