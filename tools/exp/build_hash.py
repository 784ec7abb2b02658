"""Regression net for the host stage of de265hip_picture_build: hashes everything it would upload, for a fixed list of
random pictures (no GPU needed).   python tools/exp/build_hash.py save|check [file]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import pysynth  # noqa: E402
import ref_sweep  # noqa: E402
from libde265_amd import backend  # noqa: E402


def cases():
    rng = np.random.default_rng(4711)
    out = []
    for it in range(120):
        w, h, bd, st, over = ref_sweep.small_config(rng, it)
        out.append((w, h, bd, st, 60000 + it, over))
    for it in range(24):
        w, h, bd, st, over = ref_sweep.mid_config(rng)
        out.append((w, h, bd, st, 61000 + it, over))
    for it, (cf, st) in enumerate([(2, 2), (2, 0), (3, 2), (3, 0), (3, 1), (2, 1)]):          # range-extension pictures
        out.append((416, 240, 8 + 2 * (it & 1), st, 62000 + it, dict(chroma_format=cf, cross_component_pct=30 if cf == 3 else 0, implicit_rdpcm=1,
                                                                     explicit_rdpcm_pct=30, rotation=1, tskip_pct=20, log2_max_tskip_size=4, bypass_pct=5)))
    out.append((3840, 2160, 10, 2, 0xDE265004, {}))
    out.append((3840, 2160, 10, 0, 0xDE265005, {}))
    out.append((1920, 1080, 8, 0, 0xDE265003, dict(weighted_pred=1)))
    return out


def hashes():
    L = backend.lib()
    res = []
    for (w, h, bd, st, seed, over) in cases():
        sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=seed, **over))
        rc = L.de265hip_debug_build_host_only(sp.desc, 1)
        res.append("%d:%016x" % (rc, L.de265hip_debug_last_build_hash()))
        sp.close()
    return res


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/build_hash.json"
    h = hashes()
    if mode == "save":
        json.dump(h, open(path, "w"))
        print("saved", len(h), "hashes to", path)
    else:
        ref = json.load(open(path))
        bad = [i for i, (a, b) in enumerate(zip(h, ref)) if a != b]
        print("checked", len(h), "pictures:", "IDENTICAL" if not bad else "DIFFERENT at %s" % bad[:10])
        sys.exit(1 if bad else 0)
