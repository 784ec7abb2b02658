#!/usr/bin/env python3
"""Per-picture device-time breakdown of the bench GOP (hipEvent timers of the library).
    python tools/profile_gop.py [--pictures 4] [--reps 5]"""
import argparse
import os
os.environ.setdefault("DE265HIP_TUNING", "1")      # (the library reads its DE265HIP_* switches only then: csrc/env.h)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import pysynth  # noqa: E402
from libde265_amd import backend, farm, _abi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pictures", type=int, default=4)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--bit-depth", type=int, default=10)
ap.add_argument("--only", type=int, default=-1, help="time only this picture of the GOP")
ap.add_argument("--over", default="", help="synth overrides, e.g. cbf_pct=0,split_bias=100")
ap.add_argument("--md5", action="store_true", help="print the md5 of every decoded picture (to compare library variants)")
a = ap.parse_args()
W, H, BD = a.width, a.height, a.bit_depth
dec = backend.Decoder()
pics = []
for k, (st, refs) in enumerate(farm.gop_plan(a.pictures)):
    over = dict(ref_slots=refs) if refs else {}
    over.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.over.split(",") if kv})
    sp = pysynth.SynthPicture(pysynth.default_config(W, H, BD, st, seed=farm.gop_seed(4, 0) + k, **over))
    dec.dpb_alloc(k, W, H, BD)
    pics.append((sp, dec.build(k, sp.desc)))
for sp, p in pics:
    dec.run(p, 2)
dec.sync()
dec.set_profiling(True)
dec.sync()
for k, (sp, p) in enumerate(pics):
    if a.only >= 0 and k != a.only:
        continue
    dec.kernel_times(reset=True)
    for _ in range(a.reps):
        dec.run(p, 2)
    dec.sync()
    kt = dec.kernel_times(reset=True)
    s = p.stats()
    print("pic %d type %s tus %d mc %d levels %d runs %d runlevels %d inrunlevels %d | " % (
        k, "I" if k == 0 else "B", s.n_tu_tasks, s.n_mc_tasks, s.n_levels, s.n_runs, s.n_run_levels, s.n_in_run_levels) +
          " ".join("%s=%.1fus" % (n, 1e3 * v[0] / a.reps) for n, v in kt.items() if v[1]))
    if a.md5:
        import hashlib
        h = hashlib.md5()
        for pl in dec.download(k, W, H, BD):
            h.update(pl.tobytes())
        print("pic %d md5 %s" % (k, h.hexdigest()))
