#!/usr/bin/env python3
"""Device replay rate of N concurrent GOP streams with / without their I pictures (what limits more than 3 streams?).
    python tools/exp/bstreams.py [--streams 1,2,3,4,6] [--reps 4]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import pysynth  # noqa: E402
from libde265_amd import backend, farm  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--streams", default="1,2,3,4,6")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--gop", type=int, default=16)
ap.add_argument("--over", default="", help="synth overrides for the B pictures, e.g. intra_pct=0")
a = ap.parse_args()
W, H, BD, GOP = 3840, 2160, 10, a.gop
ns = [int(x) for x in a.streams.split(",")]
decs, pics = [], []
for s in range(max(ns)):
    dec = backend.Decoder(); decs.append(dec)
    row = []
    for k, (st, refs) in enumerate(farm.gop_plan(GOP)):
        over = dict(ref_slots=refs) if refs else {}
        if refs:
            over.update({kv.split('=')[0]: int(kv.split('=')[1]) for kv in a.over.split(',') if kv})
        sp = pysynth.SynthPicture(pysynth.default_config(W, H, BD, st, seed=farm.gop_seed(4, s) + k, **over))
        dec.dpb_alloc(k, W, H, BD)
        row.append((sp, dec.build(k, sp.desc)))
    for sp, p in row:
        dec.run(p, 2)
    dec.sync()
    pics.append(row)
for with_i in (1, 0):
    for n in ns:
        ks = list(range(0 if with_i else 1, GOP))
        for d in decs[:n]: d.sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            for j in range(len(ks)):
                for s in range(n):
                    k = ks[(j + s * (len(ks) // n)) % len(ks)]
                    decs[s].run(pics[s][k][1], 2)
        for d in decs[:n]: d.sync()
        dt = time.perf_counter() - t0
        print("%s  streams %d: %7.0f pictures/s  (%.2f ms per GOP pass)" % ("I + 15 B" if with_i else "15 B only", n, a.reps * len(ks) * n / dt, 1e3 * dt / a.reps))
