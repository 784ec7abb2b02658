#!/usr/bin/env python3
"""How the HOST stage of de265hip_picture_build alone (no HIP call: de265hip_debug_build_host_only) scales over host threads:
N threads each build the bench GOP's B picture `reps` times.  Separates CPU / memory contention from contention in the HIP
runtime (uploads, events, locks)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import pysynth
from libde265_amd import backend, farm
os.environ["DE265HIP_DRY_NO_HASH"] = "1"
W, H, BD = 3840, 2160, 10
st, refs = farm.gop_plan(2)[1]
sp = pysynth.SynthPicture(pysynth.default_config(W, H, BD, st, seed=farm.gop_seed(4, 0) + 1, ref_slots=refs))
L = backend.lib()
reps = 20
for n in (1, 2, 4, 8, 16, 32, 64):
    ths = [threading.Thread(target=lambda: L.de265hip_debug_build_host_only(sp.desc, reps)) for _ in range(n)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("%2d threads: %.2f ms per build per thread, %.0f builds/s" % (n, 1e3 * dt / reps, n * reps / dt))
