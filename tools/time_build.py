#!/usr/bin/env python3
"""Host time of de265hip_picture_build for the bench GOP's I picture and first B pictures (4K Main10).
    python tools/time_build.py              on the GPU box: the real call (host stage + staging + async upload)
    python tools/time_build.py --host-only  anywhere: de265hip_debug_build_host_only (no HIP call at all)"""
import os
os.environ.setdefault("DE265HIP_TUNING", "1")      # (the library reads its DE265HIP_* switches only then: csrc/env.h)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import pysynth  # noqa: E402
from libde265_amd import backend, farm  # noqa: E402

W, H, BD = 3840, 2160, 10
host_only = "--host-only" in sys.argv
dec = None if host_only else backend.Decoder()
for k, (st, refs) in enumerate(farm.gop_plan(3)):
    over = dict(ref_slots=refs) if refs else {}
    sp = pysynth.SynthPicture(pysynth.default_config(W, H, BD, st, seed=farm.gop_seed(4, 0) + k, **over))
    best = 1e9
    if host_only:
        for _ in range(3):
            t0 = time.perf_counter()
            rc = backend.lib().de265hip_debug_build_host_only(sp.desc, 1)
            best = min(best, time.perf_counter() - t0)
            assert rc == 0, rc
        print("picture %d (%s): host stage %.1f ms, %d TUs, %d PUs" % (k, "I" if k == 0 else "B", 1e3 * best, sp.d.n_tus, sp.d.n_pus))
        continue
    dec.dpb_alloc(k, W, H, BD)
    for _ in range(3):
        t0 = time.perf_counter()
        pic = dec.build(k, sp.desc)
        best = min(best, time.perf_counter() - t0)
        st_ = pic.stats()
        pic.free()
    print("picture %d (%s): build %.1f ms, %d TU tasks, %d runs, arena %.1f MB" % (
        k, "I" if k == 0 else "B", 1e3 * best, st_.n_tu_tasks, st_.n_runs, st_.device_bytes / 1e6))
