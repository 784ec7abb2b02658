"""Host stage alone (no HIP call) in T concurrent threads, one descriptor each: does a build slow down when its neighbours build too?
    python tools/exp/dry_scaling.py 1 2 4 8 15"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import pysynth
from libde265_amd import backend, farm
L = backend.lib()
W, H, BD = 3840, 2160, 10
plan = farm.gop_plan(3)
pics = []
for k in (0, 1):
    st, refs = plan[k]
    over = dict(ref_slots=refs) if refs else {}
    pics.append(pysynth.SynthPicture(pysynth.default_config(W, H, BD, st, seed=farm.gop_seed(4, 0) + k, **over)))
for T in [int(a) for a in sys.argv[1:]]:
    for k, name in ((0, "I"), (1, "B")):
        res = [0.0] * T
        def work(i):
            L.de265hip_debug_build_host_only(pics[k].desc, 1)        # warm the thread's scratch
            t0 = time.perf_counter()
            L.de265hip_debug_build_host_only(pics[k].desc, 6)
            res[i] = (time.perf_counter() - t0) / 6
        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        [t.start() for t in th]; [t.join() for t in th]
        print("threads %2d %s: %.2f ms per build (min %.2f max %.2f)" % (T, name, 1e3 * sum(res) / T, 1e3 * min(res), 1e3 * max(res)), flush=True)
