"""Long random sweep of mid-size pictures on the GPU against the oracle (stage 2 only; the failing iteration is
re-run stage by stage with: sweep.py <seed> <n> <iteration>).  python tools/exp/sweep.py <seed> <n> [only]"""
import os, sys, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pysynth, pyoracle
from libde265_amd import backend, _abi
from test_gpu_picture_parity import run_case, random_midsize_config
dec = backend.Decoder()
rng = np.random.default_rng(int(sys.argv[1]))
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for it in range(int(sys.argv[2])):
    w, h, bd, st, over = random_midsize_config(rng)
    if only >= 0 and it != only: continue
    if only >= 0: print(w, h, bd, st, over, flush=True)
    try:
        run_case(dec, w, h, bd, st, seed=9000 + it, stages=(0, 1, 2) if only >= 0 else (2,), **over)
    except AssertionError as e:
        print('FAILED at iteration', it, w, h, bd, st, over, flush=True); raise
    if it % 10 == 9: print("ok", it + 1, flush=True)
print("sweep passed")
