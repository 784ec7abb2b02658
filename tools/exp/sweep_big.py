"""A few random LARGE pictures (up to 4096x2304, all bit depths, all slice types) on the GPU against the oracle.
python tools/exp/sweep_big.py <seed> <n>"""
import os, sys, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from libde265_amd import backend
from test_gpu_picture_parity import run_case, random_midsize_config
dec = backend.Decoder()
rng = np.random.default_rng(int(sys.argv[1]))
for it in range(int(sys.argv[2])):
    _, _, bd, st, over = random_midsize_config(rng)
    w = int(rng.integers(240, 513)) * 8; h = int(rng.integers(136, 289)) * 8
    print(it, w, h, bd, st, over, flush=True)
    run_case(dec, w, h, bd, st, seed=7000 + it, stages=(2,), **over)
print("big sweep passed")
