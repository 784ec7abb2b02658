import os, sys, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pysynth, pyoracle
from libde265_amd import backend, _abi
from test_gpu_picture_parity import run_case
dec = backend.Decoder()
rng = np.random.default_rng(int(sys.argv[1]))
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
for it in range(int(sys.argv[2])):
    log2_ctb = int(rng.choice([4, 5, 6, 6]))
    w = int(rng.integers(40, 241)) * 8; h = int(rng.integers(30, 137)) * 8
    bd = int(rng.choice([8, 10, 10, 12])); st = int(rng.choice([0, 0, 1, 2]))
    over = dict(log2_ctb_size=log2_ctb, log2_max_tb_size=min(5, log2_ctb), log2_min_tb_size=int(rng.choice([2, 2, 3])),
                intra_pct=int(rng.choice([5, 15, 40, 100])), tskip_pct=int(rng.choice([0, 20])), bypass_pct=int(rng.choice([0, 5])),
                pcm_pct=int(rng.choice([0, 10])), scaling_list=int(rng.integers(0, 2)), constrained_intra_pred=int(rng.integers(0, 2)),
                strong_intra_smoothing=int(rng.integers(0, 2)), weighted_pred=int(rng.integers(0, 2)), n_slices=int(rng.integers(1, 5)),
                split_bias=int(rng.choice([0, 30, 50, 80, 100])), cbf_pct=int(rng.choice([30, 60, 100])), mv_sigma_qpel=int(rng.choice([4, 12, 80])))
    if only >= 0 and it != only: continue
    if only >= 0: print(w, h, bd, st, over, flush=True)
    try:
        run_case(dec, w, h, bd, st, seed=9000 + it, stages=(0, 1, 2) if only >= 0 else (2,), **over)
    except AssertionError as e:
        print('FAILED at iteration', it, w, h, bd, st, over, flush=True); raise
    if it % 10 == 9: print("ok", it + 1, flush=True)
print("sweep passed")
