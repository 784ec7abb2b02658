import os, sys, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pysynth, pyoracle
from libde265_amd import backend, _abi
from test_gpu_picture_parity import run_case
dec = backend.Decoder()
base = {'log2_ctb_size': 5, 'log2_max_tb_size': 5, 'log2_min_tb_size': 3, 'intra_pct': 40, 'tskip_pct': 20, 'bypass_pct': 0, 'pcm_pct': 0, 'scaling_list': 0, 'constrained_intra_pred': 0, 'strong_intra_smoothing': 1, 'weighted_pred': 1, 'n_slices': 4, 'split_bias': 0, 'cbf_pct': 60, 'mv_sigma_qpel': 12}
def tryit(name, w=1416, h=536, bd=10, st=1, **chg):
    o = dict(base); o.update(chg)
    try:
        run_case(dec, w, h, bd, st, seed=9045, stages=(0,), **o)
        print(name, "pass", flush=True)
    except AssertionError as e:
        print(name, "FAIL", str(e)[:110], flush=True)
tryit("base")
tryit("no weighted", weighted_pred=0)
tryit("no intra", intra_pct=0)
tryit("all intra", intra_pct=100)
tryit("no tskip", tskip_pct=0)
tryit("1 slice", n_slices=1)
tryit("cbf 0", cbf_pct=0)
tryit("w 1408", w=1408)
tryit("h 512", h=512)
tryit("B slice", st=0)
tryit("8 bit", bd=8)
tryit("no strong", strong_intra_smoothing=0)
tryit("min tb 2", log2_min_tb_size=2)
