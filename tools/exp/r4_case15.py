"""round 4 debugging aid: one picture (tests' mid-size config 15) under several scan / schedule switches"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa
import numpy as np
import pysynth, pyoracle, scan_canon
from libde265_amd import backend
from test_gpu_picture_parity import random_midsize_config
which = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rng = np.random.default_rng(20261004)
for it in range(which + 1):
    w, h, bd, st, over = random_midsize_config(rng)
sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=9000 + it, **over))
refs = {0: pysynth.fill_planes(w, h, bd, 100 + 9000 + it), 1: pysynth.fill_planes(w, h, bd, 200 + 9000 + it)}
init = pysynth.fill_planes(w, h, bd, 999)
exp = [p.copy() for p in init]
pyoracle.reconstruct(sp.desc, sp.order, refs, exp, last_stage=2)
for env in ({}, {"DE265HIP_HOST_SCAN": "1"}, {"DE265HIP_NO_MAILBOX": "1"}, {"DE265HIP_NO_MB_PHASES": "1"}, {"DE265HIP_NO_FRONT": "1"}, {}):
    os.environ.update(env)
    d = backend.Decoder()
    for k in env:
        os.environ.pop(k)
    for s_, pl in refs.items():
        d.dpb_alloc(s_, w, h, bd); d.upload(s_, pl)
    bad_runs = 0
    for rep in range(6):
        pic = d.build(2, sp.desc)
        d.upload(2, init); d.run(pic, 2); d.sync()
        got = d.download(2, w, h, bd)
        nb = sum(int((g != e).sum()) for g, e in zip(got, exp))
        bad_runs += nb > 0
        if rep == 0 and not env:
            h2 = scan_canon.build_dry(sp.desc, 2)
            print("  device vs rehearsal:", scan_canon.diff(scan_canon.canon(pic._h), scan_canon.canon(h2)))
        pic.free()
    print(env, "wrong pictures in 6 runs:", bad_runs)
    d.close()
