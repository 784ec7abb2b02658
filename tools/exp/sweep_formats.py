"""Random mid-size pictures in the other chroma formats on the GPU against the oracle: 4:2:2 / 4:4:4 with the range-extension
sample tools drawn at random, and monochrome intra pictures.   python tools/exp/sweep_formats.py <seed> <n> [only]"""
import os, sys, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pysynth, pyoracle
from libde265_amd import backend
from test_gpu_picture_parity import random_midsize_config
rng = np.random.default_rng(int(sys.argv[1]))
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
dec = backend.Decoder()
count = {0: 0, 2: 0, 3: 0}
for it in range(int(sys.argv[2])):
    w, h, bd, st, over = random_midsize_config(rng)
    cf = int(rng.choice([0, 2, 2, 3, 3]))
    over.update(tskip_pct=int(rng.choice([0, 20, 40])), implicit_rdpcm=int(rng.integers(0, 2)), rotation=int(rng.integers(0, 2)),
                log2_max_tskip_size=int(rng.integers(2, 6)), intra_smoothing_disabled=int(rng.integers(0, 2)))
    if cf == 0:
        st = 2; over.update(monochrome=1)
        for k in ("weighted_pred", "bi_pct", "mv_sigma_qpel", "amp", "intra_pct"): over.pop(k, None)
    else:
        over.update(chroma_format=cf, explicit_rdpcm_pct=int(rng.choice([0, 50])), high_precision_offsets=int(rng.integers(0, 2)),
                    cross_component_pct=int(rng.choice([0, 60])) if cf == 3 else 0)
        if cf == 3 and over.get("scaling_list"): over["log2_max_tb_size"] = min(over["log2_max_tb_size"], 4)   # (32x32 chroma TUs with scaling lists: undefined in the reference)
    if only >= 0 and it != only: continue
    if only >= 0: print(w, h, bd, st, over, flush=True)
    sp = pysynth.SynthPicture(pysynth.default_config(w, h, bd, st, seed=11000 + it, **over))
    refs = {} if cf == 0 else {0: pysynth.fill_planes(w, h, bd, 100 + it, cf), 1: pysynth.fill_planes(w, h, bd, 200 + it, cf)}
    for slot, pl in refs.items():
        dec.dpb_alloc(slot, w, h, bd, chroma_format=cf); dec.upload(slot, pl)
    dec.dpb_alloc(2, w, h, bd, chroma_format=cf)
    pic = dec.build(2, sp.desc)
    for stage in ((0, 1, 2) if only >= 0 else (2,)):
        init = pysynth.fill_planes(w, h, bd, 999, cf)
        exp = [p.copy() for p in init]
        pyoracle.reconstruct(sp.desc, sp.order, refs, exp, last_stage=stage)
        dec.upload(2, init); dec.run(pic, stage); dec.sync()
        got = dec.download(2, w, h, bd)
        for c in range(3):
            bad = np.argwhere(got[c] != exp[c])
            if bad.size:
                print("FAILED at iteration", it, "stage", stage, "comp", c, len(bad), "mismatches, first", tuple(bad[0]), w, h, bd, st, over, flush=True)
                sys.exit(1)
    pic.free(); sp.close(); count[cf] += 1
    if it % 10 == 9: print("ok", it + 1, flush=True)
print("format sweep passed: %d monochrome, %d 4:2:2, %d 4:4:4 pictures identical" % (count[0], count[2], count[3]))
