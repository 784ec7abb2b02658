import os, sys, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pysynth, pyoracle
from libde265_amd import backend, _abi
dec = backend.Decoder()
over = {'log2_ctb_size': 5, 'log2_max_tb_size': 5, 'log2_min_tb_size': 3, 'intra_pct': 40, 'tskip_pct': 20, 'bypass_pct': 0, 'pcm_pct': 0, 'scaling_list': 0, 'constrained_intra_pred': 0, 'strong_intra_smoothing': 1, 'weighted_pred': 1, 'n_slices': 4, 'split_bias': 0, 'cbf_pct': 60, 'mv_sigma_qpel': 12}
w, h, bd, st, seed = 1416, 536, 10, 1, 9045
cfg = pysynth.default_config(w, h, bd, st, seed=seed, **over)
sp = pysynth.SynthPicture(cfg)
refs = {0: pysynth.fill_planes(w, h, bd, 100 + seed), 1: pysynth.fill_planes(w, h, bd, 200 + seed)}
for slot, pl in refs.items():
    dec.dpb_alloc(slot, w, h, bd); dec.upload(slot, pl)
pic = dec.build(2, sp.desc)
init = pysynth.fill_planes(w, h, bd, 999)
exp = [p.copy() for p in init]
pyoracle.reconstruct(sp.desc, sp.order, refs, exp, last_stage=0)
dec.upload(2, init); dec.run(pic, 0); dec.sync()
got = dec.download(2, w, h, bd)
d = sp.d
tus = [d.tus[i] for i in range(d.n_tus)]
pus = [d.pus[i] for i in range(d.n_pus)]
for c in range(3):
    bad = np.argwhere(got[c] != exp[c])
    print("comp", c, "mismatches", len(bad))
    if not len(bad): continue
    ys, xs = bad[:, 0], bad[:, 1]
    print(" bbox y", ys.min(), ys.max(), "x", xs.min(), xs.max())
    # which TUs / PUs cover the mismatches
    seen = {}
    for (y, x) in bad[:4000]:
        for i, t in enumerate(tus):
            if t.c_idx != c: continue
            n = 1 << t.log2_size
            if t.x0 <= x < t.x0 + n and t.y0 <= y < t.y0 + n:
                seen.setdefault(("tu", i), 0); seen[("tu", i)] += 1
        sc = 1 if c == 0 else 2
        for i, p in enumerate(pus):
            if p.x <= x * sc < p.x + p.w and p.y <= y * sc < p.y + p.h:
                seen.setdefault(("pu", i), 0); seen[("pu", i)] += 1
    for (kind, i), n in sorted(seen.items(), key=lambda kv: -kv[1])[:12]:
        if kind == "tu":
            t = tus[i]; print("  TU", i, "x0,y0", t.x0, t.y0, "log2", t.log2_size, "flags", hex(t.flags), "mode", t.intra_mode, "qp", t.qp, "ncoeff", t.n_coeff, "bad", n)
        else:
            p = pus[i]; print("  PU", i, "x,y", p.x, p.y, "w,h", p.w, p.h, "pred_flag", p.pred_flag, "slice", p.slice_idx, "ref", list(p.ref_idx), "mv", [list(m) for m in p.mv], "bad", n)
    y, x = bad[0]
    print(" first", y, x, "got", got[c][y, x], "exp", exp[c][y, x])
    print(" got row ", got[c][y, max(0, x - 4):x + 12])
    print(" exp row ", exp[c][y, max(0, x - 4):x + 12])
sl = d.slices
for i in range(d.n_slices):
    s = sl[i]
    print("slice", i, "type", s.slice_type, "addr", s.slice_addr_rs, "denomL", s.luma_log2_weight_denom, "wL0", list(s.luma_weight[0])[:4], "oL0", list(s.luma_offset[0])[:4])
