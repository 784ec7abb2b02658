"""Build-container sweep: CPU restatement (oracle/) against the COMPILED REFERENCE (oracle/_ref) on random
synthetic pictures, every stage, plus edge flags and boundary strengths.  Usage:
    python tools/ref_sweep.py [seed] [n_small] [n_mid] [n_formats]
n_formats: that many more small / mid-size pictures in the other chroma formats - 4:2:2 and 4:4:4 with the range-extension sample
tools drawn at random, monochrome intra pictures.  Prints one line per mismatching case and a summary; exit code 1 on any mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import pyoracle  # noqa: E402
import pyref  # noqa: E402
import pysynth  # noqa: E402


def small_config(rng, it):
    log2_ctb = int(rng.choice([4, 5, 6]))
    w = int(rng.integers(1, 26)) * 8 if it % 7 else int(rng.integers(26, 80)) * 8
    h = int(rng.integers(1, 18)) * 8 if it % 7 else int(rng.integers(18, 48)) * 8
    bd = int(rng.choice([8, 9, 10, 12]))
    st = int(rng.choice([0, 1, 2]))
    cols = int(rng.integers(1, 3)) if w >= 128 else 1
    rows = int(rng.integers(1, 3)) if h >= 128 else 1
    ctbs_w = -(-w // (1 << log2_ctb))
    ctbs_h = -(-h // (1 << log2_ctb))
    cols, rows = min(cols, ctbs_w), min(rows, ctbs_h)
    over = dict(log2_ctb_size=log2_ctb, log2_max_tb_size=min(5, log2_ctb),
                log2_min_tb_size=int(rng.choice([2, 2, 3])),
                intra_pct=int(rng.choice([0, 15, 50, 100])), tskip_pct=int(rng.choice([0, 30])),
                bypass_pct=int(rng.choice([0, 10])), pcm_pct=int(rng.choice([0, 20])),
                pcm_loop_filter_disable=int(rng.integers(0, 2)), scaling_list=int(rng.integers(0, 2)),
                constrained_intra_pred=int(rng.integers(0, 2)), strong_intra_smoothing=int(rng.integers(0, 2)),
                weighted_pred=int(rng.integers(0, 2)), n_slices=int(rng.integers(1, 4)),
                tile_cols=cols, tile_rows=rows, slice_per_tile=int(rng.integers(0, 2)),
                lf_across_tiles=int(rng.integers(0, 2)), lf_across_slices_pct=int(rng.choice([0, 50, 100])),
                deblocking=int(rng.integers(0, 4) > 0), sao=int(rng.integers(0, 4) > 0),
                big_coeff_pct=int(rng.choice([0, 5])), mv_sigma_qpel=int(rng.choice([4, 12, 60])),
                split_bias=int(rng.choice([0, 50, 100])), cbf_pct=int(rng.choice([0, 60, 100])))
    return w, h, bd, st, over


def mid_config(rng):
    log2_ctb = int(rng.choice([4, 5, 6, 6]))
    w = int(rng.integers(40, 241)) * 8
    h = int(rng.integers(30, 137)) * 8
    bd = int(rng.choice([8, 10, 10, 12]))
    st = int(rng.choice([0, 0, 1, 2]))
    over = dict(log2_ctb_size=log2_ctb, log2_max_tb_size=min(5, log2_ctb), log2_min_tb_size=int(rng.choice([2, 2, 3])),
                intra_pct=int(rng.choice([5, 15, 40, 100])), tskip_pct=int(rng.choice([0, 20])),
                bypass_pct=int(rng.choice([0, 5])), pcm_pct=int(rng.choice([0, 10])), scaling_list=int(rng.integers(0, 2)),
                constrained_intra_pred=int(rng.integers(0, 2)), strong_intra_smoothing=int(rng.integers(0, 2)),
                weighted_pred=int(rng.integers(0, 2)), n_slices=int(rng.integers(1, 5)),
                split_bias=int(rng.choice([0, 30, 50, 80, 100])), cbf_pct=int(rng.choice([30, 60, 100])),
                mv_sigma_qpel=int(rng.choice([4, 12, 80])), pcm_loop_filter_disable=int(rng.integers(0, 2)),
                lf_across_slices_pct=int(rng.choice([0, 50, 100])), lf_across_tiles=int(rng.integers(0, 2)),
                big_coeff_pct=int(rng.choice([0, 0, 2])), bi_pct=int(rng.choice([0, 60, 100])), amp=int(rng.integers(0, 2)),
                deblocking=int(rng.choice([1, 1, 1, 0])), sao=int(rng.choice([1, 1, 1, 0])))
    if log2_ctb >= 5 and rng.integers(0, 3) == 0:
        over.update(tile_cols=int(rng.integers(1, 4)), tile_rows=int(rng.integers(1, 3)),
                    slice_per_tile=int(rng.integers(0, 2)))
    qlo = int(rng.integers(0, 40))
    over.update(qp_min=qlo, qp_max=int(rng.integers(qlo, 52)))
    return w, h, bd, st, over


def format_config(rng, it):
    """A small (two of three) or mid-size configuration in another chroma format."""
    w, h, bd, st, over = mid_config(rng) if it % 3 == 2 else small_config(rng, it)
    cf = int(rng.choice([0, 2, 2, 3, 3]))
    over.update(tskip_pct=int(rng.choice([0, 20, 40])), implicit_rdpcm=int(rng.integers(0, 2)), rotation=int(rng.integers(0, 2)),
                log2_max_tskip_size=int(rng.integers(2, 6)), intra_smoothing_disabled=int(rng.integers(0, 2)))
    if cf == 0:
        st = 2
        over.update(monochrome=1)
    else:
        over.update(chroma_format=cf, explicit_rdpcm_pct=int(rng.choice([0, 50])), high_precision_offsets=int(rng.integers(0, 2)),
                    cross_component_pct=int(rng.choice([0, 60])) if cf == 3 else 0)
    return w, h, bd, st, over, cf


def compare(w, h, bd, st, seed, over, stages=(0, 1, 2), cf=1):
    """Returns a list of mismatch descriptions (empty = identical)."""
    cfg = pysynth.default_config(w, h, bd, st, seed=seed, **over)
    sp = pysynth.SynthPicture(cfg)
    structure = sp.structure()
    refs = {} if cf == 0 else {0: pysynth.fill_planes(w, h, bd, 100 + seed, cf), 1: pysynth.fill_planes(w, h, bd, 200 + seed, cf)}
    init = pysynth.fill_planes(w, h, bd, 999, cf)
    bad = []
    for stage in stages:
        a = [p.copy() for p in init]
        b = [p.copy() for p in init]
        pyoracle.reconstruct(sp.desc, sp.order, refs, a, stage)
        pyref.reconstruct(sp.desc, sp.order, refs, b, structure, stage)
        for c in range(3):
            n = int((a[c] != b[c]).sum())
            if n:
                yx = tuple(int(v) for v in np.argwhere(a[c] != b[c])[0])
                bad.append("stage %d comp %d: %d samples differ, first (y,x)=%s oracle %d ref %d"
                           % (stage, c, n, yx, a[c][yx], b[c][yx]))
    # a11 / a12
    ef_ref = pyref.derive_edge_flags(sp.desc, structure)
    ef_gen = sp.blk_flags()
    if not np.array_equal(ef_ref, ef_gen):
        bad.append("edge flags: %d units differ (reference vs generator)" % int((ef_ref != ef_gen).sum()))
    for vertical in (1, 0):
        bs_ref = pyref.derive_bs(sp.desc, structure, vertical)
        w4, h4 = ef_ref.shape[1], ef_ref.shape[0]
        bs_or = np.zeros((h4, w4), np.uint8)
        pyoracle.lib().oracle_derive_bs(sp.desc, vertical, bs_or.ctypes.data)
        # the reference only writes the units it visits (8-sample grid of the direction, deblock.cc:244-245)
        m = np.zeros_like(bs_ref, bool)
        if vertical:
            m[:, ::2] = True
        else:
            m[::2, :] = True
        if not np.array_equal(bs_ref[m], bs_or[m]):
            bad.append("bS %s: %d units differ" % ("V" if vertical else "H", int((bs_ref[m] != bs_or[m]).sum())))
    sp.close()
    return bad


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n_small = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    n_mid = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    rng = np.random.default_rng(seed)
    fails = 0
    for it in range(n_small):
        w, h, bd, st, over = small_config(rng, it)
        bad = compare(w, h, bd, st, 5000 + seed * 100000 + it, over)
        if bad:
            fails += 1
            print("SMALL %d: %dx%d bd=%d st=%d %r" % (it, w, h, bd, st, over))
            for b in bad:
                print("   ", b)
    for it in range(n_mid):
        w, h, bd, st, over = mid_config(rng)
        bad = compare(w, h, bd, st, 9000 + seed * 100000 + it, over)
        if bad:
            fails += 1
            print("MID %d: %dx%d bd=%d st=%d %r" % (it, w, h, bd, st, over))
            for b in bad:
                print("   ", b)
    n_fmt = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    count = {0: 0, 2: 0, 3: 0}
    for it in range(n_fmt):
        w, h, bd, st, over, cf = format_config(rng, it)
        bad = compare(w, h, bd, st, 13000 + seed * 100000 + it, over, cf=cf)
        count[cf] += 1
        if bad:
            fails += 1
            print("FORMAT %d: %dx%d bd=%d st=%d %r" % (it, w, h, bd, st, over))
            for b in bad:
                print("   ", b)
    if n_fmt:
        print("other formats: %d monochrome, %d 4:2:2, %d 4:4:4 pictures" % (count[0], count[2], count[3]))
    print("ref_sweep seed %d: %d small + %d mid pictures, %d with mismatches" % (seed, n_small, n_mid, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
