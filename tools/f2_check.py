#!/usr/bin/env python3
"""SURVEY.md 8(f2) self-check (build container only): write a synthetic stream with oracle/_ref/f2_writer, decode it with
the recording reference decoder oracle/_ref/f1_dec, and compare what the decoder parsed with what the writer coded (the
.chk sidecar: CUs are not recorded, but PUs, PCM blocks, residual blocks, coefficient count and sum of |level| are).  Any
CABAC bin the writer binarises or context-selects differently from slice.cc desynchronises the arithmetic decoder and
changes these numbers, so equality over many seeds is the evidence that the writer emits the syntax it believes it does.

    python tools/f2_check.py gop=B pics=5 w=128 h=96 seeds=0-19 [replay=1] [any f2_writer key=value]

replay=1 additionally replays every recorded picture through the CPU restatement (oracle/) and compares it, before and
after the post-filters, with libde265's own output: the oracle-vs-reference sweep on REAL bitstream semantics.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import f1_stream  # noqa: E402

REFDIR = os.path.join(ROOT, "oracle", "_ref")


def decode_stats(dumpdir):
    out = []
    for fn in sorted(os.listdir(dumpdir)):
        rp, pre, fin = f1_stream.load_dump(os.path.join(dumpdir, fn))
        tus = np.frombuffer(rp.a["tus"].tobytes(), dtype=np.dtype(f1_stream._abi.TU)) if rp.meta["n_tus"] else None
        out.append(dict(poc=rp.meta["poc"], pus=rp.meta["n_pus"], pcms=rp.meta["n_pcms"], coeffs=rp.meta["n_coeffs"],
                        abs_sum=int(np.abs(rp.a["coeff_val"].astype(np.int64)).sum()), rp=rp))
    return out


REPLAY = False


def replay(dumpdir):
    """the CPU restatement replays the recorded pictures (DPB chained through the decoder's own slots) to libde265's output"""
    import pyoracle
    from libde265_amd import _abi
    dpb = {}
    for fn in sorted(os.listdir(dumpdir)):
        rp, pre, fin = f1_stream.load_dump(os.path.join(dumpdir, fn))
        P, d = rp.params, rp.to_desc()
        for stage, want in ((_abi.STAGE_PREFILTER, pre), (_abi.STAGE_FINAL, fin)):
            out = pyoracle.alloc_planes(P.width, P.height, P.bit_depth_luma, chroma_format=P.chroma_format_idc)
            pyoracle.reconstruct(d, None, dpb, out, stage)
            diff = [int((a != b).sum()) for a, b in zip(out, want)]
            if any(diff):
                return "oracle != libde265 at poc %d stage %d: %s samples differ" % (rp.meta["poc"], stage, diff)
        dpb[rp.meta["dst_slot"]] = [p.copy() for p in fin]
    return None


def run(args, seed, keep=None):
    with tempfile.TemporaryDirectory() as td:
        bits = os.path.join(td, "s.bin")
        subprocess.check_call([os.path.join(REFDIR, "f2_writer"), "out=" + bits, "seed=%d" % seed] + args)
        dumps = os.path.join(td, "d")
        os.mkdir(dumps)
        r = subprocess.run([os.path.join(REFDIR, "f1_dec"), bits], env=dict(os.environ, F1_OUT=dumps), capture_output=True, text=True)
        want = [dict(zip(l.split()[0::2], (int(v) for v in l.split()[1::2]))) for l in open(bits + ".chk")]
        if r.returncode or r.stderr.strip():
            return "decoder: rc %d %s" % (r.returncode, r.stderr.strip()[:300]), want, []
        got = decode_stats(dumps)
        if REPLAY:
            err = replay(dumps)
            if err:
                return err, want, got
        if keep:
            os.replace(bits, keep)
        if len(got) != len(want):
            return "%d pictures decoded, %d written" % (len(got), len(want)), want, got
        for w, g in zip(want, got):
            for k in ("poc", "pus", "pcms", "coeffs", "abs_sum"):
                if w[k] != g[k]:
                    return "picture poc %d: %s written %d, decoded %d" % (w["poc"], k, w[k], g[k]), want, got
        return None, want, got


def main():
    args, seeds = [], [0]
    global REPLAY
    for a in sys.argv[1:]:
        if a == "replay=1":
            REPLAY = True
        elif a.startswith("seeds="):
            lo, _, hi = a[6:].partition("-")
            seeds = list(range(int(lo), int(hi or lo) + 1))
        else:
            args.append(a)
    bad = 0
    for s in seeds:
        err, want, got = run(args, s)
        if err:
            bad += 1
            print("seed %d: MISMATCH %s" % (s, err))
        else:
            print("seed %d: ok  %s" % (s, " ".join("poc%d:%dpu/%dco" % (w["poc"], w["pus"], w["coeffs"]) for w in want)))
    print("%d of %d seeds in sync" % (len(seeds) - bad, len(seeds)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
