#!/usr/bin/env python3
"""Summary of a DE265HIP_PIPE_TRACE=1 run (stderr of bench.py): per pipeline, where a picture's time between submission
and launch goes.  Columns of a pipetrace line: pipeline, ticket, submitted, build start, build end, enqueued, launch start,
launch end, scan reported."""
import collections
import statistics as st
import sys


def main(path, lo=64, hi=400):
    rows = collections.defaultdict(list)
    for l in open(path, errors="replace"):
        if not l.startswith("pipetrace"):
            continue
        f = l.split()
        rows[f[1]].append([float(x) for x in f[2:]])
    for p, rs in rows.items():
        rs.sort()
        sel = [r for r in rs if lo <= r[0] < hi]
        if len(sel) < 8:
            continue

        def m(x):
            x = list(x)
            return "%.2f (p90 %.2f)" % (1e3 * st.mean(x), 1e3 * sorted(x)[int(0.9 * len(x))])
        print(p, len(rs), "pictures; tickets", lo, "..", hi)
        print("  wait for a worker ", m(r[2] - r[1] for r in sel))
        print("  build (host stage)", m(r[3] - r[2] for r in sel))
        print("  built -> enqueued ", m(r[4] - r[3] for r in sel))
        print("  enqueued -> scan  ", m(r[7] - r[4] for r in sel if r[7] > 0))
        print("  scan -> launch    ", m(r[5] - r[7] for r in sel if r[7] > 0))
        print("  launch            ", m(r[6] - r[5] for r in sel))
        print("  rate %.0f pictures/s" % ((len(sel) - 1) / (sel[-1][6] - sel[0][6])))
        # what each launch waited for: its own scan (launch right behind ready) or its turn (the previous launch)
        own = sum(1 for a, b in zip(sel, sel[1:]) if b[7] > a[6])
        print("  launches that waited for their own scan: %d of %d" % (own, len(sel) - 1))


if __name__ == "__main__":
    main(sys.argv[1], *[int(x) for x in sys.argv[2:4]])
