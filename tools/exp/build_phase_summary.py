#!/usr/bin/env python3
"""Averages the per-phase host times printed by DE265HIP_BUILD_TIMING=1 (stderr lines 'de265hip build: a=1.2ms b=...')."""
import re, sys, collections
acc, n = collections.OrderedDict(), 0
for line in sys.stdin:
    if not line.startswith("de265hip build:"):
        continue
    n += 1
    for k, v in re.findall(r"(\w+)=([\d.]+)ms", line):
        acc[k] = acc.get(k, 0.0) + float(v)
print("%d builds; mean ms per phase: " % n + " ".join("%s=%.2f" % (k, v / max(n, 1)) for k, v in acc.items()) + " | total %.2f" % (sum(acc.values()) / max(n, 1)))
