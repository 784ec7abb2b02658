#!/usr/bin/env python3
"""Summarises two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of bench.py into the per-kernel
traffic file bench.py reads (profiles/rNN_x_pmc_traffic.json).

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> "<command line that was profiled>"

Values are KB per launch as rocprofv3 reports them (FETCH_SIZE is NOT doubled here; bench.py applies the
gfx950 correction of MI355X_MICROARCH.md when it builds the roofline object)."""
import csv
import glob
import json
import os
import re
import sys


def collect(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            m = re.match(r"(?:void )?(?:d265::)?([A-Za-z_0-9]+(?:<[^(]*>)?)", name)
            key = m.group(1) if m else name
            a = acc.setdefault(key, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


def main():
    fetch_dir, write_dir, out, cmd = sys.argv[1:5]
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith("k_"):
            continue
        f, w = fe.get(k, [0.0, 0]), wr.get(k, [0.0, 0])
        kernels[k] = {"launches": f[1] or w[1],
                      "fetch_kb_per_launch": round(f[0] / max(f[1], 1), 1),
                      "write_kb_per_launch": round(w[0] / max(w[1], 1), 1)}
    json.dump({"_command": cmd,
               "_units": "KB per launch, averaged over all launches of the kernel in the run (I and B pictures mixed); "
                         "FETCH_SIZE is what rocprofv3 reports, NOT yet doubled (MI355X_MICROARCH.md: gfx950 reports "
                         "half the bytes of 16-byte-per-lane streaming reads)",
               "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
